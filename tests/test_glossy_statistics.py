"""Glossy reflection: the per-pixel-seeded (path-hashed) random stream the device uses against the reference's libc rand()
stream (src/shading.cpp:169-195, src/utils.cpp:63-69, src/utils.h:31).  The two cannot agree bit for bit; SURVEY.md 8(f).2
asks for a statistical link: same mean and same variance per tile, and a stream that is uniform and uncorrelated.
CPU only: both streams are rendered by the oracle (rng_mode 0 = libc in call order, rng_mode 1 = the hash); the device
equals rng_mode 1 bit-tolerantly in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import oracle


K = 6      # libc renders per rectangle (the libc stream simply continues from one to the next: independent samples)


def _noise_stats(s, rect):
    """K independent libc renders and the hashed render of one rectangle."""
    x0, y0, x1, y1 = rect
    a = np.stack([oracle.render_rect(s, x0, y0, x1, y1, rng_mode=0, threads=1)[0].astype(np.float64) for _ in range(K)])
    b, _, _ = oracle.render_rect(s, x0, y0, x1, y1, rng_mode=1, threads=8)
    return a, b.astype(np.float64)


def _check(a, b, tile=32):
    noisy = np.abs(a - a[0]).max(axis=(0, 3)) > 0            # pixels that draw random numbers at all
    assert noisy.sum() > 2000, int(noisy.sum())
    # everything else does not depend on the stream (a handful of glossy pixels whose libc renders happen to agree --
    # all samples land on one flat colour -- may still differ in the hashed one)
    assert (np.abs(a[0] - b).max(axis=2)[~noisy] > 0).sum() <= max(4, noisy.sum() // 1000)
    m = a.mean(axis=0)                                       # libc mean image: noise sigma_libc^2 / K per pixel
    _, H, W, _ = a.shape
    n_tiles = 0
    pooled = np.zeros(2)                                     # sums of squared noise: hashed stream, libc stream
    for ty in range(0, H, tile):
        for tx in range(0, W, tile):
            msk = noisy[ty:ty + tile, tx:tx + tile]
            if msk.sum() < 200:
                continue
            n_tiles += 1
            for ch in range(3):
                A = a[:, ty:ty + tile, tx:tx + tile, ch][:, msk]            # K x n
                B = b[ty:ty + tile, tx:tx + tile, ch][msk]
                M = m[ty:ty + tile, tx:tx + tile, ch][msk]
                n = B.size
                sq_libc = ((A - M) ** 2).sum(axis=0) / (K - 1)              # per pixel, unbiased for sigma_libc^2
                sq_hash = (B - M) ** 2 - sq_libc / K                        # per pixel, unbiased for sigma_hash^2
                s2_libc, s2_hash = sq_libc.mean(), sq_hash.mean()
                # mean of the tile: hashed render vs libc mean image, 3 sigma of the difference of the two tile means
                d = B.mean() - M.mean()
                assert abs(d) <= 3 * np.sqrt(max(s2_hash, 0) / n + s2_libc / (K * n)) + 1e-6, (tx, ty, ch, d)
                # variance of the tile: inside [0.8, 1.25], or inside the sampling error of a variance estimated from n
                # heavy-tailed values (highlights reach 10x the tile mean): 4 standard errors of the difference
                se = np.sqrt((sq_hash.var() + sq_libc.var()) / n)
                assert 0.8 <= s2_hash / s2_libc <= 1.25 or abs(s2_hash - s2_libc) <= 4 * se, (tx, ty, ch, s2_hash / s2_libc)
                pooled += [sq_hash.sum(), sq_libc.sum()]
    assert n_tiles >= 4, n_tiles
    assert 0.8 <= pooled[0] / pooled[1] <= 1.25, pooled[0] / pooled[1]      # all tiles together: the bound itself


def test_glossy_sphere_of_lecture7_mean_and_variance_per_tile(load_scene):
    s = load_scene("lecture7_fixed.qdmg")               # glossiness 0.9..., 25 samples at depth 0
    old = s.settings.want_aa
    s.settings.want_aa = 0
    try:
        _check(*_noise_stats(s, (0, 0, s.settings.frame_width, s.settings.frame_height)))
    finally:
        s.settings.want_aa = old


def test_glossy_floor_of_the_dragon_scene_mean_and_variance_per_tile(load_scene):
    s = load_scene("hw9/dragon.qdmg")
    old = s.settings.want_aa
    s.settings.want_aa = 0
    try:
        W, H = s.settings.frame_width, s.settings.frame_height
        _check(*_noise_stats(s, (0, H - 48, 160, H)))      # a strip of the glossy floor in the foreground
    finally:
        s.settings.want_aa = old


def test_hashed_stream_is_uniform_and_uncorrelated():
    """random01(path, sample, draw) over the ids the shader really uses: path ids of a 1000 x 1000 frame's camera rays,
    sample slot 0, draws 0 (angle) and 1 (radius)."""
    ys, xs = np.meshgrid(np.arange(1000), np.arange(1000), indexing="ij")
    xy = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    n = len(xy)
    path = np.zeros(n, np.uint32)
    oracle.lib().orc_pixel_paths(n, xy.ctypes.data, path.ctypes.data)
    assert len(np.unique(path)) > 0.999 * n                          # path ids of neighbouring pixels do not collide
    draws = []
    for sample, draw in [(0, 0), (0, 1), (1, 0), (24, 1)]:
        out = np.zeros(n)
        samples, draw_ids = np.full(n, sample, np.uint32), np.full(n, draw, np.uint32)
        oracle.lib().orc_random01_bulk(n, path.ctypes.data, samples.ctypes.data, draw_ids.ctypes.data, out.ctypes.data)
        assert 0.0 <= out.min() and out.max() <= 1.0
        hist, _ = np.histogram(out, bins=64, range=(0, 1))
        chi2 = ((hist - n / 64) ** 2 / (n / 64)).sum()              # 63 degrees of freedom: mean 63, sd 11.2
        assert chi2 < 63 + 5 * 11.3, chi2
        assert abs(out.mean() - 0.5) < 5 * np.sqrt(1 / 12 / n)
        draws.append(out)
    for i in range(len(draws)):
        for j in range(i + 1, len(draws)):
            assert abs(np.corrcoef(draws[i], draws[j])[0, 1]) < 5 / np.sqrt(n)       # streams of different draws / samples
    img = draws[0].reshape(1000, 1000)
    assert abs(np.corrcoef(img[:, :-1].ravel(), img[:, 1:].ravel())[0, 1]) < 5 / np.sqrt(n)    # neighbouring pixels
    assert abs(np.corrcoef(img[:-1].ravel(), img[1:].ravel())[0, 1]) < 5 / np.sqrt(n)
