"""qd_render_tiles_noalloc_async (the entry point that allocates nothing: working memory in a static __device__ array)
against qd_render_tiles_async, through the C-ABI: two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

from raytracing_amd import device, sharding

pytestmark = pytest.mark.gpu

SUM_ORDER_TOL = 2e-6       # colours are summed with float atomics in arrival order (same bar as tests/test_gpu_parity.py)


def _stats(t):
    return dict(zip(("rays", "primary_rays", "shadow_rays", "depth_killed"), t.cpu().tolist()[:4]))


def test_noalloc_entry_point_equals_render_tiles_async(load_scene):
    import torch
    device.init(0)
    s = load_scene("kdtree_teapot_lowres_env.qdmg")
    st = s.settings
    old = (st.frame_width, st.frame_height, st.want_aa, st.max_trace_depth)
    st.want_aa = 0
    ds = device.DeviceScene(s)
    try:
        tile = 32
        # first call: the whole frame at the scene's own size and depth; second call: a smaller frame, one rank's tiles of 3
        for (W, H, depth, phase, stride) in [(old[0], old[1], old[3], 0, 1), (640, 360, 4, 1, 3)]:
            st.frame_width, st.frame_height, st.max_trace_depth = W, H, depth
            tpr = sharding.tiles_per_rank_padded(W, H, tile, stride)
            ref = torch.zeros((tpr, tile, tile, 3), dtype=torch.float32, device="cuda:0")
            ref_stats = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            ds.render_tiles_async(ref.data_ptr(), tile, phase, stride, stats_dev_ptr=ref_stats.data_ptr())
            got = torch.full((tpr, tile, tile, 3), 7.0, dtype=torch.float32, device="cuda:0")      # stale contents must not survive
            got_stats = torch.zeros(8, dtype=torch.int64, device="cuda:0")
            ds.render_tiles_noalloc_async(got.data_ptr(), tile, phase, stride, stats_dev_ptr=got_stats.data_ptr())
            torch.cuda.synchronize()
            ds.check()
            n_owned = device.tiles_owned(W, H, tile, phase, stride)
            a, b = ref.cpu().numpy()[:n_owned], got.cpu().numpy()[:n_owned]
            assert np.abs(a).max() > 0.1
            assert np.abs(a - b).max() <= SUM_ORDER_TOL, float(np.abs(a - b).max())
            assert _stats(got_stats) == _stats(ref_stats)
    finally:
        ds.close()
        st.frame_width, st.frame_height, st.want_aa, st.max_trace_depth = old


def test_noalloc_entry_point_rejects_a_queue_that_does_not_fit(load_scene):
    device.init(0)
    s = load_scene("lecture7_fixed.qdmg")      # glossy sphere: 25 samples at depth 0
    st = s.settings
    old = st.max_trace_depth
    ds = device.DeviceScene(s)
    try:
        import torch
        buf = torch.zeros((sharding.tiles_per_rank_padded(st.frame_width, st.frame_height, 32, 1), 32, 32, 3), dtype=torch.float32, device="cuda:0")
        st.max_trace_depth = 60                # queue depth beyond what 96 MiB hold at one wave per CU
        with pytest.raises(device.DeviceError, match="static scratch"):
            ds.render_tiles_noalloc_async(buf.data_ptr(), 32, 0, 1)
    finally:
        st.max_trace_depth = old
        ds.close()
