#!/usr/bin/env python3
"""profiles/hbm_traffic.json[key] <- a PMC summary (scripts/pmc_summary.py) of the CURRENT libqdtrace.so: HBM bytes per frame,
wave-instructions per frame, lane utilisation, L2 hit rate, with the sha256 of the device sources so that bench.py can tell a stale entry.
usage: scripts/update_traffic.py <key> <pmc_summary.json (under profiles/)> <build tag>"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key, summary, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def source_sha():
    """bench.py's lib_sha256(): sha256 over the device sources (the same for every rebuild of the same tree)."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "raytracing_amd", "csrc", "device")
    for f in sorted(os.listdir(src)) + [os.path.join(ROOT, "include", "qdtrace.h")]:
        p = f if os.path.isabs(f) else os.path.join(src, f)
        if os.path.isfile(p):
            h.update(os.path.basename(p).encode()); h.update(open(p, "rb").read())
    return h.hexdigest()

d = json.load(open(summary))
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
kinds = {"valu": "SQ_INSTS_VALU", "salu": "SQ_INSTS_SALU", "vmem": "SQ_INSTS_VMEM", "lds": "SQ_INSTS_LDS"}
insts = {k: sum(g.get(c, 0.0) for g in d["kernels"].values()) for k, c in kinds.items()}
insts["total"] = sum(insts.values())
act = sum(g.get("SQ_ACTIVE_INST_VALU", 0.0) for g in d["kernels"].values())
thr = sum(g.get("SQ_THREAD_CYCLES_VALU", 0.0) for g in d["kernels"].values())
hit = sum(g.get("TCC_HIT_sum", 0.0) for g in d["kernels"].values()); miss = sum(g.get("TCC_MISS_sum", 0.0) for g in d["kernels"].values())
h = d["hbm_bytes_per_frame"]
table[key] = {
    "bytes_per_frame": h["total"], "read": h["read"], "write": h["write"],
    "wave_insts_per_frame": insts, "valu_lane_util": thr / (64 * act) if act else None, "l2_hit": hit / (hit + miss) if hit + miss else None,
    "build": tag, "lib_sha256": source_sha(),
    "source": f"{os.path.relpath(summary, ROOT)}: per frame, summed over the frame's launches of {' + '.join(d['kernels'])} (frames one after the other), "
              "(2 x FETCH_SIZE + WRITE_SIZE) KiB; reads doubled per MI355X_MICROARCH.md (gfx950 tallies a 128-byte request as 64 B)",
}
json.dump(table, open(path, "w"), indent=1)
print(key, json.dumps(table[key])[:300])
