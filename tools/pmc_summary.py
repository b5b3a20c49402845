#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic (profiles/*.json).

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly 1/2 of the bytes of wide
coalesced streaming reads (16 B per lane: every global access of these kernels) -> doubled; WRITE_SIZE is exact for 16-byte
stores.  Usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [state label, e.g. the git commit]"""
import collections, csv, json, sys

def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        key = next((k for k in ("conv_pc_kernel", "conv_igemm_group_kernel", "conv_igemm_kernel", "conv1x1_dma_kernel", "conv_t3_dma_kernel", "stem_fwd_u8_kernel", "stem_delta_grad_kernel", "stem_mask_kernel") if k in n), None) \
            or n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
        d[key][0] += 1
        d[key][1] += float(r["Counter_Value"])
    return d

f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
out = {"state": sys.argv[4] if len(sys.argv) > 4 else "unknown",
       "note": "bytes = 2*FETCH_SIZE*1024 (gfx950 half-count correction) + WRITE_SIZE*1024; separate --pmc passes of "
               "`bench.py --steps 2 --warmup 1` (3 iterations + 1 label forward)", "kernels": {}}
for k in sorted(f, key=lambda k: -f[k][1]):
    launches = f[k][0]
    rd, wr = 2 * f[k][1] * 1024, w.get(k, [0, 0.0])[1] * 1024
    out["kernels"][k] = {"launches": launches, "read_bytes": rd, "write_bytes": wr, "bytes_per_launch": (rd + wr) / max(launches, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for kn in ("conv_pc_kernel", "conv_igemm_kernel", "conv_igemm_group_kernel"):
    if kn in out["kernels"]:
        c = out["kernels"][kn]
        print(kn + ":", c["launches"], "launches,", round(c["bytes_per_launch"] / 1e6, 1), "MB per launch")
