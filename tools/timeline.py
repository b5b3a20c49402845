#!/usr/bin/env python3
"""Concurrency / idle-gap analysis of one attack iteration from a rocprofv3 --kernel-trace CSV of bench.py (multi-stream mode).
Usage: timeline.py <kernel_trace.csv> [--list]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r["Queue_Id"])) for r in rows)
# a step ends with its reg_adam launch (the perturbation apply no longer marks its start: the plan launches one per half-batch)
ends = [i for i, e in enumerate(ev) if "reg_adam" in e[2]]
starts = [i + 1 for i in ends if i + 1 < len(ev)]


def short(n):
    if "conv_igemm" in n:
        return "conv"
    for k in ("conv1x1_dma", "stem_fwd_u8", "stem_delta_grad", "stem_mask", "wrun", "s1_tiled", "scatter", "strided_bwd", "maxpool_fwd", "head", "softmax", "reg_adam",
              "grad_reduce", "apply", "bias", "pack"):
        if k in n:
            return k
    return n[:24]


# the shortest span between consecutive step starts = a steady-state step (spans that cross the benchmark's barriers / read-backs are longer)
a, b = min(zip(starts[:-1], starts[1:]), key=lambda ab: ev[ab[1]][0] - ev[ab[0]][0])
step = ev[a:b]
t0 = step[0][0]
print("step: %.3f ms from its first kernel to the next step's first kernel, %d kernels" % ((ev[b][0] - t0) / 1e6, len(step)))
pts = sorted([(s, 1) for s, *_ in step] + [(e, -1) for _, e, *_ in step] + [(ev[b][0], 0)])
lvl, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[lvl] += t - last
    last, lvl = t, lvl + d
for k in sorted(hist):
    print("  %d kernels in flight: %.3f ms" % (k, hist[k] / 1e6))
cur_end, gaps = t0, []
for s, e, n, q in step:
    if s > cur_end + 2000:
        gaps.append(((s - cur_end) / 1e3, (s - t0) / 1e6, short(n)))
    cur_end = max(cur_end, e)
print("  idle gaps > 2 us: %d, %.1f us in total (mean %.1f us)" % (len(gaps), sum(g[0] for g in gaps), sum(g[0] for g in gaps) / max(len(gaps), 1)))
busy = collections.Counter()
for s, e, n, q in step:
    busy[short(n)] += e - s
print("  kernel time by kind (co-running kernels stretch each other): " + ", ".join("%s %.2f" % (k, v / 1e6) for k, v in busy.most_common()))
if "--list" in sys.argv:
    for s, e, n, q in step:
        print("  %7.3f %7.3f %6.1f us q%s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, q, short(n)))
