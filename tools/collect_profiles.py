#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the MI355X box) into the tracked files profiles/<tag>_*.

  <tag>_bench_i3d_bs8_full.json      the default bench line (value, roofline, cpu_baseline, parity, other_configs)
  <tag>_per_layer_serial.txt         tools/profile_ops.py: every plan launch timed alone with HIP events
  <tag>_kernel_stats_{single,multi}_stream.csv   rocprofv3 --kernel-trace --stats
  <tag>_pmc_hbm_traffic.json         FETCH_SIZE / WRITE_SIZE passes (tools/pmc_summary.py: gfx950 corrections), state label = git HEAD
  <tag>_pmc_sq_summary.json          effective clock, wave-cycle split and MFMA busy fraction per kernel template
  <tag>_timeline.txt                 tools/timeline.py on the multi-stream kernel trace: kernels in flight, idle gaps, one step listed
Usage: collect_profiles.py <tag>"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"prof_{tag}"), os.path.join(root, "profiles")
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "flickering_adversarial_video_amd", "bench.py"],
                       capture_output=True, text=True).stdout.strip()
state = head + ("+uncommitted" if dirty else "")


def one(pattern):
    g = glob.glob(os.path.join(src, pattern), recursive=True)
    assert g, pattern
    return max(g, key=os.path.getmtime)      # gpurun merges into gpurun_out/ without deleting: an earlier run of the same tag may remain


line = [l for l in open(os.path.join(src, "bench_full.json")).read().splitlines() if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_bench_i3d_bs8_full.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "per_layer_serial.txt"), os.path.join(dst, f"{tag}_per_layer_serial.txt"))
for mode in ("single", "multi"):
    shutil.copy(one(f"{mode}/**/*_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats_{mode}_stream.csv"))
with open(os.path.join(dst, f"{tag}_timeline.txt"), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "timeline.py"), one("multi/**/*_kernel_trace.csv"), "--list"], stdout=f)
subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), one("fetch/**/*_counter_collection.csv"),
                       one("write/**/*_counter_collection.csv"), os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), state])


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_sq
out = pmc_sq.summarise(one("sq/**/*_counter_collection.csv"), state, "FLK_SINGLE_STREAM=1, bench.py --steps 2 --warmup 1")
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_sq_summary.json"), "w"), indent=1)
print("wrote profiles/%s_* at state %s" % (tag, state))
