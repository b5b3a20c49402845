#!/usr/bin/env python3
"""Turn gpurun_out/prof23_<tag>/ (tools/profile_configs23.sh on the MI355X box) into tracked files:
  profiles/<tag>_<config>_kernel_stats.csv       rocprofv3 --kernel-trace --stats
  profiles/<tag>_<config>_pmc_sq_summary.json    SQ counters per kernel template (tools/pmc_sq.py)
for <config> in vrn_r2plus1d_18_bs1, vrn_r2plus1d_18_bs8, i3d_bs1.  Usage: collect_configs23.py <tag>"""
import glob, json, os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_sq

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"prof23_{tag}"), os.path.join(root, "profiles")
head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "flickering_adversarial_video_amd", "bench.py"],
                       capture_output=True, text=True).stdout.strip()
state = head + ("+uncommitted" if dirty else "")
args = {"vrn_r2plus1d_18_bs1": "--model r2plus1d_18 --batch 1 --frames 16", "vrn_r2plus1d_18_bs8": "--model r2plus1d_18 --batch 8 --frames 16",
        "i3d_bs1": "--batch 1 --frames 64"}
for cfg, ba in args.items():
    g = glob.glob(os.path.join(src, f"{cfg}_stats", "**", "*_kernel_stats.csv"), recursive=True)
    if not g:
        continue
    shutil.copy(max(g, key=os.path.getmtime), os.path.join(dst, f"{tag}_{cfg}_kernel_stats.csv"))
    c = max(glob.glob(os.path.join(src, f"{cfg}_sq", "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    s = pmc_sq.summarise(c, state, f"bench.py {ba} --steps 5 --warmup 2 (multi-stream plan, as timed)", min_ns=0)
    json.dump(s, open(os.path.join(dst, f"{tag}_{cfg}_pmc_sq_summary.json"), "w"), indent=1)
    tot = sum(k["ms"] for k in s["kernels"].values())
    print(cfg, "kernel time %.2f ms over the run;" % tot, "top:", [(n, k["us_per_launch"], k["mfma_busy_frac"], k["wait_any"]) for n, k in list(s["kernels"].items())[:4]])
print("wrote profiles/%s_* at state %s" % (tag, state))
