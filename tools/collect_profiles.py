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


def short(name):
    m = re.search(r"conv_igemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E", name)          # mangled: <T, NF, WN, MODE>
    if m:
        return "conv_igemm_kernel<%s,NF=%s,WN=%s,MODE=%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2), m.group(3), m.group(4))
    m = re.search(r"conv_igemm_group_kernelI(DF16b|f)Li(\d+)E", name)                      # mangled: <T, NFW>
    if m:
        return "conv_igemm_group_kernel<%s,NFW=%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2))
    m = re.search(r"conv_igemm_kernel<([^>]*)>", name)
    if m:
        return "conv_igemm_kernel<" + m.group(1).replace("__hip_bfloat16", "bf16").replace(" ", "") + ">"
    m = re.search(r"(conv1x1_dma_kernel|stem_fwd_u8_kernel)(?:<([^>]*)>|I([A-Za-z0-9_]*?)E)?", name)
    if m:
        return m.group(1) + ("<%s>" % m.group(2).replace(" ", "") if m.group(2) else "")
    m = re.search(r"(stem_delta_grad_kernel|stem_mask_kernel|stem_delta_bias_kernel|conv_splitk_finish_kernel|maxpool_\w+|head_\w+|apply_s2d_\w+|"
                  r"grad_reduce_\w+|softmax_adv_loss_kernel|reg_adam_kernel|dense_\w+)", name)
    if m:
        return m.group(1)
    return name.split("(")[0].strip() or name[:40]


rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(one("sq/**/*_counter_collection.csv"))):
    k = (r["Dispatch_Id"], short(r["Kernel_Name"]))
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    rows[k]["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for (_, name), c in rows.items():
    agg[name]["launches"] += 1
    for key, v in c.items():
        agg[name][key] += v
out = {"state": state,
       "note": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
               "SQ_BUSY_CYCLES, FLK_SINGLE_STREAM=1, bench.py --steps 2 --warmup 1; eff_clock = GRBM_GUI_ACTIVE/8/duration (8 XCDs); "
               "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); wait_* = fraction of SQ_WAVE_CYCLES", "kernels": {}}
for name, c in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    if c["ns"] < 2e4:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    out["kernels"][name] = {"launches": int(c["launches"]), "ms": round(c["ns"] / 1e6, 4), "eff_clock_GHz": round(cyc / c["ns"], 3),
                            "wait_any": round(c["SQ_WAIT_ANY"] / wc, 3), "wait_inst": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                            "active": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3),
                            "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(cyc * 1024, 1.0), 3)}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_sq_summary.json"), "w"), indent=1)
print("wrote profiles/%s_* at state %s" % (tag, state))
