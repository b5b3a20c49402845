#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc pass with the SQ counters of tools/profile_round.sh / profile_configs23.sh:
effective clock, wave-cycle split (parked / issue-stalled / issuing) and MFMA busy fraction per kernel template.
Usage: pmc_sq.py <counter_collection.csv> <out.json> [state label] [command note]"""
import collections, csv, json, re, sys


def short(name):
    m = re.search(r"conv_igemm_kernelI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E", name)          # mangled: <T, NF, WN, MODE>
    if m:
        return "conv_igemm_kernel<%s,NF=%s,WN=%s,MODE=%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2), m.group(3), m.group(4))
    m = re.search(r"conv_igemm_group_kernelI(DF16b|f)Li(\d+)E", name)                      # mangled: <T, NFW>
    if m:
        return "conv_igemm_group_kernel<%s,NFW=%s>" % ("bf16" if m.group(1) == "DF16b" else "f32", m.group(2))
    m = re.search(r"(conv_igemm_kernel|conv_igemm_group_kernel|conv_pc_kernel|conv2p1_kernel)<([^>]*)>", name)
    if m:
        return m.group(1) + "<" + m.group(2).replace("__hip_bfloat16", "bf16").replace(" ", "") + ">"
    m = re.search(r"(conv1x1_dma_kernel|conv_t3_dma_kernel|stem_fwd_u8_kernel)(?:<([^>]*)>|I([A-Za-z0-9_]*?)E)?", name)
    if m:
        return m.group(1) + ("<%s>" % m.group(2).replace(" ", "") if m.group(2) else "")
    m = re.search(r"(stem_delta_grad_kernel|stem_mask_kernel|stem_delta_bias_kernel|conv_splitk_finish_kernel|maxpool_\w+|head_\w+|apply_s2d_\w+|"
                  r"grad_reduce_\w+|softmax_adv_loss_kernel|reg_adam_kernel|dense_\w+)", name)
    if m:
        return m.group(1)
    return name.split("(")[0].strip() or name[:40]


def summarise(path, state="unknown", note="", min_ns=2e4):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
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
                   "SQ_BUSY_CYCLES" + (", " + note if note else "") + "; eff_clock = GRBM_GUI_ACTIVE/8/duration (8 XCDs; reads high on "
                   "dispatches shorter than ~0.3 ms); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); wait_* = fraction of "
                   "SQ_WAVE_CYCLES", "kernels": {}}
    for name, c in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        if c["ns"] < min_ns:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8
        wc = max(c["SQ_WAVE_CYCLES"], 1.0)
        out["kernels"][name] = {"launches": int(c["launches"]), "ms": round(c["ns"] / 1e6, 4), "us_per_launch": round(c["ns"] / 1e3 / c["launches"], 2),
                                "eff_clock_GHz": round(cyc / c["ns"], 3),
                                "wait_any": round(c["SQ_WAIT_ANY"] / wc, 3), "wait_inst": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
                                "active": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3),
                                "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(cyc * 1024, 1.0), 3)}
    return out


if __name__ == "__main__":
    json.dump(summarise(sys.argv[1], *(sys.argv[3:5])), open(sys.argv[2], "w"), indent=1)
