"""Condense the SQ-counter passes of tools/conv_one.py into profiles/<tag>_sq_counters_conv.json.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
            SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/sq_<H>_<W>_<cin>_<cout>_<cfg> \\
            -- python3 tools/conv_one.py H W cin cout cfg 20
  python tools/summarize_sq.py r01 gpurun_out/sq_*
"""
import collections, csv, glob, json, os, sys
tag, dirs = sys.argv[1], sys.argv[2:]
out = {"note": "rocprofv3 --pmc (8 SQ counters, one pass) around tools/conv_one.py H W cin cout cfg 20; per-launch averages "
               "over the conv_igemm launches.  SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per v_mfma_f32_32x32x16_bf16 summed "
               "over all waves; SQ_BUSY_CYCLES is summed over 32 shader engines, so kernel cycles = BUSY/32 and "
               "mfma_pipe_busy_frac = MFMA_BUSY / (1024 SIMDs x kernel cycles).  WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY "
               "are fractions of SQ_WAVE_CYCLES.", "kernels": {}}
for d in dirs:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = {}                                  # kernel name -> (sums, counts): a pass may hold several kernels (ws128_probe.py)
    for r in csv.DictReader(open(f[-1])):
        if "conv_igemm" not in r["Kernel_Name"] and "conv_ws" not in r["Kernel_Name"]:
            continue
        agg, n = per.setdefault(r["Kernel_Name"], (collections.defaultdict(float), collections.defaultdict(int)))
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
    for name, (agg, n) in per.items():
        raw = {k: int(v / n[k]) for k, v in agg.items()}
        cyc = raw["SQ_BUSY_CYCLES"] / 32
        short = name[name.index("Cfg<"):name.index(">(") - 1 if ">(" in name else len(name)] if "Cfg<" in name else name[name.index("conv_ws"):name.index("(", name.index("conv_ws"))]
        key = os.path.basename(d.rstrip("/"))[3:] + " " + short
        out["kernels"][key] = {
            "launches": max(n.values()),
            "wait_any_frac_of_wave_cycles": round(raw["SQ_WAIT_ANY"] / raw["SQ_WAVE_CYCLES"], 3),
            "wait_inst_any_frac": round(raw["SQ_WAIT_INST_ANY"] / raw["SQ_WAVE_CYCLES"], 3),
            "active_inst_any_frac": round(raw["SQ_ACTIVE_INST_ANY"] / raw["SQ_WAVE_CYCLES"], 3),
            "lds_bank_conflict_frac_of_lds_cycles": round(raw["SQ_LDS_BANK_CONFLICT"] / max(1, raw["SQ_LDS_IDX_ACTIVE"]), 4),
            "raw": raw, "kernel_cycles_est": int(cyc),
            "mfma_pipe_busy_frac": round(raw["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 3)}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_sq_counters_conv.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, "mfma busy", v["mfma_pipe_busy_frac"], "wait_any", v["wait_any_frac_of_wave_cycles"])
