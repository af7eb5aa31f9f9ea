"""Condense rocprofv3 output (gpurun_out/<dir>) into small, committed files under profiles/.

usage: python tools/summarize_profile.py <round-tag> <stats_dir_512> <stats_dir_1024> [<pmc_fetch_dir> <pmc_write_dir>]
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)
for size, d in (("512", sys.argv[2]), ("1024", sys.argv[3])):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats_{size}.csv"))
if len(sys.argv) > 5:
    summary = {"note": "rocprofv3 --pmc, bench.py --size 1024 --steps 3 --warmup 2; per-launch averages in KiB as "
                       "reported; FETCH_SIZE on gfx950 counts 128-B requests as 64 B (MI355X_MICROARCH.md §HBM), "
                       "so hbm_read_bytes = 2 * FETCH_SIZE * 1024; hbm_write_bytes = WRITE_SIZE * 1024.",
               "kernels": {}}
    for key, d in (("FETCH_SIZE", sys.argv[4]), ("WRITE_SIZE", sys.argv[5])):
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != key:
                continue
            k = r["Kernel_Name"]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            if "anonymous namespace" not in k:
                continue
            e = summary["kernels"].setdefault(k, {})
            e[key + "_avg_KiB"] = round(v / n, 1)
            e[key + "_launches"] = n
    for k, e in summary["kernels"].items():
        rd = 2 * e.get("FETCH_SIZE_avg_KiB", 0.0) * 1024
        wr = e.get("WRITE_SIZE_avg_KiB", 0.0) * 1024
        e["hbm_bytes_per_launch"] = int(rd + wr)
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc_hbm_1024.json"), "w"), indent=1)
print(os.listdir(out))
