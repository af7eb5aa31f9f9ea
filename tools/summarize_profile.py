"""Condense rocprofv3 output (gpurun_out/<prefix>_*) into small, committed files under profiles/.

usage: python tools/summarize_profile.py <tag> <prefix>
  expects <prefix>_s{512,1024}  : rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py ...
          <prefix>_f{512,1024}  : rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 2
          <prefix>_w{512,1024}  : the same with --pmc WRITE_SIZE  (the two counters do not fit one pass)
writes   profiles/<tag>_kernel_stats_{512,1024}.csv and profiles/<tag>_pmc_hbm_{512,1024}.json
"""
import collections, csv, glob, json, os, shutil, sys

tag, prefix = sys.argv[1], sys.argv[2]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern, d):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return hits[0] if hits else None


for size in ("512", "1024"):
    f = one("*_kernel_stats.csv", f"{prefix}_s{size}")
    if f:
        shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats_{size}.csv"))
    dirs = {"FETCH_SIZE": f"{prefix}_f{size}", "WRITE_SIZE": f"{prefix}_w{size}"}
    if not all(os.path.isdir(d) for d in dirs.values()):
        continue
    summary = {"note": f"rocprofv3 --pmc <counter> --kernel-trace, bench.py --size {size} --steps 3 --warmup 2, one counter "
                       "per pass; per-launch averages in KiB as reported (includes the tile-autotune launches of the same "
                       "kernel on scratch data).  FETCH_SIZE on gfx950 tallies 128-B requests at 64 B "
                       "(MI355X_MICROARCH.md, HBM): hbm_read_bytes = 2 * FETCH_SIZE * 1024; hbm_write_bytes = "
                       "WRITE_SIZE * 1024.  Infinity-Cache hits are counted, not excluded.",
               "kernels": {}}
    for key, d in dirs.items():
        f = one("*_counter_collection.csv", d)
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != key:
                continue
            agg[r["Kernel_Name"]][0] += 1
            agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            if "anonymous namespace" not in k:
                continue
            e = summary["kernels"].setdefault(k, {})
            e[key + "_avg_KiB"] = round(v / n, 1)
            e[key + "_launches"] = n
    for e in summary["kernels"].values():
        e["hbm_bytes_per_launch"] = int(2 * e.get("FETCH_SIZE_avg_KiB", 0.0) * 1024 + e.get("WRITE_SIZE_avg_KiB", 0.0) * 1024)
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc_hbm_{size}.json"), "w"), indent=1)
print(sorted(os.listdir(out)))
