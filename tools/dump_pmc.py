"""Per-kernel averages of every counter in rocprofv3 --pmc output directories: python tools/dump_pmc.py DIR..."""
import collections, csv, glob, os, sys
for d in sys.argv[1:]:
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(lambda: (collections.defaultdict(float), collections.defaultdict(int)))
    for r in csv.DictReader(open(f[-1])):
        agg, n = per[r["Kernel_Name"][:90]]
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
    for name, (agg, n) in per.items():
        if "conv" not in name:
            continue
        print(os.path.basename(d.rstrip("/")), name)
        print("   ", {k: int(v / n[k]) for k, v in sorted(agg.items())}, "launches", max(n.values()))
