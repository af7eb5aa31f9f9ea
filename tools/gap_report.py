"""Idle time between consecutive kernels of the replayed step, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps_S -- python3 bench.py --no-cpu-baseline --no-extra --size S --steps 20 --warmup 5
    python tools/gap_report.py gpurun_out/gaps_S [first_kernel_substring]

Takes occurrences of the step's first kernel as step boundaries (the last 10 steps of the most common length) and prints, per position in
the step, kernel name, mean duration and the mean idle time before it started.
"""
import csv, glob, sys, collections

def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]

def main():
    d = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "conv_first_fwd"
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:]) if b - a > 10]     # (per-op rating loops repeat one kernel)
    n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if len(s) == n][-10:]
    print(f"{len(steps)} steps of {n} kernels")
    tot_d = tot_g = 0.0
    for k in range(n):
        dur = sum(s[k][1] - s[k][0] for s in steps) / len(steps) / 1e3
        gap = sum((s[k][0] - s[k - 1][1]) for s in steps) / len(steps) / 1e3 if k else 0.0
        tot_d += dur; tot_g += gap
        print(f"{k:3d} {short(steps[0][k][2]):72s} dur {dur:8.2f} us   idle before {gap:6.2f} us")
    wall = sum(s[-1][1] - s[0][0] for s in steps) / len(steps) / 1e3
    print(f"sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us, first start -> last end {wall:.1f} us")

if __name__ == "__main__":
    main()
