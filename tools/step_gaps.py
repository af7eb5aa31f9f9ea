"""Idle time inside and BETWEEN the replayed steps, from a rocprofv3 kernel trace of tools/step_time.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sg -- python3 tools/step_time.py 512 100
    python tools/step_gaps.py gpurun_out/sg
Steps are cut at the first-layer forward kernel; the last 60 steps of the common length are averaged."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
starts = [i for i, r in enumerate(rows) if "conv_first_fwd" in r[2]]
steps = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if b - a > 10]
n = collections.Counter(b - a for a, b in steps).most_common(1)[0][0]
steps = [(a, b) for a, b in steps if b - a == n][-60:]
busy = inner = between = period = 0.0
big = collections.Counter()
for a, b in steps:
    ks = rows[a:b]
    busy += sum(e - s for s, e, _ in ks)
    for (s0, e0, _), (s1, e1, nm) in zip(ks[:-1], ks[1:]):
        g = max(0, s1 - e0)
        inner += g
        if g > 2000:
            big[nm.split("(")[0][-60:]] += g
    between += max(0, rows[b][0] - ks[-1][1])
    period += rows[b][0] - ks[0][0]
k = len(steps)
print(f"{d}: {k} steps of {n} kernels: period {period / k / 1e3:.1f} us = kernels {busy / k / 1e3:.1f} + idle inside the step {inner / k / 1e3:.1f} "
      f"+ idle between steps {between / k / 1e3:.1f} (kernels overlap where the sum exceeds the period)")
for nm, g in big.most_common(5):
    print(f"    idle before {nm}: {g / k / 1e3:.1f} us per step")
