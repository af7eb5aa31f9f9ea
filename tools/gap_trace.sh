set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gapc -- python3 tools/gap_probe.py 512 40 > $R/gpurun_out/gapc.log 2> $R/gpurun_out/gapc.err
python - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = sorted(glob.glob(R + "/gpurun_out/gapc/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# gaps in front of every pass_a launch, in order of appearance (mode A: 120+20+40, B: 20+40, C: 1+1+20+40, D: 20+40)
gaps = [(rows[i][0] - rows[i - 1][1]) / 1e3 for i in range(1, len(rows)) if "pass_a" in rows[i][2]]
prev = [rows[i - 1][2][:40] for i in range(1, len(rows)) if "pass_a" in rows[i][2]]
print(len(gaps), "pass_a launches")
for a, b, name in ((120, 180, "A eager after graph"), (200, 240, "B own graph"), (262, 302, "C one graph"), (322, 362, "D all eager")):
    g = gaps[a:b]
    print(name, "mean gap before pass_a %.2f us  (min %.2f max %.2f)  prev kernel %s" % (sum(g) / len(g), min(g), max(g), prev[a]))
PY
find $R/gpurun_out/gapc -name "*kernel_trace.csv" -delete || true
