"""Steady-state pass_a / pass_b durations from a rocprofv3 kernel trace directory (diagnostic)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
out = []
for tag in ('pass_a', 'pass_b', 'solve_kernel'):
    v = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows if tag in r['Kernel_Name']]
    out.append(f"{tag} {sum(v[-30:]) / 30:.1f}")
print(sys.argv[1], ' '.join(out))
