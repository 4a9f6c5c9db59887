"""Diagnostic (not a test): per-launch durations of the march kernels of the last forward of a bench.py run from a
rocprofv3 CSV trace directory (argv[1]); s = merged step (k_step), m = k_march, c = k_march16 (16-ray / cluster)."""
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_step' in r['Kernel_Name'] or 'k_march' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 's' if 'k_step' in r['Kernel_Name'] else ('c' if 'k_march16' in r['Kernel_Name'] else 'm')) for r in rows]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
last = d[-n:]
print(' '.join('%s%.0f' % (t, x) for x, t in last))
print('sum: %.1f us' % sum(x for x, _ in last))
