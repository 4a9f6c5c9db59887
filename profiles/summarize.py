#!/usr/bin/env python
"""Condenses a rocprofv3 CSV output directory into a small text summary (runs on the GPU box; the raw traces are
too large to copy back). Usage: python profiles/summarize.py <rocprof_out_dir> <summary.md> [--pmc]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace('distr::', '')
    return name[:70]


def main():
    d, out = sys.argv[1], sys.argv[2]
    lines = []
    for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)):
        lines.append('## kernel stats (%s)\n' % os.path.basename(f))
        lines.append('| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|')
        for r in csv.DictReader(open(f)):
            g = lambda k: float(r.get(k, 0) or 0)
            lines.append('| %s | %s | %.3f | %.1f | %.1f | %.1f | %.2f |' % (short(r['Name']), r['Calls'], g('TotalDurationNs') / 1e6,
                         g('AverageNs') / 1e3, g('MinNs') / 1e3, g('MaxNs') / 1e3, g('Percentage')))
        lines.append('')
    for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)):
        # per-launch durations of the march kernel in launch order (first forward only): shows the live-ray tail
        rows = [r for r in csv.DictReader(open(f))]
        rows.sort(key=lambda r: int(r['Start_Timestamp']))
        march = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size', ''), r.get('VGPR_Count', ''),
                  r.get('Accum_VGPR_Count', ''), r.get('LDS_Block_Size', ''), r.get('Scratch_Size', '')) for r in rows if 'k_march' in r['Kernel_Name'] or 'k_step' in r['Kernel_Name'] or 'k_tail' in r['Kernel_Name']]
        if march:
            lines.append('## k_march-family launches of the last forward in launch order: <tile size>:<us> (16/32/64-ray kernels; c = coarse level)\n')
            def tag(name):
                if 'k_step' in name: return 'step'
                if 'k_tail' in name: return 'tail'
                if 'k_march16' in name: return '16'
                if 'k_march<1' in name: return 'c'
                return '32' if ', 1, ' in name else '64'
            # last forward = everything after the last coarse-level launch block
            idx = [i for i, m in enumerate(march) if 'k_march<1' in m[0]]
            start = 0
            if idx:
                j = idx[-1]
                while j > 0 and 'k_march<1' in march[j - 1][0]: j -= 1
                start = j
            last = march[start:]
            lines.append(' '.join('%s:%.0f' % (tag(m[0]), m[1]) for m in last))
            lines.append('')
        tot = defaultdict(float)
        t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
        for r in rows:
            tot[short(r['Kernel_Name'])] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        busy = sum(tot.values())
        lines.append('kernel-busy %.1f ms of %.1f ms wall between first and last kernel (%d dispatches)\n' % (busy, (t1 - t0) / 1e6, len(rows)))
    for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        lines.append('## PMC counters per dispatch (%s)\n' % os.path.basename(f))
        lines.append('| kernel | counter | dispatches | sum | mean per dispatch | max |\n|---|---|---|---|---|---|')
        for k in sorted(acc):
            for cn in sorted(acc[k]):
                v = acc[k][cn]
                lines.append('| %s | %s | %d | %.6g | %.6g | %.6g |' % (k, cn, len(v), sum(v), sum(v) / len(v), max(v)))
        lines.append('')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:60]))


if __name__ == '__main__':
    main()
