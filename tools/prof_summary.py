"""Condense a rocprofv3 --kernel-trace CSV into a per-kernel table for the TIMED region of
bench.py (between the two lsn::selftest32_kernel marker dispatches it emits when
LSNET_PROF_MARKERS=1); falls back to the whole trace when there are no markers."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
files = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
if not files:
    print('no kernel_trace.csv under', d, glob.glob(d + '/**/*', recursive=True)[:20])
    sys.exit(1)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'selftest32_kernel' in r['Kernel_Name']]
if len(marks) >= 2:
    sel = rows[marks[-2] + 1:marks[-1]]
    region = 'timed region (between markers)'
else:
    sel, region = rows, 'whole trace (no markers)'
agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for r in sel:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg[r['Kernel_Name']]
    a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
tot = sum(a[1] for a in agg.values())
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e3 if sel else 0
lines = [f'# source: {files[0]}', f'# {region}: {len(sel)} dispatches, {steps} step(s)',
         f'# sum of kernel durations {tot / 1e3:.2f} ms ({tot / 1e3 / steps:.2f} ms/step); wall span {span / 1e3:.2f} ms '
         f'({span / 1e3 / steps:.2f} ms/step); GPU busy {100 * tot / max(span, 1e-9):.1f}%',
         f'{"ms/step":>9} {"pct":>6} {"calls/step":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9}  name']
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    lines.append(f'{a[1] / 1e3 / steps:9.3f} {100 * a[1] / tot:6.2f} {a[0] / steps:10.1f} {a[1] / a[0]:10.1f} {a[2]:9.1f} '
                 f'{a[3]:9.1f}  {name[:160]}')
# ---- where the GPU waits: idle time between consecutive dispatches of the timed region (union of busy intervals: kernels of
# the side stream overlap the launch stream's), by the kernel that FOLLOWS the gap ----
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in sel)
gaps = defaultdict(lambda: [0, 0.0])
idle, busy_end, hist = 0.0, ev[0][1] if ev else 0, [0, 0, 0, 0, 0]
big = []
for st, en, name in ev[1:]:
    if st > busy_end:
        g = (st - busy_end) / 1e3
        idle += g
        key = name.split('(')[0][:90]
        gaps[key][0] += 1
        gaps[key][1] += g
        hist[0 if g < 2 else 1 if g < 5 else 2 if g < 10 else 3 if g < 50 else 4] += 1
        if g >= 20:
            big.append((g, name[:100]))
    busy_end = max(busy_end, en)
lines.append('')
lines.append(f'# idle between dispatches: {idle / 1e3 / steps:.2f} ms/step in {sum(hist) / steps:.0f} gaps/step; gaps < 2 us: {hist[0] / steps:.0f}, 2 - 5: '
             f'{hist[1] / steps:.0f}, 5 - 10: {hist[2] / steps:.0f}, 10 - 50: {hist[3] / steps:.0f}, >= 50: {hist[4] / steps:.0f} per step '
             '(under the tracer: every dispatch pays its instrumentation)')
lines.append(f'{"idle ms/step":>13} {"gaps/step":>10} {"avg_us":>8}  kernel that follows the gap')
for name, g in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    lines.append(f'{g[1] / 1e3 / steps:13.3f} {g[0] / steps:10.1f} {g[1] / g[0]:8.1f}  {name}')
lines.append('# gaps of 20 us and more (what follows them):')
for g, name in sorted(big, reverse=True)[:15]:
    lines.append(f'{g:10.1f} us  {name}')
text = '\n'.join(lines)
print(text)
if out:
    open(out, 'w').write(text + '\n')
