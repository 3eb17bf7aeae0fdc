"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (mean of each counter over dispatches)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
if not files:
    print('no counter csv', glob.glob(d + '/**/*', recursive=True)[:30]); sys.exit(1)
agg = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    if 'lsn::' not in k:
        continue
    print(k[:90])
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} n={len(v):4d} mean={sum(v) / len(v):16.1f} max={max(v):16.1f}')
