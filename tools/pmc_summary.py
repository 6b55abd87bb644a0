#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 CSV output: kernel-trace stats and PMC counter_collection files under a directory.
  python tools/pmc_summary.py <dir> [kernel-name regex]"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else '.')
for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)):
    print('==', f)
    for r in csv.DictReader(open(f)):
        if pat.search(r['Name']):
            print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_ns {float(r['AverageNs']):12.0f} pct {r['Percentage']}")
for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat.search(r['Kernel_Name']):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', f)
    for k, cs in acc.items():
        print('  ', k[:90])
        for c, v in cs.items():
            print(f'      {c:36s} n={len(v):4d} mean={sum(v) / len(v):.6g} last={v[-1]:.6g}')
