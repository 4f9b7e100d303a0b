#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel name from pmc*.csv files."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for fp in sorted(glob.glob(os.path.join(out, 'pmc*.csv'))):
    with open(fp) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '')
            short = name.split('(')[0][-60:]
            agg[short][row['Counter_Name']].append(float(row['Counter_Value']))
for kern, ctrs in agg.items():
    print(f'== {kern}')
    for c, v in sorted(ctrs.items()):
        print(f'   {c:28s} n={len(v):4d}  mean={sum(v) / len(v):16.1f}')
