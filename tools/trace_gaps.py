#!/usr/bin/env python3
"""Kernel timeline of the last steps from a rocprofv3 kernel-trace CSV."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -12:]
prev = None
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    print('%-60s dur %7.1f us  gap %6.1f us  lds %s' % (r['Kernel_Name'][:60], (e - s) / 1e3, gap, r.get('LDS_Block_Size', '')))
    prev = e
