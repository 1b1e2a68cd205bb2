#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per (kernel, counter): mean per launch.
python tools/pmc_sum.py file.csv [kernel-name filter]"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name']
    m = re.search(r'(k_\w+(<[^>]*>)?)', name)
    agg[(m.group(1) if m else name[:48], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print('%-40s %-28s launches=%3d mean=%.6g' % (k, c, len(v), sum(v) / len(v)))
