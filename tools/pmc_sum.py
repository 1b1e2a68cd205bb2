#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv per (kernel, counter): mean per launch."""
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[(r['Kernel_Name'].split('(')[0][-48:], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print('%-50s %-28s launches=%3d mean=%.4g' % (k, c, len(v), sum(v) / len(v)))
