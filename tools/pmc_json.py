#!/usr/bin/env python3
"""The counter_collection.csv files of rocprofv3 --pmc passes over one target -> one JSON: per kernel the mean per launch of
every counter and the shares derived from them (VALU active / waiting share of the wave cycles, LDS array busy share of the
kernel, bank conflicts per LDS cycle).   python tools/pmc_json.py out.json pass1.csv pass2.csv ... [--note text]"""
import collections
import csv
import json
import re
import sys

args = sys.argv[1:]
note = None
if '--note' in args:
    i = args.index('--note')
    note = args[i + 1]
    del args[i:i + 2]
out, files = args[0], args[1:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
        agg[m.group(1) if m else r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, cs in agg.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    der = {}
    if d.get('SQ_WAVE_CYCLES'):
        if 'SQ_ACTIVE_INST_VALU' in d:
            der['valu_active_share_of_wave_cycles'] = d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES']
        if 'SQ_WAIT_ANY' in d:
            der['waiting_share_of_wave_cycles'] = d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']
    if d.get('SQ_BUSY_CYCLES') and 'SQ_LDS_IDX_ACTIVE' in d:
        der['lds_array_busy_share_of_kernel'] = d['SQ_LDS_IDX_ACTIVE'] / d['SQ_BUSY_CYCLES']
    if d.get('SQ_LDS_IDX_ACTIVE') and 'SQ_LDS_BANK_CONFLICT' in d:
        der['lds_bank_conflict_share_of_lds_cycles'] = d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']
    d['_derived'] = der
    res[k] = d
if note:
    res['_note'] = note
json.dump(res, open(out, 'w'), indent=1)
for k, d in res.items():
    if k != '_note':
        print(k, {a: round(b, 4) for a, b in d['_derived'].items()}, 'busy cycles %.0f' % d.get('SQ_BUSY_CYCLES', float('nan')))
