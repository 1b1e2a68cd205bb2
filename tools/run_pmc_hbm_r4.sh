#!/bin/bash
# Round 4: HBM bytes of the K3 kernels over a whole c3 step (10 000 records fitted): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# their own runs over bench.py (no CPU baseline, no secondary objects, 1 warm-up + 1 timed step), summed per kernel.
# Run from the repository root on the GPU box:  bash tools/run_pmc_hbm_r4.sh
set -e
ROOT=$(pwd)
mkdir -p gpurun_out/pmc4h
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc4h/$c -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 > $ROOT/gpurun_out/pmc4h/$c.json 2> $ROOT/gpurun_out/pmc4h/$c.err
  echo "$c done"
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('gpurun_out/pmc4h/%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r'(k_\w+(<[^>]*>)?|Cijk_\w{0,12}|__amd\w+)', r['Kernel_Name'])
            k = m.group(1) if m else r['Kernel_Name'][:40]
            tot[k][r['Counter_Name']] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE':
                n[k] += 1
print('kernel, launches (2 steps), FETCH_SIZE sum (KiB as reported), WRITE_SIZE sum (KiB as reported)')
for k in sorted(tot, key=lambda k: -tot[k].get('FETCH_SIZE', 0))[:14]:
    print('%-36s %7d %16.0f %16.0f' % (k, n[k], tot[k].get('FETCH_SIZE', 0), tot[k].get('WRITE_SIZE', 0)))
PY
