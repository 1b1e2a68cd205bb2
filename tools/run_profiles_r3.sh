# (every bench.py run under the profiler with --no-cpu-baseline: the baseline's worker processes start under the profiler's
# preloaded tool too, and a run hung there once)
# Round-3 profiles: kernel traces of the bench step and of workload c3, PMC passes (HBM bytes) of the resident-basis evaluation.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p
rocprofv3 --kernel-trace -d gpurun_out/p/bench -o b -- python3 bench.py --no-cpu-baseline --no-batched > gpurun_out/p/bench.json 2> gpurun_out/p/bench.err
python tools/rocpd_stats.py $(find gpurun_out/p/bench -name "*.db" | tail -1) --csv gpurun_out/p/bench_kernel_stats.csv --step 2 > gpurun_out/p/bench_kernel_trace.txt
rocprofv3 --kernel-trace -d gpurun_out/p/c3 -o c -- python3 bench.py --workload c3 --records 2000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/p/c3.json 2> gpurun_out/p/c3.err
python tools/rocpd_stats.py $(find gpurun_out/p/c3 -name "*.db" | tail -1) --csv gpurun_out/p/c3_kernel_stats.csv > gpurun_out/p/c3_kstats.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p/pmc_f -o f -- python3 tools/perf_eval_resident.py > gpurun_out/p/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p/pmc_w -o w -- python3 tools/perf_eval_resident.py > gpurun_out/p/pmc_w.log 2>&1
for f in $(find gpurun_out/p/pmc_f gpurun_out/p/pmc_w -name "*counter_collection.csv"); do python tools/pmc_sum.py $f k_eval_resident; done > gpurun_out/p/pmc_resident.txt
python3 tools/exp_hull_phases.py 2>&1 | tail -5 > gpurun_out/p/hull_phases.txt
python3 tools/exp_brent_stamps.py 1000 2>&1 | tail -14 > gpurun_out/p/brent_stamps.txt
rm -rf gpurun_out/p/bench/*/*.db gpurun_out/p/c3/*/*.db 2>/dev/null; find gpurun_out/p -name "*.db" -size +20M -delete
