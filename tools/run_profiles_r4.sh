# Round-4 profiles (run on the GPU box through gpurun; every bench.py run under the profiler with --no-cpu-baseline: the
# baseline's worker processes would start under the profiler's preloaded tool too).  Summaries are copied to profiles/r4_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p4
# 1. kernel trace of the default bench (workload c3 at N = 1), one warm-up + one timed step
rocprofv3 --kernel-trace -d gpurun_out/p4/c3 -o c -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 > gpurun_out/p4/c3.json 2> gpurun_out/p4/c3.err
python tools/rocpd_stats.py $(find gpurun_out/p4/c3 -name "*.db" | tail -1) --csv gpurun_out/p4/c3_kernel_stats.csv > gpurun_out/p4/c3_kstats.txt
# 2. one pipeline's share of it (2500 records, one stream: launch durations are not inflated by concurrent launches)
rocprofv3 --kernel-trace -d gpurun_out/p4/b1 -o b -- python3 tools/trace_batch.py 2500 > gpurun_out/p4/b1.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/p4/b1 -name "*.db" | tail -1) --csv gpurun_out/p4/b1_kernel_stats.csv > gpurun_out/p4/b1_kstats.txt
# 3. the single-record latency path (bench --workload c1)
rocprofv3 --kernel-trace -d gpurun_out/p4/c1 -o s -- python3 bench.py --workload c1 --no-cpu-baseline --no-batched --no-eval-many > gpurun_out/p4/c1.json 2> gpurun_out/p4/c1.err
python tools/rocpd_stats.py $(find gpurun_out/p4/c1 -name "*.db" | tail -1) --csv gpurun_out/p4/c1_kernel_stats.csv --step 2 > gpurun_out/p4/c1_kernel_trace.txt
# 4. configs[4] order (N = 1152), 8 records
rocprofv3 --kernel-trace -d gpurun_out/p4/c5 -o k -- python3 tools/perf_c5.py 8 > gpurun_out/p4/c5.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/p4/c5 -name "*.db" | tail -1) --csv gpurun_out/p4/c5_kernel_stats.csv > gpurun_out/p4/c5_kstats.txt
find gpurun_out/p4 -name "*.db" -delete
# 5. stamped build: where k_brent_warm spends its cycles (1000 records)
python3 tools/exp_brent_stamps.py 1000 2>&1 | tail -14 > gpurun_out/p4/brent_stamps.txt
# (K3 and K3p are the kernels of round 3, bit for bit: their PMC passes are profiles/r3_cold_solve_pmc.json)
