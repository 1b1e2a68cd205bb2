# Round-4 profiles (run on the GPU box through gpurun): kernel trace of the default bench (workload c3 at N = 1).
# (every bench.py run under the profiler with --no-cpu-baseline: the baseline's worker processes would start under the profiler's
# preloaded tool too)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p4
rocprofv3 --kernel-trace -d gpurun_out/p4/c3 -o c -- python3 bench.py --no-cpu-baseline --no-secondary --steps 1 --warmup 1 > gpurun_out/p4/c3.json 2> gpurun_out/p4/c3.err
python tools/rocpd_stats.py $(find gpurun_out/p4/c3 -name "*.db" | tail -1) --csv gpurun_out/p4/c3_kernel_stats.csv > gpurun_out/p4/c3_kstats.txt
find gpurun_out/p4 -name "*.db" -size +20M -delete
