cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/p/c3 -o c -- python3 bench.py --workload c3 --records 2000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/p/c3.json 2> gpurun_out/p/c3.err
echo c3 done rc=$?
python tools/rocpd_stats.py $(find gpurun_out/p/c3 -name "*.db" | tail -1) --csv gpurun_out/p/c3_kernel_stats.csv > gpurun_out/p/c3_kstats.txt
timeout -k 10 120 python3 tools/exp_hull_phases.py 2>&1 | tail -5 > gpurun_out/p/hull_phases.txt
timeout -k 10 120 python3 tools/exp_brent_stamps.py 1000 2>&1 | tail -14 > gpurun_out/p/brent_stamps.txt
echo stamps done
timeout -k 10 300 bash tools/run_fitprof.sh
echo fitprof done
find gpurun_out/p -name "*.db" -size +20M -delete
