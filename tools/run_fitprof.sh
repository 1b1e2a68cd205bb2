cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VINTERP_PIPELINES=1 VINTERP_STAGE_TIMES=1 VINTERP_TRACE=1 python3 tools/stats_fit1000.py > gpurun_out/fit1000_p1.log 2>&1
python3 tools/perf_fit.py 1000 > gpurun_out/fit1000_p4.log 2>&1
python3 tools/perf_fit.py 4000 >> gpurun_out/fit1000_p4.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/prof_fit -o f -- python3 tools/perf_fit.py 1000 > gpurun_out/fit1000_prof.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/prof_fit -name "*.db" | tail -1) --csv gpurun_out/fit1000_kernel_stats.csv > gpurun_out/fit1000_kstats.txt
