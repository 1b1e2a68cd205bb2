cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/p4
rocprofv3 --kernel-trace -d gpurun_out/p4/b1 -o b -- python3 tools/trace_batch.py 2500 > gpurun_out/p4/b1.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/p4/b1 -name "*.db" | tail -1) --csv gpurun_out/p4/b1_kernel_stats.csv > gpurun_out/p4/b1_kstats.txt
find gpurun_out/p4 -name "*.db" -delete
tail -3 gpurun_out/p4/b1.log
