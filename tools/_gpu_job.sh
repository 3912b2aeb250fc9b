cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rans -o rans -- python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline --no-mg > gpurun_out/bench_rans.log 2>&1
tail -1 gpurun_out/bench_rans.log
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_rans/rans_results.db gpurun_out/rans_trace.md "(rans)" | head -12
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mg 2>&1 | tail -1
timeout 100 python tools/time_kernels.py 2>&1 | tail -3
