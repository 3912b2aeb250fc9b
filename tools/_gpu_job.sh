cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 1. default bench (headline, with MG section and cpu baseline)
timeout 600 python bench.py 2>gpurun_out/bench_default.err | tail -1 > gpurun_out/r01_o_bench.json; cat gpurun_out/r01_o_bench.json | cut -c1-600
# 2. kernel trace of the same command (no cpu baseline: the reference's threads would only add noise to the trace)
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_euler -o euler -- python bench.py --no-cpu-baseline > gpurun_out/bench_euler_prof.log 2>&1
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_euler/euler_results.db gpurun_out/r01_o_bench_euler_kernel_trace.md "(bench.py --no-cpu-baseline)" | head -12
# 3. RANS config 3
timeout 300 python bench.py --workload rans_sa_jst_8x128x128x96 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01_o_bench_rans_jst.json; cut -c1-300 gpurun_out/r01_o_bench_rans_jst.json
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rans3 -o rans -- python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_rans_prof.log 2>&1
timeout 60 python tools/rocpd_summary.py gpurun_out/prof_rans3/rans_results.db gpurun_out/r01_o_bench_rans_config3_kernel_trace.md "(bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline)" | head -8
