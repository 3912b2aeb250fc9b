cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01n
O=gpurun_out/r01n
timeout 300 python -m pytest tests/test_gpu_rans.py tests/test_gpu_smoothers.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rans.json 2> $O/bench_rans.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$O/bench_rans.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['mg']['ms_per_cycle'])"
TDB=$(find $O/trace -name "*.db" | head -1)
timeout 60 python tools/rocpd_summary.py $TDB $O/kernel_trace_rans.md "(bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline)" > /dev/null
rm -rf $O/trace
head -12 $O/kernel_trace_rans.md | cut -c1-150
