cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01g
O=gpurun_out/r01g
timeout 300 python -m pytest tests/test_gpu_smoothers.py tests/test_gpu_multigrid.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err; echo "trace rc=$?"
TDB=$(find $O/trace -name "*.db" | head -1)
timeout 60 python tools/rocpd_summary.py $TDB $O/kernel_trace.md "(bench.py --steps 20 --warmup 3 --no-cpu-baseline)" > /dev/null
rm -rf $O/trace
head -22 $O/kernel_trace.md | cut -c1-200
python -c "import json;d=json.loads(open('$O/bench_trace.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['mg'])"
