cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01j
O=gpurun_out/r01j
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['mg'])"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --workload euler_jst_512x32 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_512.json 2> $O/bench_512.err; echo "bench512 rc=$?"
tail -3 $O/bench_512.err
python -c "import json;d=json.loads(open('$O/bench_512.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['mg'])"
TDB=$(find $O/trace -name "*.db" | head -1)
timeout 60 python tools/rocpd_summary.py $TDB $O/kernel_trace_512.md "(bench.py --workload euler_jst_512x32 --steps 20 --warmup 3 --no-cpu-baseline)" > /dev/null
rm -rf $O/trace
head -24 $O/kernel_trace_512.md | cut -c1-170
