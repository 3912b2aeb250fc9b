cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01h
O=gpurun_out/r01h
timeout 300 python -m pytest tests/test_gpu_rans.py tests/test_gpu_smoothers.py -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for t in 0 1; do
timeout 200 python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-mg --no-cpu-baseline --tuning viscous_tiled=$t > $O/bench_rans_t$t.json 2> $O/bench_rans_t$t.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$O/bench_rans_t$t.json').read().strip().splitlines()[-1]);print($t, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-mg --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err; echo "trace rc=$?"
TDB=$(find $O/trace -name "*.db" | head -1)
timeout 60 python tools/rocpd_summary.py $TDB $O/kernel_trace.md "(bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-mg --no-cpu-baseline)" > /dev/null
rm -rf $O/trace
head -14 $O/kernel_trace.md | cut -c1-180
