cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01l
O=gpurun_out/r01l
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['mg'])"
TDB=$(find $O/trace -name "*.db" | head -1)
timeout 60 python tools/rocpd_summary.py $TDB $O/kernel_trace.md "(bench.py --steps 20 --warmup 3 --no-cpu-baseline)" > /dev/null
rm -rf $O/trace
grep "res_averaging\|ra_rfl\|stage_update\|march" $O/kernel_trace.md | cut -c1-150
timeout 300 python bench.py --workload rans_sa_jst_8x128x128x96 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_rans.json 2> $O/bench_rans.err
python -c "import json;d=json.loads(open('$O/bench_rans.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['mg'])"
