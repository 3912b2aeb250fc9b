# kernel traces of single extras of bench.py: EXTRAS="matvec config3 config2" (one rocprofv3 run each), TAG
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r03_x}
for E in ${EXTRAS:-matvec config3 config2}; do
  B="python bench.py --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0.2 --only-extras $E ${BENCH_EXTRA}"
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_${E}.json 2> $O/${TAG}_${E}.log
  grep -a "ms\b\|ms/" $O/${TAG}_${E}.log | tail -4
  python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_${E}_trace.md "($TAG, git ${GIT}: $B)" | cut -c1-150 | head -${ROWS:-30}
  rm -rf $O/prof
done
