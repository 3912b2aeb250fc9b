# HBM-side bytes per kernel of single extras of bench.py: EXTRAS="config3 matvec ...", TAG, CELLS (cells of the workload)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r04_x}
for E in ${EXTRAS:-config3}; do
  B="python bench.py --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0.2 --only-extras $E ${BENCH_EXTRA}"
  timeout 500 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $B > /dev/null 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $B > /dev/null 2>&1
  C=10485760; if [ "$E" = "config2" ]; then C=16777216; fi; if [ "$E" = "small" ]; then C=11239424; fi
  python tools/pmc_kernel_bytes.py $O/pmc_f/f_results.db $O/pmc_w/w_results.db $C $O/${TAG}_${E}_pmc_bytes.md "($TAG, git ${GIT}: $B)" | cut -c1-160 | head -${ROWS:-24}
  rm -rf $O/pmc_f $O/pmc_w
done
