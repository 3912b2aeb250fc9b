cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; TAG=r04_c
timeout 900 python -m pytest tests/test_gpu_rans.py tests/test_gpu_nk.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | tee $O/${TAG}_pytest.txt
timeout 600 python bench.py --no-cpu-baseline --only-extras pc,config3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; grep -a "ms" $O/${TAG}_bench.log | tail -8
for E in pc; do
  B="python bench.py --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0.2 --only-extras $E"
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_${E}.json 2> $O/${TAG}_${E}.log
  python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_${E}_trace.md "($TAG: $B)" | cut -c1-170 | head -30
  rm -rf $O/prof
done
