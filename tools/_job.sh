cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; TAG=r04_a
timeout 900 python -m pytest tests/test_gpu_rans.py -m gpu -x -q -k "wall or six_physical or visc_gradient_fused or split" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | tee $O/${TAG}_pytest.txt
timeout 600 python bench.py --no-cpu-baseline --only-extras periodic > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; grep -a "ms" $O/${TAG}_bench.log | tail -8
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG: $B)" | head -30
rm -rf $O/prof
