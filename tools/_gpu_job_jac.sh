# GPU job: Jacobian parity tests + full GPU suite + bench with extras (PC assembly timing)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_r}
timeout 900 python -m pytest tests/test_gpu_jacobian.py -q 2>&1 | tail -30 | tee $O/${TAG}_pytest_jac.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/${TAG}_pytest.txt
timeout 900 python bench.py --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -c 1500 $O/${TAG}_bench.log; cut -c1-300 $O/${TAG}_bench.json
