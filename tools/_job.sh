# scratch: the command of the last gpurun call of a session (the kept job scripts are tools/_gpu_job_*.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_jacobian.py -m gpu -x -q 2>&1 | grep -a "passed\|failed\|error" | tail -2
timeout 200 python tests/fuzz_parity.py --gpu --jac --cases 800 --seed 121 2>&1 | tail -1
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0.2 --only-extras pc 2>&1 | grep -a "PC matrix"
