cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "jacobian or _ad or pc_" 2>&1 | tail -4
TAG=r05_y EXTRAS=pc ROWS=12 bash tools/_gpu_job_extras.sh
timeout 300 python tests/fuzz_parity.py --gpu --jac --cases 600 --seed 51 2>&1 | tail -1
