cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "jacobian or _ad or pc_" 2>&1 | tail -4
TAG=r05_v EXTRAS=pc ROWS=16 bash tools/_gpu_job_extras.sh
