cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_smoothers.py -m gpu -x -q -k "dadi" 2>&1 | tail -2
export EXTRAS=config3 TAG=r04_q ROWS=12
bash tools/_gpu_job_extras.sh
