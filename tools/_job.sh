cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "jacobian or forward or _ad or useAD or matvec or pc_" 2>&1 | tail -5
TAG=r05_r EXTRAS=pc ROWS=25 bash tools/_gpu_job_extras.sh
cat $O/r05_r_pc.json | tail -1 | cut -c1-1500
