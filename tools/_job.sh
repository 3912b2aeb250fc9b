export GIT=972f398
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=r05_fin4 EXTRAS=pc ROWS=8 bash tools/_gpu_job_extras.sh
