cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; TAG=r04_i
timeout 600 python bench.py --no-cpu-baseline --only-extras shard > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; grep -a "shard\|failed" $O/${TAG}_bench.log | tail -3
export EXTRAS=shard TAG=r04_i ROWS=16
bash tools/_gpu_job_extras.sh
