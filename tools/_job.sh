# scratch: the command of the last gpurun call of a session (the kept job scripts are tools/_gpu_job_*.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
T0=$(date +%s); python bench.py > $O/r05_last_bench.json 2> $O/r05_last_bench.log; echo "bench.py wall: $(( $(date +%s) - T0 )) s"; grep -a "PC matrix\|config 5\|config 3\|MG cycle\|343\|timed loop" $O/r05_last_bench.log | tail -8; tail -1 $O/r05_last_bench.json | cut -c1-200
