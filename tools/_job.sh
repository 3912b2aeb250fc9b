# scratch: the command of the last gpurun call of a session (the kept job scripts are tools/_gpu_job_*.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
( for S in 111 112 113; do timeout 200 python tests/fuzz_parity.py --gpu --cases 3000 --seed $S 2>&1 | tail -1; done
  timeout 200 python tests/fuzz_parity.py --gpu --big --cases 150 --seed 114 2>&1 | tail -1
  for S in 115 116; do timeout 300 python tests/fuzz_parity.py --gpu --jac --cases 2500 --seed $S 2>&1 | tail -1; done ) | cut -c1-400 | tee $O/r05_fin5_fuzz.txt
