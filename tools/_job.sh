export GIT=c67b1d3
TAG=r05_fin3 bash tools/_gpu_job_full.sh
TAG=r05_fin3 EXTRAS="pc config3 config2 matvec" ROWS=10 bash tools/_gpu_job_extras.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 300 python tests/fuzz_parity.py --gpu --cases 2000 --seed 91 2>&1 | tail -1; timeout 200 python tests/fuzz_parity.py --gpu --big --cases 100 --seed 92 2>&1 | tail -1; timeout 300 python tests/fuzz_parity.py --gpu --jac --cases 2500 --seed 93 2>&1 | tail -1 ) | cut -c1-300 | tee $O/r05_fin3_fuzz.txt
TAG=r05_fin3 bash tools/_gpu_job_sq.sh 2>&1 | tail -6
