export GIT=1278614
TAG=r05_fin2 bash tools/_gpu_job_full.sh
TAG=r05_fin2 EXTRAS="pc config3 config2 matvec" ROWS=10 bash tools/_gpu_job_extras.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out
( timeout 420 python tests/fuzz_parity.py --gpu --cases 2500 --seed 61 2>&1 | tail -2; timeout 200 python tests/fuzz_parity.py --gpu --big --cases 120 --seed 62 2>&1 | tail -2; timeout 300 python tests/fuzz_parity.py --gpu --jac --cases 2500 --seed 63 2>&1 | tail -2 ) | cut -c1-300 | tee $O/r05_fin2_fuzz.txt
TAG=r05_fin2 bash tools/_gpu_job_sq.sh 2>&1 | tail -12
