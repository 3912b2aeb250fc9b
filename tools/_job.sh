cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O; TAG=r04_p
(timeout 1400 python tests/fuzz_parity.py --cases 20000 --seed 777 --gpu 2>&1 | grep -v " ok " | tail -5) | tee $O/${TAG}_fuzz_gpu.txt
(timeout 900 python tests/fuzz_parity.py --cases 600 --seed 778 --gpu --big 2>&1 | grep -v " ok " | tail -5) | tee -a $O/${TAG}_fuzz_gpu.txt
