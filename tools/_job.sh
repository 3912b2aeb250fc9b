cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
(timeout 170 python tests/fuzz_parity.py --cases 100000 --seed 404 --gpu 2>&1 | tail -4; echo "--- big"; timeout 120 python tests/fuzz_parity.py --cases 100000 --seed 505 --gpu --big 2>&1 | tail -4) | tee $O/r04_ad_fuzz_gpu.txt
