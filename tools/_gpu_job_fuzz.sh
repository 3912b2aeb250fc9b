# randomised parity sweeps on the GPU (tests/fuzz_parity.py --gpu): every entry point, large blocks, the Jacobian assemblies
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-fuzz}
: > $O/${TAG}_fuzz.txt
for s in ${SEEDS:-101 102 103 104}; do timeout 600 python tests/fuzz_parity.py --gpu --cases ${CASES:-2000} --seed $s 2>&1 | tail -1 | tee -a $O/${TAG}_fuzz.txt; done
timeout 900 python tests/fuzz_parity.py --gpu --big --cases ${BIG:-150} --seed ${BIGSEED:-105} 2>&1 | tail -1 | tee -a $O/${TAG}_fuzz.txt
for s in ${JSEEDS:-106 107}; do timeout 900 python tests/fuzz_parity.py --gpu --jac --cases ${JAC:-1000} --seed $s 2>&1 | tail -1 | tee -a $O/${TAG}_fuzz.txt; done
