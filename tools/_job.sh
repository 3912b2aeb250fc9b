cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "jacobian or _ad or pc_" 2>&1 | tail -4
for S in 21 22 23; do
  timeout 500 python tests/fuzz_parity.py --gpu --jac --cases 400 --seed $S > $O/r05_fuzz_jac_$S.txt 2>&1
  tail -3 $O/r05_fuzz_jac_$S.txt | cut -c1-600
done
