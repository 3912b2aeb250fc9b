# A/B of tuning variants on the default bench step: VARIANTS="a=1 b=2;c=3" (semicolon separated sets)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_v}
if [ -n "$PYTEST_K" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$PYTEST_K" 2>&1 | tail -4; fi
S="python bench.py --no-extras --no-cpu-baseline --min-seconds 0.5 ${WORKLOAD:+--workload $WORKLOAD}"
IFS=';' read -ra SETS <<< "$VARIANTS"
for T in "${SETS[@]}"; do
  ARGS=""; for kv in $T; do ARGS="$ARGS --tuning $kv"; done
  echo "== $T"; timeout 300 $S $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
done 2>&1 | tee $O/${TAG}_variants.txt
