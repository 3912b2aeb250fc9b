# A/B of tuning sets on the whole bench (headline + extras): VARIANTS="a=1;b=2"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_x}
if [ -n "$PYTEST_K" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$PYTEST_K" 2>&1 | tail -3; fi
IFS=';' read -ra SETS <<< "$VARIANTS"
for T in "${SETS[@]}"; do
  ARGS=""; for kv in $T; do ARGS="$ARGS --tuning $kv"; done
  echo "== $T"; timeout 600 python bench.py --no-cpu-baseline --force-extras $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); e=d['extra']
print('headline', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})
print('4b', round(e['crm_rans_sa_matrix_8x160x128x64']['ms_per_step'],3), 'matvec', round(e['config5_gmres_proxy']['ms_per_matvec'],3), 'pc', round(e['pc_matrix_assembly']['ms'],1), 'config3', round(e['config3_dadi_iteration']['ms_per_iteration'],2), 'euler', round(e['euler_jst_8x128']['ms_per_step'],3), 'mg3w', round(e['mg_3w_cycle']['ms_per_cycle'],2))"
done 2>&1 | tee $O/${TAG}_extras_ab.txt
