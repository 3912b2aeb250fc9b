# A/B of tuning sets on the default bench workload: AB="k=v k=v|k=v|..." (sets separated by |), REPS repetitions, optional TESTS="file ..."
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r04_x}
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest $TESTS -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | tee $O/${TAG}_pytest.txt; fi
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.5 ${BENCH_EXTRA}"
IFS='|' read -ra SETS <<< "$AB"
for rep in $(seq 1 ${REPS:-2}); do
for S in "${SETS[@]}"; do
  T=""; for kv in $S; do T="$T --tuning $kv"; done
  echo "== $S" | tee -a $O/${TAG}_ab.txt
  timeout 300 $B $T 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['roofline'].get('kernels_ms'))" | tee -a $O/${TAG}_ab.txt
done; done
if [ -n "$TRACE" ]; then
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG, git ${GIT}: $B)" | head -12
rm -rf $O/prof
fi
if [ -n "$PMC" ]; then
BP="$B --tuning overlap=0"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $BP > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $BP > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db ${WL:-crm_rans_sa_upwind_8x160x128x64_bc} $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $BP)" > $O/${TAG}_pmc_traffic.txt 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/pmc_traffic.json'))
import os
e=t[os.environ.get('WL','crm_rans_sa_upwind_8x160x128x64_bc')]
cells=10485760
for k,v in e['kernels'].items():
    print(f"{k:18s} fetch {v['fetch_bytes']/1e9:7.3f} GB write {v['write_bytes']/1e9:6.3f} GB  -> {v['traffic_bytes_per_launch']/cells:7.1f} B/cell")
print(e['traffic_bytes_per_eval']/cells, "B/cell per eval", e['git'])
PY
rm -rf $O/pmc_f $O/pmc_w
fi
