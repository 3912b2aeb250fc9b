# round 3, first measurement of the fused gradient + viscous kernel: GPU parity of the RANS tests, A/B against the kernel pair, trace, PMC traffic
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r03_a}
timeout 900 python -m pytest tests/test_gpu_rans.py -m gpu -x -q 2>&1 | tail -6 | tee $O/${TAG}_pytest.txt
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.5"
for rep in 1 2; do
for T in "visc_gf=1" "visc_gf=0"; do
  echo "== $T" | tee -a $O/${TAG}_ab.txt
  timeout 300 $B --tuning $T 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['roofline'].get('kernels_ms'))" | tee -a $O/${TAG}_ab.txt
done; done
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG, git ${GIT}: $B)" | head -14
BP="$B --tuning overlap=0"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $BP > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $BP > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db crm_rans_sa_upwind_8x160x128x64 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $BP)" > $O/${TAG}_pmc_traffic.txt 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/pmc_traffic.json'))
e=t['crm_rans_sa_upwind_8x160x128x64']
cells=10485760
for k,v in e['kernels'].items():
    print(f"{k:18s} fetch {v['fetch_bytes']/1e9:7.3f} GB write {v['write_bytes']/1e9:6.3f} GB  -> {v['traffic_bytes_per_launch']/cells:7.1f} B/cell")
print(e['traffic_bytes_per_eval']/cells, "B/cell per eval", e['git'])
PY
rm -rf $O/prof $O/pmc_f $O/pmc_w
if [ -n "$SQ" ]; then
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $O/pmc_sq -o s -- $BP > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_sq/s_results.db | grep -v rocclr > $O/${TAG}_pmc_sq.txt
cat $O/${TAG}_pmc_sq.txt
rm -rf $O/pmc_sq
fi
