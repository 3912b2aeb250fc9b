# PMC traffic + SQ counters of the default bench step (short)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_c}
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2 ${BENCH_EXTRA}"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $B > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db crm_rans_sa_upwind_8x160x128x64 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $B)" > $O/${TAG}_pmc_traffic.txt 2>&1
(python tools/pmc_summary.py $O/pmc_f/f_results.db; python tools/pmc_summary.py $O/pmc_w/w_results.db) | grep -v rocclr >> $O/${TAG}_pmc_traffic.txt
python - <<'PY'
import json
t=json.load(open('gpurun_out/pmc_traffic.json'))
e=t['crm_rans_sa_upwind_8x160x128x64']
cells=10485760
for k,v in e['kernels'].items():
    print(f"{k:18s} fetch {v['fetch_bytes']/1e9:7.3f} GB write {v['write_bytes']/1e9:6.3f} GB  -> {v['traffic_bytes_per_launch']/cells:7.1f} B/cell")
print(e['traffic_bytes_per_eval']/cells, "B/cell per eval", e['git'])
PY
rm -rf $O/pmc_f $O/pmc_w
