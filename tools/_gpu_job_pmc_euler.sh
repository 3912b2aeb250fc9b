# PMC traffic of the Euler workload (time step + march)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_e}
B2="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2 --workload euler_jst_8x128 --tuning overlap=0 ${BENCH_EXTRA}"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f2 -o f -- $B2 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w2 -o w -- $B2 > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f2/f_results.db $O/pmc_w2/w_results.db euler_jst_8x128 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic_euler.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $B2)" > $O/${TAG}_pmc_traffic_euler.txt 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/pmc_traffic.json'))
e=t['euler_jst_8x128']; cells=16777216
for k,v in e['kernels'].items():
    print(f"{k:18s} fetch {v['fetch_bytes']/1e9:7.3f} GB write {v['write_bytes']/1e9:6.3f} GB  -> {v['traffic_bytes_per_launch']/cells:7.1f} B/cell")
PY
rm -rf $O/pmc_f2 $O/pmc_w2
