# final artefacts of the round at HEAD: full GPU suite, default bench (with extras + CPU baseline), kernel trace, PMC traffic, SQ counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_k}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/${TAG}_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
(python bench.py --gpus 2 --no-extras --no-cpu-baseline 2>&1 | tail -2) | tee $O/${TAG}_gpus2_on_1gpu_box.txt
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -c 900 $O/${TAG}_bench.log; cut -c1-300 $O/${TAG}_bench.json
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG, git ${GIT}: $B)" | head -12
# counters per kernel: the kernels one after the other on one queue (concurrent kernels share the counters)
BP="$B --tuning overlap=0 --tuning roe_grad_mix=0"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $BP > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $BP > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db crm_rans_sa_upwind_8x160x128x64 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $B)" > $O/${TAG}_pmc_traffic.txt 2>&1
(python tools/pmc_summary.py $O/pmc_f/f_results.db; python tools/pmc_summary.py $O/pmc_w/w_results.db) | grep -v rocclr >> $O/${TAG}_pmc_traffic.txt
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $O/pmc_sq -o s -- $BP > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_sq/s_results.db | grep -v rocclr > $O/${TAG}_pmc_sq.txt
B2="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2 --workload euler_jst_8x128"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f2 -o f -- $B2 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w2 -o w -- $B2 > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f2/f_results.db $O/pmc_w2/w_results.db euler_jst_8x128 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic_euler.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $B2)" > $O/${TAG}_pmc_traffic_euler.txt 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/pmc_traffic.json'))
for wl,cells in (('crm_rans_sa_upwind_8x160x128x64',10485760),('euler_jst_8x128',16777216)):
    e=t[wl]
    for k,v in e['kernels'].items():
        print(f"{wl[:12]} {k:18s} fetch {v['fetch_bytes']/1e9:7.3f} GB write {v['write_bytes']/1e9:6.3f} GB  -> {v['traffic_bytes_per_launch']/cells:7.1f} B/cell")
    print(e['traffic_bytes_per_eval']/cells, "B/cell per eval", e['git'])
PY
rm -rf $O/prof $O/pmc_f $O/pmc_w $O/pmc_sq $O/pmc_f2 $O/pmc_w2
