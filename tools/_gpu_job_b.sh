# run B: full GPU suite + bench + A/B of the viscous variants + kernel trace + PMC
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_b}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/${TAG}_pytest.txt
S="python bench.py --no-extras --no-cpu-baseline --min-seconds 0.5"
for T in "visc_sb=0" "visc_sb=1" "viscous_tiled=1" "roe_march=0"; do
  echo "== $T"; timeout 300 $S --tuning $T 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['kernels_ms'].items()})"
done 2>&1 | tee $O/${TAG}_variants.txt
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -c 600 $O/${TAG}_bench.log; cut -c1-600 $O/${TAG}_bench.json
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG: $B)" | head -12
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $B > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db crm_rans_sa_upwind_8x160x128x64 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $B)" > $O/${TAG}_pmc_traffic.txt 2>&1
(python tools/pmc_summary.py $O/pmc_f/f_results.db; python tools/pmc_summary.py $O/pmc_w/w_results.db) | grep -v rocclr >> $O/${TAG}_pmc_traffic.txt
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $O/pmc_sq -o s -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_sq/s_results.db | grep -v rocclr > $O/${TAG}_pmc_sq.txt
rm -rf $O/prof $O/pmc_f $O/pmc_w $O/pmc_sq
