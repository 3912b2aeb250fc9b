# GPU-box job of round 2, run A: probe + parity subset + bench at HEAD + kernel trace + PMC traffic (calibrated) + SQ counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_a}
./tools/pmc_calib.bin probe > $O/${TAG}_probe.txt 2>&1; cat $O/${TAG}_probe.txt
timeout 1200 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -6 | tee $O/${TAG}_pytest.txt
timeout 900 python bench.py ${BENCH_ARGS} > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -c 1500 $O/${TAG}_bench.log; cut -c1-1200 $O/${TAG}_bench.json
timeout 300 python bench.py --no-extras --no-cpu-baseline --tuning roe_march=0 2>/dev/null | tail -1 | cut -c1-900 > $O/${TAG}_bench_roe0.json; cut -c1-400 $O/${TAG}_bench_roe0.json
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG: $B)" | head -14
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/cal_f -o f -- ./tools/pmc_calib.bin copy > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/cal_w -o w -- ./tools/pmc_calib.bin copy > /dev/null 2>&1
python tools/pmc_traffic.py calib $O/cal_f/f_results.db $O/cal_w/w_results.db $O/pmc_traffic.json > $O/${TAG}_pmc_calib.txt 2>&1; cat $O/${TAG}_pmc_calib.txt
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db crm_rans_sa_upwind_8x160x128x64 $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $B)" > $O/${TAG}_pmc_traffic.txt 2>&1; tail -5 $O/${TAG}_pmc_traffic.txt
(python tools/pmc_summary.py $O/pmc_f/f_results.db; python tools/pmc_summary.py $O/pmc_w/w_results.db) | grep -v rocclr >> $O/${TAG}_pmc_traffic.txt
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD -d $O/pmc_sq -o s -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_sq/s_results.db | grep -v rocclr > $O/${TAG}_pmc_sq.txt; grep -c . $O/${TAG}_pmc_sq.txt
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $O/pmc_sq2 -o s -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_sq2/s_results.db | grep -v rocclr >> $O/${TAG}_pmc_sq.txt
rm -rf $O/prof $O/pmc_f $O/pmc_w $O/cal_f $O/cal_w $O/pmc_sq $O/pmc_sq2
ls -la $O | head -30
