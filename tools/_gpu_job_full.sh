# full artefact set: GPU suite, calibration, default bench (extras + CPU baselines), kernel trace, PMC traffic
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r04_full}
if [ -z "$NOTESTS" ]; then timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | tee $O/${TAG}_pytest.txt; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
tools/pmc_calib.bin bw2 | tee $O/${TAG}_calibration.json
cp $O/${TAG}_calibration.json profiles/calibration.json
timeout 1200 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; tail -c 1500 $O/${TAG}_bench.log; cut -c1-400 $O/${TAG}_bench.json
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- $B > $O/${TAG}_prof.log 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_kernel_trace.md "($TAG, git ${GIT}: $B)" | head -12
BP="$B --tuning overlap=0"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f -- $BP > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w -- $BP > /dev/null 2>&1
python tools/pmc_traffic.py bench $O/pmc_f/f_results.db $O/pmc_w/w_results.db ${WL:-crm_rans_sa_upwind_8x160x128x64_bc} $O/pmc_traffic.json "${GIT:-unknown}" "profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- $BP)" > $O/${TAG}_pmc_traffic.txt 2>&1
tail -3 $O/${TAG}_pmc_traffic.txt
rm -rf $O/prof $O/pmc_f $O/pmc_w
