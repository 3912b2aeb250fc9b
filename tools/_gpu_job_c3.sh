# config-3 iteration (D-ADI x3 + SA DDADI x3): timing + kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r02_c3}
if [ -n "$PYTEST_K" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$PYTEST_K" 2>&1 | tail -3; fi
timeout 300 python tools/ab_config3.py "" 2>&1 | grep config-3 | tee $O/${TAG}_config3.txt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python tools/ab_config3.py "" > /dev/null 2>&1
python tools/rocpd_summary.py $O/prof/t_results.db $O/${TAG}_config3_trace.md "($TAG: config-3 iteration)" | grep -E "dadi|sa_s|sa_r|res_av|stage|time_step|kernel" | cut -c1-150
rm -rf $O/prof
