# SQ counters of the kernels of the default bench step: two rocprofv3 --pmc passes of four counters each (no trace domains beside them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
TAG=${TAG:-r04_sq}
B="python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --min-seconds 0.2 --tuning overlap=0"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq_a -o a -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY -d $O/sq_b -o b -- $B > /dev/null 2>&1
(echo "# $TAG, git ${GIT}: rocprofv3 --pmc (two passes) -- $B"; python tools/pmc_summary.py $O/sq_a/a_results.db; python tools/pmc_summary.py $O/sq_b/b_results.db) | grep -v "rocclr\|k_face_vectors\|k_etot\|at::native" | sort > $O/${TAG}_pmc_sq.txt
python tools/pmc_sq.py $O/sq_a/a_results.db $O/sq_b/b_results.db ${WL:-crm_rans_sa_upwind_8x160x128x64_bc} $O/pmc_sq.json "${GIT:-unknown}" "profiles/${TAG}_pmc_sq.txt (rocprofv3 --pmc, two passes of four SQ counters -- $B)"
rm -rf $O/sq_a $O/sq_b
