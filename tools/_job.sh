cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="--no-extras --no-cpu-baseline --steps 20 --warmup 3 --min-seconds 0.3"
for r in 1 2 3; do
for L in libadflow_gpu.so libadflow_gpu_ilp.so; do
python -c "
import sys, os
import adflow_amd.capi as c
c.LIB_PATH = os.path.join(os.path.dirname(c.__file__), 'lib', '$L')
sys.argv = ['bench.py'] + '$B'.split()
import bench
bench.main()
" 2>&1 | grep -a "phases\|timed loop" | sed "s/^/$L: /" | cut -c1-260
done; done
