cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for T in "0 0" "0 1" "1 0" "1 1" "0 0" "0 1"; do set -- $T
echo "jpipe=$1 kpipe=$2: $(timeout 600 python bench.py --no-cpu-baseline --only-extras config3 --force-extras --tuning dadi_jpipe=$1 --tuning dadi_kpipe=$2 2>&1 >/dev/null | grep -a 'config 3' | tail -1)"
done
