R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/tools/sustained_probe.py 2>&1 | grep -v "^\[" | tail -8
