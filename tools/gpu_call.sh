R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 700 python $R/tools/e2e_wall.py --skip-index-check --out $O/r02_e2e_wall_v3.json > $O/e2e.log 2>&1; tail -2 $O/e2e.log | cut -c1-1500
