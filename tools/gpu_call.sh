R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/r02_pytest_gpu_v10.log; tail -4 $O/r02_pytest_gpu_v10.log
cd /tmp; timeout 500 python $R/tools/pmc_traffic.py --reads 20000 > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log; cp $R/profiles/pmc_traffic.json $O/pmc_traffic_new.json
