R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -u -m pytest -p no:cacheprovider -m gpu -q -x tests/test_gpu_dropin.py -k "rmq" > $O/r02_pytest_gpu_rmq.log 2>&1; tail -25 $O/r02_pytest_gpu_rmq.log
