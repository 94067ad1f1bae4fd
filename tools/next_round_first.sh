# First GPU call of the next round: the opt-in device paths, then a first short-read bench line.
# usage (from the dev container): gpurun --timeout 900 -- 'bash tools/next_round_first.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
MM2AMD_PENDING=1 timeout 300 python -u -m pytest -v -p no:cacheprovider -m gpu tests/test_gpu_pending.py > $O/r02_pending_gpu.log 2>&1
tail -25 $O/r02_pending_gpu.log
cd /tmp; export TMPDIR=/tmp
timeout 500 python $R/bench.py --preset sr --reads 1000000 --steps 2 --warmup 1 --cpu-sample 100000 > $O/r02_bench_sr_v1.json 2> $O/r02_bench_sr_v1.log
tail -3 $O/r02_bench_sr_v1.log; cat $O/r02_bench_sr_v1.json
