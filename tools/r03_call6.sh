# Round 3, sixth GPU call: the wave-scalar tie replay on hardware (suite), un-overlapped sort times, sustained-load probe
V=${1:-v6}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
MM2AMD_BENCH_TRACE=1 timeout 400 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log; grep "batch" $O/r03_bench_full_$V.log | tail -30 | cut -c1-100
timeout 300 python $R/tools/sustained_probe.py 2>&1 | grep pause
python - <<EOF
import json
for f in ['r03_bench_full_$V.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), d.get('output_stage'))
    print(' unoverlapped', r.get('unoverlapped_ms'))
EOF
