# Round 3, seventh GPU call (light): bench with the pipeline trace after the zero-copy hand-over
V=${1:-v7}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 300 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_aligner.py -x -q -m gpu -k "ont or hifi or fixtures or pair" 2>&1 | tail -2)
MM2AMD_BENCH_TRACE=1 timeout 400 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log; grep "batch" $O/r03_bench_full_$V.log | tail -30 | cut -c1-100
python - <<EOF
import json
for f in ['r03_bench_full_$V.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), d.get('output_stage'), d['config']['host_cpu_s_per_step'])
    print(' unoverlapped', r.get('unoverlapped_ms'))
EOF
