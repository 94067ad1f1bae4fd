# Round 3, tenth GPU call: extension kernel + blocking stream waits + quota-aware thread counts: suite, bench, rocprof summary
V=${1:-v10}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
MM2AMD_BENCH_TRACE=1 timeout 400 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log; grep "batch" $O/r03_bench_full_$V.log | tail -12 | cut -c1-100; grep "steps in\|host CPU\|probe" $O/r03_bench_full_$V.log | cut -c1-300
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r03_bench_full_kernel_stats_$V.txt; rm -rf $O/prof_ont
python - <<EOF
import json
for f in ['r03_bench_full_$V.json','r03_bench_full_${V}_under_rocprof.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']
        print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'), d.get('output_stage'), d['config']['host_cpu_s_per_step'], d['config']['host_threads_per_rank'])
        print(' unoverlapped', r.get('unoverlapped_ms'))
    except Exception as e: print(f, 'FAILED', e)
EOF
head -14 $O/r03_bench_full_kernel_stats_$V.txt | cut -c1-150
