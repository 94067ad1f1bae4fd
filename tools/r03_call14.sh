# Round 3, call 14 (light): flat region_finish walk, exact band condition + 768-column extension kernel: DP and full-size cases, one bench run
V=${1:-v14}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests/test_gpu_ksw.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
MM2AMD_HOST_PROF=1 timeout 500 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
grep "host piece\|steps in\|host CPU\|un-overlapped" $O/r03_bench_full_$V.log | cut -c1-600
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), d['config']['host_cpu_s_per_step'], d['config']['host_threads_per_rank'])
    for k,v in sorted(r['unoverlapped_ms'].items(), key=lambda x:-x[1]): print('   %-44s %8.2f  %s'%(k,v,r.get('unoverlapped_gcells_per_s',{}).get(k,'')))
    print('   sum', sum(r['unoverlapped_ms'].values()))
EOF2
