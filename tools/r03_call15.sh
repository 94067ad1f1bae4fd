# Round 3, call 15: region_finish with the CIGAR in LDS and a 32-lane walk; exact band condition; 768-column extension kernel: suite + bench
V=${1:-v15}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
MM2AMD_HOST_PROF=1 timeout 500 python $R/bench.py --steps 8 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
grep "steps in\|host CPU\|un-overlapped\|probe" $O/r03_bench_full_$V.log | cut -c1-600
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']; c=d.get('cpu_baseline') or {}
    print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), d['config']['host_cpu_s_per_step'], d['config']['host_threads_per_rank'], c.get('value'), c.get('hits_identical_to_gpu'))
    for k,v in sorted(r['unoverlapped_ms'].items(), key=lambda x:-x[1]): print('   %-44s %8.2f  %s'%(k,v,r.get('unoverlapped_gcells_per_s',{}).get(k,'')))
    print('   sum', sum(r['unoverlapped_ms'].values()))
EOF2
