# Round 3, call 19: the evidence call at the last commit (host threads = the CPU quota): suite, smoke, bench with the reference baseline, rocprofv3 kernel stats, map-hifi line
V=${1:-v19}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
(cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/r03_smoke_$V.log; tail -2 $O/r03_smoke_$V.log
MM2AMD_HOST_PROF=1 MM2AMD_BENCH_TRACE=1 timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
grep "steps in\|host CPU\|un-overlapped\|probe\|host piece" $O/r03_bench_full_$V.log | cut -c1-500
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r03_bench_full_kernel_stats_$V.txt; rm -rf $O/prof_ont
timeout 600 python $R/bench.py --preset map-hifi --reads 200000 --steps 3 --warmup 1 --cpu-sample 20000 > $O/r03_bench_hifi_$V.json 2> $O/r03_bench_hifi_$V.log
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_full_${V}_under_rocprof.json','r03_bench_hifi_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']; c=d.get('cpu_baseline') or {}
        print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'threads', d['config']['host_threads_per_rank'], 'ref', c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'))
        print('  roofline', {k:v for k,v in r.items() if k in ('bound','achieved','peak','unit','frac','traffic')}, 'valu', r.get('valu'))
        for k,v in sorted(r['unoverlapped_ms'].items(), key=lambda x:-x[1]): print('   %-44s %8.2f  %s'%(k,v,r.get('unoverlapped_gcells_per_s',{}).get(k,'')))
        print('   sum', sum(r['unoverlapped_ms'].values()))
    except Exception as e: print(f,'FAILED',e)
EOF2
head -24 $O/r03_bench_full_kernel_stats_$V.txt | cut -c1-150
