# Round 3, call 13: region_finish on the device + 512-column extension kernel: suite, bench at three thread counts, kernel stats, what is left in the lane-exact kernel
V=${1:-v13}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
MM2AMD_HOST_PROF=1 MM2AMD_BENCH_TRACE=1 timeout 500 python $R/bench.py --steps 8 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
grep "host piece\|steps in\|host CPU\|probe\|un-overlapped" $O/r03_bench_full_$V.log | cut -c1-400
for T in 32; do
  timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --threads $T > $O/r03_bench_t${T}_$V.json 2> $O/r03_bench_t${T}_$V.log
done
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r03_bench_full_kernel_stats_$V.txt; rm -rf $O/prof_ont
rm -f /tmp/jobs.tsv
MM2AMD_DUMP_JOBS=/tmp/jobs.tsv timeout 300 python $R/bench.py --reads 10000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/r03_bench_dump_$V.log
python - <<PY > $O/r03_ext_jobs_$V.txt
import collections
tot=collections.Counter(); cells=collections.Counter(); bad=0
for l in open('/tmp/jobs.tsv'):
    f5=l.rstrip('\n').split('\t')
    if len(f5)!=5: bad+=1; continue
    try: rnd,q,t,f,w=int(f5[0]),int(f5[1]),int(f5[2]),int(f5[3],16),int(f5[4])
    except ValueError: bad+=1; continue
    if f in (0x40,0xC2):
        kind='ext'
        if q<=0 or t<=0: why='empty'
        elif w>=0 and w<q+t: why='band binds, q+t<=%d'%(1024 if q+t<=1024 else 2048 if q+t<=2048 else 4096 if q+t<=4096 else 99999)
        elif t<=256 and q<=512: why='ext kernel 4 sets'
        elif t<=512 and q<=512: why='ext kernel 8 sets'
        else: why='band free but larger'
    else:
        kind='fill(0x%x)'%f; why='t<=%d'%(64 if t<=64 else 256 if t<=256 else 1024 if t<=1024 else 99999)
    key=(kind,'round%d'%min(rnd,2),why)
    tot[key]+=1; cells[key]+=q*t
print('unparsed lines (concurrent writers):',bad)
for k in sorted(tot): print('%-14s %-8s %-28s jobs %9d  cells %14d  mean q*t %9.0f'%(k[0],k[1],k[2],tot[k],cells[k],cells[k]/max(1,tot[k])))
PY
cat $O/r03_ext_jobs_$V.txt
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_t32_$V.json','r03_bench_full_${V}_under_rocprof.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']
        print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'), d['config']['host_cpu_s_per_step'], d['config']['host_threads_per_rank'])
        print(' unoverlapped', r.get('unoverlapped_ms'))
    except Exception as e: print(f, 'FAILED', e)
EOF2
head -16 $O/r03_bench_full_kernel_stats_$V.txt | cut -c1-150
