# Round 3, call 11: where the host CPU seconds of a step go (per stage and per piece), thread-count variants under the 16-CPU quota,
# and which extension jobs still land in the lane-exact kernel
V=${1:-v11}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
MM2AMD_HOST_PROF=1 MM2AMD_BENCH_TRACE=1 timeout 500 python $R/bench.py --steps 6 --warmup 2 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
grep "host piece\|cpu_\|steps in\|host CPU\|probe\|un-overlapped" $O/r03_bench_full_$V.log | cut -c1-400
for T in 12 24 32; do
  timeout 300 python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --threads $T > $O/r03_bench_t${T}_$V.json 2> $O/r03_bench_t${T}_$V.log
done
rm -f /tmp/jobs.tsv
MM2AMD_DUMP_JOBS=/tmp/jobs.tsv timeout 300 python $R/bench.py --reads 10000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $O/r03_bench_dump_$V.log
python - <<PY > $O/r03_ext_jobs_$V.txt
import collections
tot=collections.Counter(); cells=collections.Counter()
for l in open('/tmp/jobs.tsv'):
    rnd,q,t,f,w=l.split('\t'); q=int(q); t=int(t); f=int(f,16); w=int(w); rnd=int(rnd)
    if f in (0x40,0xC2): kind='ext'
    elif f & 0x40: kind='ext+other(0x%x)'%f
    else: kind='fill(0x%x)'%f
    if kind=='ext':
        if q<=0 or t<=0: why='empty'
        elif w>=0 and w<q+t: why='band binds: q+t<=%d'%(1024 if q+t<=1024 else 2048 if q+t<=2048 else 4096 if q+t<=4096 else 99999)
        elif t<=256 and q<=512: why='eligible'
        elif t<=512 and q<=512: why='t<=512'
        elif t<=1024 and q<=1024: why='<=1024'
        else: why='larger'
        key=(kind,'round%d'%min(rnd,2),why)
    else: key=(kind,'round%d'%min(rnd,2),'')
    tot[key]+=1; cells[key]+=q*t
for k in sorted(tot): print('%-28s %-8s %-24s jobs %9d  cells %14d  mean q*t %9.0f'%(k[0],k[1],k[2],tot[k],cells[k],cells[k]/max(1,tot[k])))
PY
cat $O/r03_ext_jobs_$V.txt
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_t12_$V.json','r03_bench_t24_$V.json','r03_bench_t32_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']
        print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), d['config'].get('handover_then_map_gbases_per_s'), c.get('value'), c.get('cores'), d['config']['host_cpu_s_per_step'], d['config']['host_threads_per_rank'])
        print(' unoverlapped', r.get('unoverlapped_ms'))
    except Exception as e: print(f, 'FAILED', e)
EOF2
