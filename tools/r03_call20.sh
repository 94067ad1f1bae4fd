# Round 3, call 20: the headline line as the driver runs it (no diagnostics in the environment), twice
V=${1:-v20}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
timeout 600 python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_full_${V}b.json 2> $O/r03_bench_full_${V}b.log
grep "steps in\|host CPU\|probe" $O/r03_bench_full_$V.log $O/r03_bench_full_${V}b.log | cut -c1-200
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_full_${V}b.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}
    print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'threads', d['config']['host_threads_per_rank'], 'ref', c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'), 'valu', d['roofline']['valu']['frac'])
EOF2
