# Round 3, call 27: the other configurations at the last commit, pipeline clock (BASELINE configs[4]: spliced cDNA reads; 2 x 150 b read pairs)
V=${1:-v27}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 500 python $R/bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --cpu-sample 3000 > $O/r03_bench_splice_$V.json 2> $O/r03_bench_splice_$V.log
timeout 500 python $R/bench.py --preset sr --reads 1000000 --steps 2 --warmup 1 --cpu-sample 100000 > $O/r03_bench_sr_$V.json 2> $O/r03_bench_sr_$V.log
python - <<EOF2
import json
for f in ['r03_bench_splice_$V.json','r03_bench_sr_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}
        print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'ref', c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'))
    except Exception as e: print(f,'FAILED',e)
EOF2
tail -3 $O/r03_bench_splice_$V.log | cut -c1-300; tail -3 $O/r03_bench_sr_$V.log | cut -c1-300
