# Round 3, call 24: the replay's two-bucket closed form: suite, then the headline line twice (sort time in roofline.unoverlapped_ms)
V=${1:-v24}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
MM2AMD_NO_TWO_BUCKET=1 timeout 600 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r03_bench_walk_$V.json 2> $O/r03_bench_walk_$V.log
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_walk_$V.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}
    u=d['roofline']['unoverlapped_ms']
    print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'ref', c.get('value'), c.get('hits_identical_to_gpu'), 'sort', {k:v for k,v in u.items() if k.startswith('anchor_sort')}, 'sum %.0f'%sum(u.values()))
EOF2
