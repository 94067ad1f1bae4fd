# Round 3, call 23: long-join re-chaining on the device: the suite, then the headline line (the benchmark's random reference has few re-chains: the path must cost nothing)
V=${1:-v23}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
grep "steps in\|host CPU\|probe" $O/r03_bench_full_$V.log | cut -c1-700
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}
    print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'ref', c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'))
    u=d['roofline']['unoverlapped_ms']; print({k:v for k,v in u.items() if 'long-join' in k or 'rechain' in k})
EOF2
