# Round 3, call 28: the suite and the driver's command after the backtrack kernel's LDS diet (16-bit path entries)
V=${1:-v28}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
python - <<EOF2
import json
d=json.loads(open('$O/r03_bench_full_$V.json').read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; u=d['roofline']['unoverlapped_ms']
print(d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'ref', c.get('value'), c.get('hits_identical_to_gpu'), 'backtrack', u.get('chain_backtrack_kernel'), 'fill', u.get('chain_fill_kernel'), 'sum %.0f'%sum(u.values()))
EOF2
