# Round 3, call 22: the suite and the headline line at the round's last code commit
V=${1:-v22}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
(cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/r03_smoke_$V.log; tail -1 $O/r03_smoke_$V.log
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
MM2AMD_HOST_PROF=1 timeout 600 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_${V}_hostprof.json 2> $O/r03_bench_full_${V}_hostprof.log
grep "steps in\|host CPU\|probe\|host piece\|un-overlapped" $O/r03_bench_full_$V.log $O/r03_bench_full_${V}_hostprof.log | cut -c1-330
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_full_${V}_hostprof.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}
    print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'threads', d['config']['host_threads_per_rank'], 'ref', c.get('value'), c.get('cores'), c.get('hits_identical_to_gpu'), 'valu', d['roofline']['valu']['frac'])
EOF2
