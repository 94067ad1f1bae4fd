cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline"
O=$GRAFT_REPO_ROOT/gpurun_out
MM2AMD_HOST_PROF=1 $B > $O/r04_bench_hostprof_v5.json 2> $O/r04_bench_hostprof_v5.log
MM2AMD_DEVICE_FINISH=1 $B > $O/r04_bench_devfin_v5.json 2> $O/r04_bench_devfin_v5.log
MM2AMD_LANES=7 $B > $O/r04_bench_lanes7_v5.json 2> $O/r04_bench_lanes7_v5.log
$B --threads 24 > $O/r04_bench_t24_v5.json 2> $O/r04_bench_t24_v5.log
for f in hostprof devfin lanes7 t24; do python - $O/r04_bench_${f}_v5.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'cpu_s', d['config']['host_cpu_s_per_step'], 'resident', d['config']['resident_gbases_per_s'])
PY
done
grep -i -A40 "cycles" $O/r04_bench_hostprof_v5.log | head -80
