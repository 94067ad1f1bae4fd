cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out
T=$1; shift
i=0
for E in "$@"; do
  i=$((i+1))
  env $(echo $E | tr ',' ' ') python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --timed-only --threads 128 --as-rank-of 8 > $O/r04_bench_rank8_${T}_$i.json 2> $O/r04_bench_rank8_${T}_$i.log
  python - $O/r04_bench_rank8_${T}_$i.json "$E" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']; r=c['as_rank_of']
print(sys.argv[2], 'full', d['value'], 'share ms', r.get('ms_per_step'), 'share rate', r.get('share_gbases_per_s'), 'x8', r.get('n_x_share_gbases_per_s'), 'scaling', r.get('predicted_strong_scaling'), 'cpu/Gbase', r.get('host_cpu_s_per_gbase'), r.get('error'))
PY
done
