# One gpurun call, several environments: bench.py --timed-only (the pipeline clock only) under each comma-separated VAR=value list, in the order given
# ("-" = the defaults).   usage: bash tools/r04_env.sh TAG ENV1 ENV2 ...      Measurement scaffolding.
cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=$1; shift
i=0
for E in "$@"; do
  i=$((i+1))
  if [ "$E" = "-" ]; then EV=""; else EV=$(echo $E | tr ',' ' '); fi
  env $EV timeout 300 python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --timed-only > $O/r04_env_${T}_$i.json 2> $O/r04_env_${T}_$i.log
  python - $O/r04_env_${T}_$i.json "$E" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']
    print(sys.argv[2], d['value'], 'ms/step', d['ms_per_step'], 'cpu_s', c['host_cpu_s_per_step'], 'text_identical', c.get('pipeline_text_identical'))
except Exception as e: print(sys.argv[2], 'FAILED', e)
PY
done
