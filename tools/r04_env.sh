# One gpurun call, several environments: bench.py --timed-only (the pipeline clock only) under each comma-separated VAR=value list, in the order given
# ("-" = the defaults; "lib:PATH" = copy that library over libmm2amd.so for the runs that follow, "lib:-" = back to the built one).
#   usage: bash tools/r04_env.sh TAG ENV1 ENV2 ...      Measurement scaffolding.
cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T=$1; shift
i=0
cp $GRAFT_REPO_ROOT/minimap2_amd/libmm2amd.so /tmp/libmm2amd_built.so
for E in "$@"; do
  case "$E" in
    lib:-) cp /tmp/libmm2amd_built.so $GRAFT_REPO_ROOT/minimap2_amd/libmm2amd.so; echo "library: built"; continue ;;
    lib:*) cp $GRAFT_REPO_ROOT/${E#lib:} $GRAFT_REPO_ROOT/minimap2_amd/libmm2amd.so; echo "library: ${E#lib:}"; continue ;;
  esac
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
cp /tmp/libmm2amd_built.so $GRAFT_REPO_ROOT/minimap2_amd/libmm2amd.so
