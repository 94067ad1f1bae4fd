# lanes A/B inside one gpurun call, map-hifi and map-ont.   usage: bash tools/r05_lanes_ab.sh TAG
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() { env $3 timeout 900 python $R/bench.py $2 --no-cpu-baseline > $O/r05_bench_$1_$V.json 2> $O/r05_bench_$1_$V.log
  python - $O/r05_bench_$1_$V.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], 'resident', c.get('resident_gbases_per_s'), 'cpu_s', c['host_cpu_s_per_step'])
PY
}
H="--preset map-hifi --reads 200000 --steps 3 --warmup 2"
run hifi_stag45 "$H" MM2AMD_LANE_STAGGER_MS=45
run hifi_stag0 "$H" MM2AMD_X=1
run hifi_stag90 "$H" MM2AMD_LANE_STAGGER_MS=90
run ont_stag45 "--steps 8 --warmup 3" MM2AMD_LANE_STAGGER_MS=45
run ont_stag0 "--steps 8 --warmup 3" MM2AMD_X=1
