# map-hifi A/B inside one gpurun call: the device's chains -> hits -> windows path against the host's, and two polling intervals.   usage: bash tools/r05_hifi_ab.sh TAG
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() { env $2 timeout 900 python $R/bench.py --preset map-hifi --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline > $O/r05_bench_hifi_$1_$V.json 2> $O/r05_bench_hifi_$1_$V.log
  python - $O/r05_bench_hifi_$1_$V.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']
print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], 'resident', c.get('resident_gbases_per_s'), 'cpu_s', c['host_cpu_s_per_step'], c.get('host_cpu_s_per_step_by_thread_name'))
PY
}
run dev MM2AMD_X=1
run host MM2AMD_DEVICE_REGIONS=0
run dev_wait200 MM2AMD_WAIT_MAX_US=200
run dev_lanes12 MM2AMD_LANES=12
