V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() { env $2 timeout 900 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/r05_gate_$1_$V.json 2> $O/r05_gate_$1_$V.log
  python - $O/r05_gate_$1_$V.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'avg_launch_ms', r['avg_launch_ms'], 'cpu_s', d['config']['host_cpu_s_per_step'])
PY
}
run g2a MM2AMD_DP_GATE=2
run g0a MM2AMD_X=1
run g4a MM2AMD_DP_GATE=4
run g2b MM2AMD_DP_GATE=2
run g0b MM2AMD_X=1
