# lane-exact kernel: a sixteen-wave workgroup per wide job (one round per band-751 row) against eight waves (two rounds).   usage: bash tools/r05_team16_ab.sh TAG
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R; (MM2AMD_KSW_TEAM16=1 timeout 900 python -m pytest tests/test_gpu_ksw.py tests/test_gpu_regions.py -x -q -m gpu 2>&1 | tail -3) > $O/r05_pytest_team16_$V.log; tail -1 $O/r05_pytest_team16_$V.log
cd /tmp
run() { env $2 timeout 900 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/r05_bench_$1_$V.json 2> $O/r05_bench_$1_$V.log
  python - $O/r05_bench_$1_$V.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']; r=d['roofline']
u=r['unoverlapped_ms']; g=r['unoverlapped_gcells_per_s']
print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], {k:(round(u[k],1), g.get(k)) for k in u if 'extd2' in k}, c['pipeline_text_identical'])
PY
}
run team8 MM2AMD_X=1
run team16 MM2AMD_KSW_TEAM16=1
run team8b MM2AMD_X=1
run team16b MM2AMD_KSW_TEAM16=1
