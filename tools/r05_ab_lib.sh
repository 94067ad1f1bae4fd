# A/B of two builds of the library inside one gpurun call: the tree's libmm2amd.so (A) against a variant .so (B) copied over it for its runs; the GPU cases named by
# -k EXPR run on A first.   usage: bash tools/r05_ab_lib.sh TAG VARIANT.so "PYTEST_K_EXPR"     Measurement scaffolding.
V=$1; B=$2; K=$3; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd $R; (timeout 1200 python -m pytest tests -x -q -m gpu -k "$K" 2>&1 | tail -3) > $O/r05_pytest_ab_$V.log; tail -1 $O/r05_pytest_ab_$V.log
cp $R/minimap2_amd/libmm2amd.so /tmp/libA.so
cd /tmp
run() { timeout 900 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/r05_ab_$1_$V.json 2> $O/r05_ab_$1_$V.log
  python - $O/r05_ab_$1_$V.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']; r=d['roofline']
u=r['unoverlapped_ms']; g=r['unoverlapped_gcells_per_s']
print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], {k:(round(u[k],1), g.get(k)) for k in u if 'extd2' in k}, c['pipeline_text_identical'])
PY
}
for i in 1 2; do
  cp /tmp/libA.so $R/minimap2_amd/libmm2amd.so; run A$i
  cp $R/$B $R/minimap2_amd/libmm2amd.so; run B$i
done
cp /tmp/libA.so $R/minimap2_amd/libmm2amd.so
