# A/B inside one gpurun call: bench.py with the built library (A) against a variant library (B, e.g. minimap2_amd/variants/libmm2amd_v1cell.so =
# the commit before a kernel change, built in a worktree), in the order A B A B ...; the variant is copied over libmm2amd.so for its runs (the box's
# copy of the tree is scratch).   usage: bash tools/r04_ab.sh TAG VARIANT.so ROUNDS [bench.py arguments]     Measurement scaffolding.
V=$1; B=$2; N=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
ARGS=${@:---steps 8 --warmup 3 --no-cpu-baseline}
cp $R/minimap2_amd/libmm2amd.so /tmp/libmm2amd_A.so
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']; u=r.get('unoverlapped_ms') or {}
    fam={}
    for k,v in u.items(): fam[k.split('[')[0].split('<')[0]]=fam.get(k.split('[')[0].split('<')[0],0)+v
    print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], 'cpu_s', d['config']['host_cpu_s_per_step'], 'text_identical', d['config'].get('pipeline_text_identical'),
          'valu', (r.get('valu') or {}).get('frac'), {k:round(v,1) for k,v in sorted(fam.items())}, 'sum %.0f'%sum(u.values()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
# (VARIANT.so may be a comma-separated list: A B C A B C ...; the output files are named A, B, C, ...)
NAMES=(B C D E F)
IFS=',' read -ra VARS <<< "$B"
for i in $(seq 1 $N); do
  cp /tmp/libmm2amd_A.so $R/minimap2_amd/libmm2amd.so
  (cd /tmp; timeout 600 python $R/bench.py $ARGS > $O/r04_ab_${V}_A$i.json 2> $O/r04_ab_${V}_A$i.log)
  show $O/r04_ab_${V}_A$i.json
  k=0
  for X in "${VARS[@]}"; do
    W=${NAMES[$k]}; k=$((k+1))
    cp $R/$X $R/minimap2_amd/libmm2amd.so
    (cd /tmp; timeout 600 python $R/bench.py $ARGS > $O/r04_ab_${V}_$W$i.json 2> $O/r04_ab_${V}_$W$i.log)
    show $O/r04_ab_${V}_$W$i.json
  done
done
cp /tmp/libmm2amd_A.so $R/minimap2_amd/libmm2amd.so
