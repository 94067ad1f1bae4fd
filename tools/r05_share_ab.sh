# one rank's share of an 8-GPU job (bench.py --as-rank-of 8) with different sub-batch sizes.   usage: bash tools/r05_share_ab.sh TAG
V=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
run() { env $2 timeout 900 python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --as-rank-of 8 > $O/r05_share_$1_$V.json 2> $O/r05_share_$1_$V.log
  python - $O/r05_share_$1_$V.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); a=d['config']['as_rank_of']
print(sys.argv[1].split('/')[-1], 'N1', d['value'], d['ms_per_step'], 'share ms', a['ms_per_step'], 'pred', a['predicted_strong_scaling'], 'cpu/Gb', a['host_cpu_s_per_gbase'])
PY
}
run two_1 MM2AMD_MIN_SUBBATCHES=2
run four_1 MM2AMD_MIN_SUBBATCHES=4
run two_2 MM2AMD_MIN_SUBBATCHES=2
run four_2 MM2AMD_MIN_SUBBATCHES=4
run six_1 MM2AMD_MIN_SUBBATCHES=6
