R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 80 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/last.json 2> $O/last.log
python -c "
import json; d=json.loads(open('$O/last.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('index_probes'))"
