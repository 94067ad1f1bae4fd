R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  timeout 200 python $R/bench.py --preset sr --reads 1000000 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/sr_$n.json 2> $O/sr_$n.log || tail -3 $O/sr_$n.log
  python -c "
import json; d=json.loads(open('$O/sr_$n.json').read().strip().split('\n')[-1]); print('$n', d['value'], d['ms_per_step'], d['config']['host_cpu_s_per_step'])"
  grep "step 1" $O/sr_$n.log | cut -c1-330
}
run t64 --threads 64
run t128 --threads 128
run t192 --threads 192
