R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/r02_pytest_gpu_v11.log; tail -3 $O/r02_pytest_gpu_v11.log
cd /tmp
run() { n=$1; shift
  timeout 300 python $R/bench.py --no-cpu-baseline "$@" > $O/sz_$n.json 2> $O/sz_$n.log || tail -3 $O/sz_$n.log
  python -c "
import json; d=json.loads(open('$O/sz_$n.json').read().strip().split('\n')[-1]); u=d['roofline']['unoverlapped_ms']; print('$n', d['value'], d['ms_per_step'], 'sketch', u.get('sketch_kernel'), 'unoverlapped step', d['roofline']['unoverlapped_step_ms'])"
}
if grep -q passed $O/r02_pytest_gpu_v11.log && ! grep -q failed $O/r02_pytest_gpu_v11.log; then
run ont --steps 5 --warmup 1
run hifi --preset map-hifi --reads 200000 --steps 2 --warmup 1
fi
