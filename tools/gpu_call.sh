R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ksw.py -x -q -m gpu 2>&1 | tail -15 > $O/stream_tests.log; tail -5 $O/stream_tests.log
bench() { n=$1; shift
  (cd /tmp; env "$@" timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/b_$n.out 2> $O/b_$n.log)
  tail -1 $O/b_$n.out > $O/b_$n.json
  python -c "
import json; d=json.load(open('$O/b_$n.json')); u=d['roofline'].get('unoverlapped_ms',{}); print('$n', d['value'], d['ms_per_step'], 'gapfill family unoverlapped', d['roofline']['valu']['unoverlapped_ms_per_step'], {k:v for k,v in u.items() if 'stream' in k or 'gapfill' in k})"
}
if grep -q "passed" $O/stream_tests.log && ! grep -q "failed" $O/stream_tests.log; then
bench stream A=1
cp $R/minimap2_amd/libmm2amd.so /tmp/libmm2amd_main.so
cp $R/minimap2_amd/build/variants/libmm2amd_w5.so $R/minimap2_amd/libmm2amd.so
bench w5 A=1
cp /tmp/libmm2amd_main.so $R/minimap2_amd/libmm2amd.so
fi
