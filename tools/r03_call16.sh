# Round 3, call 16: A/B on one box -- regions finished on the device (default) vs on the host; the DP cases followed by the full-size cases in one process
V=${1:-v16}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for MODE in device host device2 host2; do
  if [ ${MODE:0:4} = host ]; then export unset MM2AMD_DEVICE_FINISH; else export MM2AMD_DEVICE_FINISH=1; fi
  timeout 400 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r03_bench_${MODE}_$V.json 2> $O/r03_bench_${MODE}_$V.log
done
unset MM2AMD_DEVICE_FINISH
python - <<EOF2
import json
for m in ['device','host','device2','host2']:
    try:
        d=json.loads(open('$O/r03_bench_%s_$V.json'%m).read().strip().split('\n')[-1]); r=d['roofline']
        u=r['unoverlapped_ms']
        print(m, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'finish %.1f ext %.1f extd2 %.1f sum %.0f'%(u.get('region_finish_kernel',0), sum(v for k,v in u.items() if k.startswith('ksw_ext_kernel')), sum(v for k,v in u.items() if k.startswith('ksw_extd2')), sum(u.values())))
    except Exception as e: print(m,'FAILED',e)
EOF2
(cd $R && timeout 900 python -m pytest tests/test_gpu_aligner.py tests/test_gpu_ksw.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -40) > $O/r03_pytest_ksw_fullsize_$V.log; tail -40 $O/r03_pytest_ksw_fullsize_$V.log | cut -c1-300
