# Round 3, call 21: what allocations cost (per call or per byte); region_finish with LDS sized by the launch's longest CIGAR: A/B on one box
V=${1:-v21}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$R/tools/build/alloc_cost > $O/r03_alloc_cost_$V.txt 2>&1; cat $O/r03_alloc_cost_$V.txt
for MODE in device host device2 host2; do
  if [ ${MODE:0:4} = host ]; then unset MM2AMD_DEVICE_FINISH; else export MM2AMD_DEVICE_FINISH=1; fi
  timeout 400 python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/r03_bench_${MODE}_$V.json 2> $O/r03_bench_${MODE}_$V.log
done
unset MM2AMD_DEVICE_FINISH
python - <<EOF2
import json
for m in ['device','host','device2','host2']:
    try:
        d=json.loads(open('$O/r03_bench_%s_$V.json'%m).read().strip().split('\n')[-1]); r=d['roofline']
        u=r['unoverlapped_ms']
        print(m, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'finish %.1f sum %.0f'%(u.get('region_finish_kernel',0), sum(u.values())))
    except Exception as e: print(m,'FAILED',e)
EOF2
