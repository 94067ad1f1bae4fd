# Round 3, call 26: the evidence call at the round's last commit: suite, smoke, the driver's command, the same under rocprofv3; then host threads scarce
# (4 per GPU, as a rank of an 8-GPU job under a 16-CPU quota has them): regions finished on the device against on the host
V=${1:-v26}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_$V.log; tail -3 $O/r03_pytest_gpu_$V.log
(cd $R && timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > $O/r03_smoke_$V.log; tail -1 $O/r03_smoke_$V.log
timeout 600 python $R/bench.py --steps 10 --warmup 3 > $O/r03_bench_full_$V.json 2> $O/r03_bench_full_$V.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r03_bench_full_kernel_stats_$V.txt; rm -rf $O/prof_ont
for MODE in device host; do
  if [ $MODE = host ]; then export MM2AMD_DEVICE_FINISH=0; else export MM2AMD_DEVICE_FINISH=1; fi
  timeout 400 python $R/bench.py --threads 4 --steps 4 --warmup 1 --no-cpu-baseline > $O/r03_bench_t4_${MODE}_$V.json 2> $O/r03_bench_t4_${MODE}_$V.log
done
unset MM2AMD_DEVICE_FINISH
python - <<EOF2
import json
for f in ['r03_bench_full_$V.json','r03_bench_full_${V}_under_rocprof.json','r03_bench_t4_device_$V.json','r03_bench_t4_host_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; u=d['roofline']['unoverlapped_ms']
        print(f, d['value'], d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu', d['config']['host_cpu_s_per_step'], 'threads', d['config']['host_threads_per_rank'], 'ref', c.get('value'), c.get('hits_identical_to_gpu'), 'valu', d['roofline']['valu']['frac'], 'finish', u.get('region_finish_kernel'), 'sum %.0f'%sum(u.values()))
    except Exception as e: print(f,'FAILED',e)
d=json.loads(open('$O/r03_bench_full_$V.json').read().strip().split('\n')[-1]); u=d['roofline']['unoverlapped_ms']
for k,v in sorted(u.items(), key=lambda x:-x[1]): print('   %-44s %8.2f  %s'%(k,v,d['roofline'].get('unoverlapped_gcells_per_s',{}).get(k,'')))
EOF2
head -22 $O/r03_bench_full_kernel_stats_$V.txt | cut -c1-150
