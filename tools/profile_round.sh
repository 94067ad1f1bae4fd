# Round-end evidence in one gpurun call: smoke(), the bench lines, the rocprofv3 kernel summary of the headline command.
# usage (from the dev container): gpurun --timeout 1500 -- 'bash tools/profile_round.sh v5'   (SHORT=1: without the splice and sr lines)
V=${1:-vX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_smoke_$V.log 2>&1; tail -1 $O/r02_smoke_$V.log)
timeout 400 python $R/bench.py --steps 10 --warmup 3 > $O/r02_bench_full_$V.json 2> $O/r02_bench_full_$V.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r02_bench_full_kernel_stats_$V.txt; rm -rf $O/prof_ont
timeout 400 python $R/bench.py --preset map-hifi --reads 200000 --steps 2 --warmup 1 --cpu-sample 20000 > $O/r02_bench_hifi_$V.json 2> $O/r02_bench_hifi_$V.log
if [ -z "$SHORT" ]; then
timeout 400 python $R/bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --cpu-sample 3000 > $O/r02_bench_splice_$V.json 2> $O/r02_bench_splice_$V.log
timeout 400 python $R/bench.py --preset sr --reads 1000000 --steps 2 --warmup 1 --cpu-sample 100000 > $O/r02_bench_sr_$V.json 2> $O/r02_bench_sr_$V.log
fi
python -c "
import json,sys
for f in ['r02_bench_full_$V.json','r02_bench_full_${V}_under_rocprof.json','r02_bench_hifi_$V.json','r02_bench_splice_$V.json','r02_bench_sr_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']
        print(f, d['value'], d['ms_per_step'], c.get('value'), c.get('hits_identical_to_gpu'), r['kernel'], r['achieved'], r['traffic'], (r.get('valu') or {}).get('frac'))
    except Exception as e: print(f, 'FAILED', e)"
head -12 $O/r02_bench_full_kernel_stats_$V.txt
