# Round-end evidence in one gpurun call: GPU test suite, smoke(), the three bench lines, rocprofv3 kernel summaries, PMC traffic.
# usage (from the dev container): gpurun --timeout 2400 -- 'bash tools/profile_round.sh v5'
V=${1:-vX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 > $O/r01_pytest_gpu_$V.log; python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/r01_pytest_gpu_$V.log 2>&1)
python $R/bench.py > $O/r01_bench_full_$V.json 2> $O/r01_bench_full_$V.log
rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --no-cpu-baseline > $O/r01_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r01_bench_full_kernel_stats_$V.txt; rm -rf $O/prof_ont
cd $R; python tools/pmc_traffic.py --reads 20000 > $O/pmc_ont.log 2>&1; python tools/pmc_traffic.py --preset splice --reads 10000 > $O/pmc_spl.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json; cd /tmp
python $R/bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --cpu-sample 3000 > $O/r01_bench_splice_$V.json 2> $O/r01_bench_splice_$V.log
rocprofv3 --kernel-trace --stats -d $O/prof_spl -o bench -- python $R/bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --no-cpu-baseline > $O/r01_bench_splice_${V}_under_rocprof.json 2> $O/prof_spl.log
python $R/tools/rocpd_summary.py $(ls $O/prof_spl/*.db $O/prof_spl/*/*.db 2>/dev/null | head -1) > $O/r01_bench_splice_kernel_stats_$V.txt; rm -rf $O/prof_spl
cat $O/r01_pytest_gpu_$V.log; python -c "
import json,sys
for f in ['r01_bench_full_$V.json','r01_bench_splice_$V.json']:
    d=json.load(open('$O/'+f)); print(f, d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['hits_identical_to_gpu'], d['roofline']['achieved'], d['roofline']['traffic'])"
