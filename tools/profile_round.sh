R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --no-cpu-baseline > $O/r01_bench_full_v4_under_rocprof.json 2> $O/prof_ont.log
python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r01_bench_full_kernel_stats_v4.txt; rm -rf $O/prof_ont
rocprofv3 --kernel-trace --stats -d $O/prof_spl -o bench -- python $R/bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --no-cpu-baseline > $O/r01_bench_splice_v2_under_rocprof.json 2> $O/prof_spl.log
python $R/tools/rocpd_summary.py $(ls $O/prof_spl/*.db $O/prof_spl/*/*.db 2>/dev/null | head -1) > $O/r01_bench_splice_kernel_stats_v2.txt; rm -rf $O/prof_spl
cd $R; python tools/pmc_traffic.py --reads 20000 > $O/pmc_ont.log 2>&1; python tools/pmc_traffic.py --preset splice --reads 10000 > $O/pmc_spl.log 2>&1; cp profiles/pmc_traffic.json $O/pmc_traffic.json
head -12 $O/r01_bench_full_kernel_stats_v4.txt; head -8 $O/r01_bench_splice_kernel_stats_v2.txt; tail -3 $O/pmc_spl.log
