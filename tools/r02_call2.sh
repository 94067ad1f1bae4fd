# Round 2, GPU call 2: issue-rate microbenchmark, the opt-in device paths, un-overlapped kernel times (MM2AMD_LANES=1) under rocprofv3, SQ counters.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 tools/build/valu_issue_bench > $O/r02_valu_issue_bench.txt 2>&1; tail -25 $O/r02_valu_issue_bench.txt
MM2AMD_PENDING=1 timeout 500 python -u -m pytest -p no:cacheprovider -m gpu -q tests/test_gpu_pending.py tests/test_gpu_shortreads.py::test_python_map_pairs_equals_the_reference_sam tests/test_gpu_dropin.py::test_single_anchor_chains_identical > $O/r02_pending_gpu_v2.log 2>&1
tail -15 $O/r02_pending_gpu_v2.log
cd /tmp; export TMPDIR=/tmp
MM2AMD_LANES=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_l1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r02_bench_lanes1_v1_under_rocprof.json 2> $O/prof_l1.log
python $R/tools/rocpd_summary.py $(ls $O/prof_l1/*.db $O/prof_l1/*/*.db 2>/dev/null | head -1) > $O/r02_bench_lanes1_kernel_stats_v1.txt; rm -rf $O/prof_l1
head -30 $O/r02_bench_lanes1_kernel_stats_v1.txt
cd $R
PMC_SQ_TAG=a timeout 300 python tools/pmc_sq.py SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY > $O/r02_pmc_sq_a.txt 2>&1; cat $O/r02_pmc_sq_a.txt
PMC_SQ_TAG=b timeout 300 python tools/pmc_sq.py SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM > $O/r02_pmc_sq_b.txt 2>&1; cat $O/r02_pmc_sq_b.txt
