# Round 3, first GPU call: the GPU suite with the new parity cases, this round's starting bench line, SQ counters of every kernel of the path
# (ksw_stream_kernel included), FETCH/WRITE calibration for our access patterns, HBM traffic at the bench's launch size, the per-SIMD issue-rate table.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
(cd $R && timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > $O/r03_pytest_gpu_v1.log; tail -3 $O/r03_pytest_gpu_v1.log
timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_full_v0.json 2> $O/r03_bench_full_v0.log; tail -c 600 $O/r03_bench_full_v0.json
timeout 120 $R/tools/build/valu_issue_bench > $O/r03_valu_issue_bench_v1.txt 2>&1; head -8 $O/r03_valu_issue_bench_v1.txt | cut -c1-400
timeout 300 python $R/tools/pmc_calib.py > $O/r03_pmc_calibration.json 2> $O/pmc_calib.err; tail -3 $O/pmc_calib.err; head -c 1500 $O/r03_pmc_calibration.json
(cd $R && PMC_SQ_TAG=r03_v1 timeout 400 python tools/pmc_sq.py SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --reads 20000 2>&1 | tail -20)
(cd $R && timeout 600 python tools/pmc_traffic.py --reads 100000 --out $O/r03_pmc_traffic_v1.json 2>&1 | tail -5)
