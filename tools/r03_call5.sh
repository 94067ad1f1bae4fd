# Round 3, fifth GPU call (measurement only): where the anchor sort's time goes (with and without the tie replay; SQ counters), when the pipeline's steps run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
MM2AMD_BENCH_TRACE=1 timeout 300 python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r03_bench_trace_v5.json 2> $O/r03_bench_trace_v5.log; grep "batch\|steps in" $O/r03_bench_trace_v5.log | tail -22 | cut -c1-200
MM2AMD_SORT_NO_REPLAY=1 timeout 300 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r03_bench_noreplay_v5.json 2> $O/r03_bench_noreplay_v5.log
python - <<EOF
import json
for f in ['r03_bench_trace_v5.json','r03_bench_noreplay_v5.json']:
    d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], d['config'].get('resident_gbases_per_s'), {k:v for k,v in r['unoverlapped_ms'].items() if 'sort' in k or 'chain' in k})
EOF
(cd $R && PMC_SQ_TAG=r03_v5 timeout 400 python tools/pmc_sq.py SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --reads 20000 2>&1 | grep "anchor_sort\|chain_\|sketch\|seed_" | cut -c1-330)
