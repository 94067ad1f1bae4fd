# Round 5: one gpurun call = a list of steps.   usage: gpurun --timeout N -- 'bash tools/r05_call.sh TAG step [step ...]'
#   suite          the whole -m gpu suite                     suite:<expr>   pytest -k <expr>
#   bench          the driver's command (20 steps, 5 warm-up)  bench:<K>      K steps, 3 warm-up
#   prof           rocprofv3 --kernel-trace --stats of bench (5 steps), summary -> r05_bench_full_kernel_stats_TAG.txt
#   pmc            tools/pmc_traffic.py (FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/pmc_traffic_TAG.json
#   rank8          bench.py --as-rank-of 8 (8 steps): one rank's share under 1/8 of the CPU quota
#   pmcsq          tools/pmc_sq.py (SQ counters of the DP kernels)   hifi / splice   BASELINE.json configs[3] / [4], driver-shaped line
#   benchenv:NAME:VAR=VAL   8 steps with an environment variable set (A/B inside one call) -> r05_bench_NAME_TAG.json
#   e2e:N          tools/e2e_wall.py on N reads: the unmodified minimap2 binary against the drop-in pipeline driver, same .mmi, SAM compared
#   smoke          __graft_entry__.smoke()
#   sh:<command>   anything else, from the repo root
V=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']; u=r.get('unoverlapped_ms') or {}
    print(sys.argv[1].split('/')[-1], d['value'], 'ms/step', d['ms_per_step'], 'resident', d['config'].get('resident_gbases_per_s'), 'cpu_s', d['config']['host_cpu_s_per_step'],
          'text_identical', d['config'].get('pipeline_text_identical'), 'ref', c.get('value'), c.get('hits_identical_to_gpu'), 'valu', (r.get('valu') or {}).get('frac'))
    fam={}
    for k,v in u.items(): fam[k.split('[')[0].split('<')[0]]=fam.get(k.split('[')[0].split('<')[0],0)+v
    print('  unoverlapped ms:', {k:round(v,1) for k,v in sorted(fam.items())}, 'sum %.0f'%sum(u.values()))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
for S in "$@"; do
  cd $R
  case "$S" in
    suite)   (timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -12) > $O/r05_pytest_gpu_$V.log; tail -3 $O/r05_pytest_gpu_$V.log ;;
    suite:*) (timeout 1200 python -m pytest tests -x -q -m gpu -k "${S#suite:}" 2>&1 | tail -12) > $O/r05_pytest_gpu_k_$V.log; tail -3 $O/r05_pytest_gpu_k_$V.log ;;
    bench)   cd /tmp; timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_full_$V.json 2> $O/r05_bench_full_$V.log; line $O/r05_bench_full_$V.json ;;
    bench:*) cd /tmp; timeout 900 python $R/bench.py --steps ${S#bench:} --warmup 3 > $O/r05_bench_full_$V.json 2> $O/r05_bench_full_$V.log; line $O/r05_bench_full_$V.json ;;
    benchenv:*) X=${S#benchenv:}; NM=${X%%:*}; EV=${X#*:}; cd /tmp; env $EV timeout 900 python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/r05_bench_${NM}_$V.json 2> $O/r05_bench_${NM}_$V.log; line $O/r05_bench_${NM}_$V.json ;;
    prof)    cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 8 --warmup 2 --timed-only > $O/r05_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
             python $R/tools/rocpd_summary.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) > $O/r05_bench_full_kernel_stats_$V.txt
             python $R/tools/exposed_time.py $(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1) 3.0 > $O/r05_exposed_time_$V.txt 2>&1; cat $O/r05_exposed_time_$V.txt; rm -rf $O/prof_ont; head -14 $O/r05_bench_full_kernel_stats_$V.txt ;;
    pmc)     cd /tmp; MM2AMD_COMMIT=$V timeout 900 python $R/tools/pmc_traffic.py --out $O/pmc_traffic_$V.json > $O/pmc_traffic_$V.log 2>&1; tail -3 $O/pmc_traffic_$V.log | cut -c1-300 ;;
    rank8)   cd /tmp; timeout 900 python $R/bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --as-rank-of 8 > $O/r05_bench_rank8_$V.json 2> $O/r05_bench_rank8_$V.log; line $O/r05_bench_rank8_$V.json
             python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split(chr(10))[-1]); print('  as_rank_of', d['config']['as_rank_of'])" $O/r05_bench_rank8_$V.json ;;
    pmcsq)   cd /tmp; PMC_SQ_TAG=$V timeout 600 python $R/tools/pmc_sq.py SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > $O/r05_pmc_sq_$V.txt 2>&1; mv $O/pmc_sq_$V.json $O/r05_pmc_sq_$V.json; tail -8 $O/r05_pmc_sq_$V.txt | cut -c1-200 ;;
    hifi)    cd /tmp; timeout 900 python $R/bench.py --preset map-hifi --reads 200000 --steps 3 --warmup 2 > $O/r05_bench_hifi_$V.json 2> $O/r05_bench_hifi_$V.log; line $O/r05_bench_hifi_$V.json ;;
    splice)  cd /tmp; timeout 900 python $R/bench.py --preset splice --reads 50000 --steps 3 --warmup 1 > $O/r05_bench_splice_$V.json 2> $O/r05_bench_splice_$V.log; line $O/r05_bench_splice_$V.json ;;
    e2e:*)   cd /tmp; df -h /tmp | tail -1; timeout 1500 python $R/tools/e2e_wall.py --reads ${S#e2e:} --out $O/r05_e2e_wall_$V.json > $O/r05_e2e_wall_$V.log 2>&1; tail -1 $O/r05_e2e_wall_$V.log | cut -c1-1500; rm -rf /tmp/e2e ;;
    smoke)   python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke_$V.log 2>&1; tail -1 $O/r05_smoke_$V.log ;;
    sh:*)    bash -c "${S#sh:}" ;;
  esac
done
