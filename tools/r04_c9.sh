cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline"
O=$GRAFT_REPO_ROOT/gpurun_out
T=$1; shift
show() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'cpu_s', c['host_cpu_s_per_step'], 'resident', c['resident_gbases_per_s'], 'text', c['pipeline_text_identical'])
print('   by thread:', c.get('host_cpu_s_per_step_by_thread_name')); print('   by stage (one lane):', c.get('host_cpu_s_per_stage_one_lane_pass')); print('   lane drivers, last timed batch:', c.get('lane_driver_cpu_s_last_timed_batch'))
PY
}
for V in "$@"; do
  case $V in
    a) $B > $O/r04_bench_a_$T.json 2> $O/r04_bench_a_$T.log; show $O/r04_bench_a_$T.json ;;
    devfin7) MM2AMD_LANES=7 MM2AMD_DEVICE_FINISH=1 $B > $O/r04_bench_devfin7_$T.json 2> $O/r04_bench_devfin7_$T.log; show $O/r04_bench_devfin7_$T.json ;;
    noearly) MM2AMD_NO_EARLY_START=1 $B > $O/r04_bench_noearly_$T.json 2> $O/r04_bench_noearly_$T.log; show $O/r04_bench_noearly_$T.json ;;
    env:*) E=${V#env:}; N=$(echo $E | tr -c 'A-Za-z0-9\n' '_'); env $(echo $E | tr ',' ' ') $B > $O/r04_bench_${N}_$T.json 2> $O/r04_bench_${N}_$T.log; show $O/r04_bench_${N}_$T.json ;;
    devfin) MM2AMD_DEVICE_FINISH=1 $B > $O/r04_bench_devfin_$T.json 2> $O/r04_bench_devfin_$T.log; show $O/r04_bench_devfin_$T.json ;;
    lanes*) MM2AMD_LANES=${V#lanes} $B > $O/r04_bench_${V}_$T.json 2> $O/r04_bench_${V}_$T.log; show $O/r04_bench_${V}_$T.json ;;
    t*) $B --threads ${V#t} > $O/r04_bench_${V}_$T.json 2> $O/r04_bench_${V}_$T.log; show $O/r04_bench_${V}_$T.json ;;
  esac
done
