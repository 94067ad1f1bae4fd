cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline"
O=$GRAFT_REPO_ROOT/gpurun_out
T=$1
$B > $O/r04_bench_a_$T.json 2> $O/r04_bench_a_$T.log
for f in a; do python - $O/r04_bench_${f}_$T.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); c=d['config']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'cpu_s', c['host_cpu_s_per_step'], 'resident', c['resident_gbases_per_s'], 'text', c['pipeline_text_identical'])
print('   by thread:', c.get('host_cpu_s_per_step_by_thread_name')); print('   by stage (one lane):', c.get('host_cpu_s_per_stage_one_lane_pass')); print('   lane drivers, last timed batch:', c.get('lane_driver_cpu_s_last_timed_batch'))
PY
done
