R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_full_v4.json 2> $O/r02_bench_full_v4.log; python -c "
import json; d=json.load(open('$O/r02_bench_full_v4.json')); print('full', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['valu'])"
timeout 500 python $R/bench.py --preset map-hifi --reads 200000 --steps 2 --warmup 1 --cpu-sample 20000 > $O/r02_bench_hifi_v1.json 2> $O/r02_bench_hifi_v1.log; python -c "
import json; d=json.load(open('$O/r02_bench_hifi_v1.json')); print('hifi', d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['hits_identical_to_gpu'])"; tail -2 $O/r02_bench_hifi_v1.log | cut -c1-300
timeout 400 python $R/bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --cpu-sample 3000 > $O/r02_bench_splice_v2.json 2> $O/r02_bench_splice_v2.log; python -c "
import json; d=json.load(open('$O/r02_bench_splice_v2.json')); print('splice', d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['hits_identical_to_gpu'])"
