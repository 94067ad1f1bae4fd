R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift
  env "$@" timeout 200 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/sz_$n.json 2> $O/sz_$n.log
  python -c "
import json; d=json.load(open('$O/sz_$n.json')); print('$n', d['value'], d['ms_per_step'])"
}
run sub200M MM2AMD_SUBBATCH_BASES=200000000
run sub125M MM2AMD_SUBBATCH_BASES=125000000
run sub100M A=1
# two in-process replicas on one device, with the parity check of the cpu_baseline leg
MM2AMD_GPUS=2 MM2AMD_DEVICE_IDS=0,0 timeout 300 python $R/bench.py --reads 30000 --steps 2 --warmup 1 --cpu-sample 6000 > $O/rep2.json 2> $O/rep2.log; python -c "
import json; d=json.load(open('$O/rep2.json')); print('2 replicas on one device', d['value'], d['ms_per_step'], d['cpu_baseline']['hits_identical_to_gpu'])"
# the torchrun path (strong scaling, one batch sharded over the ranks) with two ranks sharing the one GPU (gloo: plumbing check)
MM2AMD_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --steps 2 --warmup 1 --reads 20000 --ref-mb 500 > $O/tr2.json 2> $O/tr2.log; tail -1 $O/tr2.json | cut -c1-700; tail -3 $O/tr2.log | cut -c1-300
