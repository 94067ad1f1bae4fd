R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -u -m pytest -p no:cacheprovider -m gpu -q -x tests/test_gpu_ksw.py tests/test_gpu_dropin.py tests/test_gpu_aligner.py > $O/r02_pytest_gpu_v7.log 2>&1; tail -4 $O/r02_pytest_gpu_v7.log
cd /tmp; export TMPDIR=/tmp
run() { n=$1; shift; r=$1; shift
  env "$@" timeout 200 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --reads $r > $O/ss_$n.json 2> $O/ss_$n.log
  python -c "
import json; d=json.load(open('$O/ss_$n.json')); print('$n', d['value'], d['ms_per_step'])"
}
run full 100000 A=1
run r12k_def 12500 MM2AMD_TRACE=$O/trace_12k.txt
run r12k_62M 12500 MM2AMD_SUBBATCH_BASES=62500000
run r12k_t32 12500 MM2AMD_SUBBATCH_BASES=62500000 
run r25k_62M 25000 MM2AMD_SUBBATCH_BASES=62500000
run r25k_def 25000 A=1
