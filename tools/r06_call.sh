# Round-6 GPU calls (one gpurun call each): bash tools/r06_call.sh <tag> <what...>
#   ksw    : the kernel-level GPU cases (tests/test_gpu_ksw.py)
#   ab     : the headline bench with and without the banded gap fill, short (no CPU baseline)
#   bench  : the headline bench as the driver runs it
#   tests  : the whole GPU suite
V=${1:-vX}; shift; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
for what in "$@"; do
case $what in
ksw)   timeout 900 python -m pytest tests/test_gpu_ksw.py -x -q -m gpu > $O/r06_pytest_ksw_$V.log 2>&1; tail -3 $O/r06_pytest_ksw_$V.log ;;
tests) timeout 2400 python -m pytest tests -x -q -m gpu > $O/r06_pytest_gpu_$V.log 2>&1; tail -3 $O/r06_pytest_gpu_$V.log ;;
ab)    MM2AMD_BAND_DEBUG=1 timeout 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_band_$V.json 2> $O/r06_bench_band_$V.log
       MM2AMD_NO_BAND=1 timeout 500 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_noband_$V.json 2> $O/r06_bench_noband_$V.log
       grep -h "band:" $O/r06_bench_band_$V.log | tail -3
       python - <<P
import json
for f in ['r06_bench_band_$V.json','r06_bench_noband_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); r=d['roofline']
        print(f, d['value'], d['ms_per_step'], r['kernel'], r.get('unoverlapped_step_ms'))
        u=r.get('unoverlapped_ms') or {}
        print('   ', {k: v for k, v in u.items() if k.startswith('ksw')})
    except Exception as e: print(f, 'FAILED', e)
P
       ;;
bench) timeout 900 python bench.py --steps 20 --warmup 5 > $O/r06_bench_full_$V.json 2> $O/r06_bench_full_$V.log; tail -c 600 $O/r06_bench_full_$V.json ;;
e2e)   # SURVEY 8(d)'s primary figure at scale: the reference's reader and writer around the GPU path, 10 Gbases, SAM digest vs the minimap2 binary, index digest vs mm_idx_gen
       timeout 2400 python tools/e2e_wall.py --reads ${E2E_READS:-1000000} --out $O/r06_e2e_wall_$V.json > $O/r06_e2e_wall_$V.log 2>&1; tail -c 1500 $O/r06_e2e_wall_$V.log ;;
arena) # the first batches with and without the arenas behind the work buffers
       for m in arena noarena; do
         if [ $m = noarena ]; then export MM2AMD_NO_ARENA=1; else unset MM2AMD_NO_ARENA; fi
         timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --timed-only > $O/r06_bench_${m}_$V.json 2> $O/r06_bench_${m}_$V.log
         grep -h "warmup\|steps in" $O/r06_bench_${m}_$V.log | cut -c1-330
       done; unset MM2AMD_NO_ARENA ;;
rank8) # one rank's share of an eight-rank strong-scaling job on this one GPU with an eighth of the CPU quota (the prediction of DESIGN.md section 6)
       timeout 600 python bench.py --as-rank-of 8 --steps 8 --warmup 3 --no-cpu-baseline > $O/r06_bench_rank8_$V.json 2> $O/r06_bench_rank8_$V.log
       python -c "
import json; d=json.loads(open('$O/r06_bench_rank8_$V.json').read().strip().split('\n')[-1]); print('rank8', d['value'], d['config']['as_rank_of'])" ;;
rank8trace) MM2AMD_TRACE=$O/r06_rank8_trace_$V.tsv timeout 600 python bench.py --as-rank-of 8 --steps 3 --warmup 2 --no-cpu-baseline > $O/r06_bench_rank8t_$V.json 2> $O/r06_bench_rank8t_$V.log
       python tools/trace_summary.py $O/r06_rank8_trace_$V.tsv 0.5; python tools/trace_ascii.py $O/r06_rank8_trace_$V.tsv --win 0.2 --res 1 2>/dev/null | head -40; tail -c 3000000 $O/r06_rank8_trace_$V.tsv > $O/r06_rank8_trace_tail_$V.tsv; rm -f $O/r06_rank8_trace_$V.tsv ;;
repsplit) MM2AMD_KSW_SPLIT_RINGS=1 timeout 900 python bench.py --workload repeats --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_repeats_split_$V.json 2> $O/r06_bench_repeats_split_$V.log
       python -c "
import json; d=json.loads(open('$O/r06_bench_repeats_split_$V.json').read().strip().split('\n')[-1]); print('repeats, ring classes apart', d['value'], d['ms_per_step'])" ;;
abband4) # windows beyond 512 x 512 in a band of 512 diagonals (the four-set class) against the strip kernel's rectangles: the headline and the repeats workload
       for m in band4 noband4; do
         if [ $m = noband4 ]; then export MM2AMD_BAND_MAX=512; else unset MM2AMD_BAND_MAX; fi
         for w in ont repeats; do
           wl=""; [ $w = repeats ] && wl="--workload repeats"
           MM2AMD_BAND_DEBUG=1 timeout 900 python bench.py $wl --steps 8 --warmup 4 --no-cpu-baseline > $O/r06_bench_${w}_${m}_$V.json 2> $O/r06_bench_${w}_${m}_$V.log
           grep -h "band:" $O/r06_bench_${w}_${m}_$V.log | tail -2 | cut -c1-200
           python - <<P
import json
d=json.loads(open('$O/r06_bench_${w}_${m}_$V.json').read().strip().split('\n')[-1]); r=d['roofline']; u=r.get('unoverlapped_ms') or {}
print('$w $m', d['value'], d['ms_per_step'], 'unoverlapped', r.get('unoverlapped_step_ms'), {k: v for k, v in u.items() if k.startswith('ksw_band') or k.startswith('ksw_gapfill') or k.startswith('ksw_stream')})
print('    ', d['config'].get('banded_gap_fill'))
P
         done
       done; unset MM2AMD_BAND_MAX ;;
rank8subs) # one rank's share of eight: how many sub-batches (= lanes in flight) its 12.5 k reads are cut into
       for n in 4 6 8 12 16; do
         MM2AMD_MIN_SUBBATCHES=$n timeout 600 python bench.py --as-rank-of 8 --steps 8 --warmup 3 --no-cpu-baseline > $O/r06_bench_rank8_subs${n}_$V.json 2> $O/r06_bench_rank8_subs${n}_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_rank8_subs${n}_$V.json').read().strip().split('\n')[-1]); a=d['config']['as_rank_of']; print('min sub-batches $n:', d['value'], 'share ms', a['ms_per_step'], 'predicted', a['predicted_strong_scaling'], 'host cpu s/Gbase', a['host_cpu_s_per_gbase'])"
       done ;;
srsweep) # short reads are host-bound: lanes / sub-batch size
       for cfg in "8 100000000" "8 40000000" "8 25000000" "12 25000000" "16 15000000"; do
         set -- $cfg
         MM2AMD_LANES=$1 MM2AMD_SUBBATCH_BASES=$2 timeout 600 python bench.py --preset sr --reads 1000000 --steps 3 --warmup 1 --no-cpu-baseline --timed-only > $O/r06_bench_sr_l$1_s$2_$V.json 2> $O/r06_bench_sr_l$1_s$2_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_sr_l$1_s$2_$V.json').read().strip().split('\n')[-1]); print('sr lanes $1 sub-batch bases $2:', d['value'], d['ms_per_step'], d['config']['host_cpu_s_per_step'])"
       done ;;
rmqcap) # heavy reads' long-join re-chaining on the host's tree instead of one wavefront walking 100 000 anchors
       for cap in 131072 30000 8000; do
         MM2AMD_RMQ_DEV_MAX_ANCHORS=$cap timeout 900 python bench.py --workload repeats --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_repeats_cap${cap}_$V.json 2> $O/r06_bench_repeats_cap${cap}_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_repeats_cap${cap}_$V.json').read().strip().split('\n')[-1]); u=d['roofline']['unoverlapped_ms']; print('cap $cap:', d['value'], d['ms_per_step'], 'host cpu', d['config']['host_cpu_s_per_step'], 'rmq[lj]', u.get('chain_rmq_kernel[long-join]'), d['config']['device_path_last_batch'])"
       done ;;
sweep) # lanes x DP gate
       for cfg in "8 4" "8 3" "8 6" "6 4" "10 4" "10 6" "12 6"; do
         set -- $cfg
         MM2AMD_LANES=$1 MM2AMD_DP_GATE=$2 timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --timed-only > $O/r06_bench_l$1_g$2_$V.json 2> $O/r06_bench_l$1_g$2_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_l$1_g$2_$V.json').read().strip().split('\n')[-1]); print('lanes $1 gate $2:', d['value'], d['ms_per_step'])"
       done ;;
abside2) # the strip kernel's gap fills on a stream of their own against in front of the banded kernel's launches: whole batch and one rank's share of eight
       for m in side2 noside2; do
         if [ $m = side2 ]; then export MM2AMD_SIDE2=1; else unset MM2AMD_SIDE2; fi
         timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --as-rank-of 8 > $O/r06_bench_${m}_$V.json 2> $O/r06_bench_${m}_$V.log
         python - <<P
import json
d=json.loads(open('$O/r06_bench_${m}_$V.json').read().strip().split('\n')[-1]); r=d['roofline']; a=d['config']['as_rank_of']
print('$m', d['value'], d['ms_per_step'], 'share ms', a['ms_per_step'], 'predicted', a['predicted_strong_scaling'])
P
       done; unset MM2AMD_SIDE2 ;;
abfirst) # index probes through the per-bucket first-key record against bucket_start -> slots only
       for m in first nofirst; do
         if [ $m = nofirst ]; then export MM2AMD_NO_FIRST_SLOT=1; else unset MM2AMD_NO_FIRST_SLOT; fi
         timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/r06_bench_${m}_$V.json 2> $O/r06_bench_${m}_$V.log
         python - <<P
import json
d=json.loads(open('$O/r06_bench_${m}_$V.json').read().strip().split('\n')[-1]); r=d['roofline']
print('$m', d['value'], d['ms_per_step'], r.get('unoverlapped_step_ms'), 'seed_collect', (r.get('unoverlapped_ms') or {}).get('seed_collect_kernel'), r.get('index_probes', {}).get('per_s_unoverlapped'), 'index build', d['config']['index_build_s'])
P
       done; unset MM2AMD_NO_FIRST_SLOT ;;
abring) # the lane-exact kernel's ring classes 512 / 1024 / 2048 merged into one launch class against apart
       for m in merged split; do
         if [ $m = split ]; then export MM2AMD_KSW_SPLIT_RINGS=1; else unset MM2AMD_KSW_SPLIT_RINGS; fi
         timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/r06_bench_rings_${m}_$V.json 2> $O/r06_bench_rings_${m}_$V.log
         python - <<P
import json
d=json.loads(open('$O/r06_bench_rings_${m}_$V.json').read().strip().split('\n')[-1]); r=d['roofline']
print('$m', d['value'], d['ms_per_step'], r['kernel'], r.get('unoverlapped_step_ms'))
print('   ', {k: v for k, v in (r.get('unoverlapped_ms') or {}).items() if k.startswith('ksw_ext')})
P
       done; unset MM2AMD_KSW_SPLIT_RINGS ;;
abext) # the extension kernel with the query across the lanes against round 3's (the target across the lanes)
       for m in extq bytarget; do
         if [ $m = bytarget ]; then export MM2AMD_EXT_BY_TARGET=1; else unset MM2AMD_EXT_BY_TARGET; fi
         timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/r06_bench_${m}_$V.json 2> $O/r06_bench_${m}_$V.log
         python - <<P
import json
d=json.loads(open('$O/r06_bench_${m}_$V.json').read().strip().split('\n')[-1]); r=d['roofline']
print('$m', d['value'], d['ms_per_step'], r['kernel'], r.get('unoverlapped_step_ms'))
print('   ', {k: v for k, v in (r.get('unoverlapped_ms') or {}).items() if k.startswith('ksw')})
P
       done; unset MM2AMD_EXT_BY_TARGET ;;
pieces) # heavy reads chained in pieces (and chain_rmq_kernel's small first launch) against a wavefront per read: the repeats workload, then the headline
       for m in pieces nopieces; do
         if [ $m = nopieces ]; then export MM2AMD_CHAIN_PIECE=0 MM2AMD_RMQ_PIECE=0; else unset MM2AMD_CHAIN_PIECE MM2AMD_RMQ_PIECE; fi
         MM2AMD_PIECE_DEBUG=1 timeout 900 python bench.py --workload repeats --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_repeats_${m}_$V.json 2> $O/r06_bench_repeats_${m}_$V.log
         grep -h "work list" $O/r06_bench_repeats_${m}_$V.log | sort | uniq -c | sort -rn | head -4
         python - <<P
import json
d=json.loads(open('$O/r06_bench_repeats_${m}_$V.json').read().strip().split('\n')[-1]); c=d['config']; r=d['roofline']
print('repeats $m', d['value'], d['ms_per_step'], 'host cpu', c['host_cpu_s_per_step'], c['device_path_last_batch'])
print('   ', {k: v for k, v in sorted((r.get('unoverlapped_ms') or {}).items(), key=lambda kv: -kv[1])[:10]})
P
       done; unset MM2AMD_CHAIN_PIECE MM2AMD_RMQ_PIECE
       timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > $O/r06_bench_ont_pieces_$V.json 2> $O/r06_bench_ont_pieces_$V.log
       python -c "
import json; d=json.loads(open('$O/r06_bench_ont_pieces_$V.json').read().strip().split('\n')[-1]); print('ont', d['value'], d['ms_per_step'])" ;;
ring)  # chain_fill_kernel's LDS window of 128 entries (eight workgroups per CU) against 256 (five); then what chain_rmq_kernel's wavefronts spend (MM2AMD_RMQ_TIMING)
       for m in 256 128; do
         for w in repeats ont; do
           wl=""; [ $w = repeats ] && wl="--workload repeats"
           MM2AMD_CHAIN_RING=$m timeout 900 python bench.py $wl --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_${w}_ring${m}_$V.json 2> $O/r06_bench_${w}_ring${m}_$V.log
           python - <<P
import json
d=json.loads(open('$O/r06_bench_${w}_ring${m}_$V.json').read().strip().split('\n')[-1]); u=d['roofline'].get('unoverlapped_ms') or {}
print('$w ring $m', d['value'], d['ms_per_step'], 'chain_fill', u.get('chain_fill_kernel'))
P
         done
       done
       MM2AMD_RMQ_TIMING=1 MM2AMD_LANES=1 timeout 900 python bench.py --workload repeats --steps 1 --warmup 1 --no-cpu-baseline --timed-only > $O/r06_bench_repeats_rmqtiming_$V.json 2> $O/r06_bench_repeats_rmqtiming_$V.log
       grep -h "chain_rmq_kernel:" $O/r06_bench_repeats_rmqtiming_$V.log | tail -12 | cut -c1-420 ;;
wide)  # chain_rmq_kernel's long clusters by workgroups (chain_rmq_wide_kernel) against by one wavefront: the repeats workload, with the per-wavefront timing of one lane
       for m in wide nowide; do
         if [ $m = nowide ]; then export MM2AMD_RMQ_DENSE=0; else unset MM2AMD_RMQ_DENSE; fi
         timeout 900 python bench.py --workload repeats --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_repeats_${m}_$V.json 2> $O/r06_bench_repeats_${m}_$V.log
         python - <<P
import json
d=json.loads(open('$O/r06_bench_repeats_${m}_$V.json').read().strip().split('\n')[-1]); c=d['config']; r=d['roofline']
print('repeats $m', d['value'], d['ms_per_step'], 'host cpu', c['host_cpu_s_per_step'], c['device_path_last_batch'])
print('   ', {k: v for k, v in sorted((r.get('unoverlapped_ms') or {}).items(), key=lambda kv: -kv[1])[:10]})
P
       done; unset MM2AMD_RMQ_DENSE
       MM2AMD_RMQ_TIMING=1 MM2AMD_LANES=1 timeout 900 python bench.py --workload repeats --steps 1 --warmup 1 --no-cpu-baseline --timed-only > $O/r06_bench_repeats_rmqtiming_$V.json 2> $O/r06_bench_repeats_rmqtiming_$V.log
       grep -h "chain_rmq_kernel:" $O/r06_bench_repeats_rmqtiming_$V.log | tail -4 | cut -c1-420 ;;
widesweep) # from how many anchors a cluster goes to a workgroup, and from how many to sixteen wavefronts
       for cfg in "1024 4" "1024 2" "512 4" "512 2" "768 3"; do
         set -- $cfg
         MM2AMD_RMQ_DENSE=$1 MM2AMD_RMQ_DENSE_BIG=$2 timeout 900 python bench.py --workload repeats --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_repeats_dense$1x$2_$V.json 2> $O/r06_bench_repeats_dense$1x$2_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_repeats_dense$1x$2_$V.json').read().strip().split('\n')[-1]); u=d['roofline']['unoverlapped_ms']; print('dense $1 big x$2:', d['value'], d['ms_per_step'], 'rmq[lj]', u.get('chain_rmq_kernel[long-join]'))"
       done ;;
profrep) # the repeats workload's kernels one lane at a time under rocprofv3: which of the chaining launches lasts
       cd /tmp
       MM2AMD_LANES=1 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_rep -o bench -- python $R/bench.py --workload repeats --steps 2 --warmup 1 --no-cpu-baseline --timed-only > $O/r06_bench_repeats_${V}_under_rocprof.json 2> $O/prof_rep.log
       DB=$(ls $O/prof_rep/*.db $O/prof_rep/*/*.db 2>/dev/null | head -1)
       python $R/tools/rocpd_summary.py $DB > $O/r06_bench_repeats_kernel_stats_$V.txt; rm -rf $O/prof_rep
       head -24 $O/r06_bench_repeats_kernel_stats_$V.txt | cut -c1-200
       cd $R ;;
rank)  # the RMQ chainer's small neighbourhoods sorted by counting larger keys against the bitonic network
       for m in 256 0; do
         MM2AMD_RMQ_RANK_MAX=$m timeout 900 python bench.py --workload repeats --steps 4 --warmup 2 --no-cpu-baseline > $O/r06_bench_repeats_rank${m}_$V.json 2> $O/r06_bench_repeats_rank${m}_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_repeats_rank${m}_$V.json').read().strip().split('\n')[-1]); u=d['roofline']['unoverlapped_ms']; print('rank max $m:', d['value'], d['ms_per_step'], 'rmq[lj]', u.get('chain_rmq_kernel[long-join]'), d['config']['device_path_last_batch'])"
       done
       timeout 500 python bench.py --preset map-hifi --reads 200000 --steps 3 --warmup 2 --cpu-sample 20000 > $O/r06_bench_hifi_$V.json 2> $O/r06_bench_hifi_$V.log
       python -c "
import json; d=json.loads(open('$O/r06_bench_hifi_$V.json').read().strip().split('\n')[-1]); print('hifi', d['value'], d['ms_per_step'], (d.get('cpu_baseline') or {}).get('hits_identical_to_gpu'))" ;;
chain) timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_aligner.py tests/test_gpu_regions.py -x -q -m gpu > $O/r06_pytest_chain_$V.log 2>&1; tail -3 $O/r06_pytest_chain_$V.log ;;
prof)  # evidence at HEAD in one call: rocprofv3 kernel stats of the headline command, the exposed-time split, HBM traffic (FETCH / WRITE passes) and the SQ counters
       cd /tmp
       timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ont -o bench -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --timed-only > $O/r06_bench_full_${V}_under_rocprof.json 2> $O/prof_ont.log
       DB=$(ls $O/prof_ont/*.db $O/prof_ont/*/*.db 2>/dev/null | head -1)
       python $R/tools/rocpd_summary.py $DB > $O/r06_bench_full_kernel_stats_$V.txt; python $R/tools/exposed_time.py $DB 1.5 > $O/r06_exposed_time_$V.txt; rm -rf $O/prof_ont
       head -14 $O/r06_bench_full_kernel_stats_$V.txt; head -12 $O/r06_exposed_time_$V.txt
       cd $R
       timeout 900 python tools/pmc_traffic.py --out $O/pmc_traffic_$V.json > $O/pmc_traffic_$V.log 2>&1; tail -c 300 $O/pmc_traffic_$V.log
       PMC_SQ_TAG=r06_$V timeout 600 python tools/pmc_sq.py SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > $O/r06_pmc_sq_$V.txt 2>&1; grep "ksw_band\|ksw_extq\|ksw_extd2" $O/r06_pmc_sq_$V.txt | head -8 ;;
rccl)  timeout 300 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_aligner.py -x -q -m gpu > $O/r06_pytest_rccl_$V.log 2>&1; tail -3 $O/r06_pytest_rccl_$V.log ;;
repeats) # the side figure on a repeat- and SV-bearing reference: how much of a batch leaves the device path
       timeout 900 python bench.py --workload repeats --steps 6 --warmup 3 --cpu-sample 20000 > $O/r06_bench_repeats_$V.json 2> $O/r06_bench_repeats_$V.log
       python - <<P
import json
d=json.loads(open('$O/r06_bench_repeats_$V.json').read().strip().split('\n')[-1]); c=d['config']; r=d['roofline']
print('repeats', d['value'], d['ms_per_step'], c['device_path_last_batch'], c['banded_gap_fill'], (d.get('cpu_baseline') or {}).get('hits_identical_to_gpu'), (d.get('cpu_baseline') or {}).get('value'))
print('   ', {k: v for k, v in sorted((r.get('unoverlapped_ms') or {}).items(), key=lambda kv: -kv[1])[:12]})
P
       ;;
others) # the other BASELINE configurations: map-hifi, splice, sr
       timeout 500 python bench.py --preset map-hifi --reads 200000 --steps 3 --warmup 2 --cpu-sample 20000 > $O/r06_bench_hifi_$V.json 2> $O/r06_bench_hifi_$V.log
       timeout 500 python bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --cpu-sample 3000 > $O/r06_bench_splice_$V.json 2> $O/r06_bench_splice_$V.log
       timeout 500 python bench.py --preset sr --reads 1000000 --steps 2 --warmup 1 --cpu-sample 100000 > $O/r06_bench_sr_$V.json 2> $O/r06_bench_sr_$V.log
       python - <<P
import json
for f in ['r06_bench_hifi_$V.json','r06_bench_splice_$V.json','r06_bench_sr_$V.json']:
    try:
        d=json.loads(open('$O/'+f).read().strip().split('\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']
        print(f, d['value'], d['ms_per_step'], 'cpu', c.get('value'), c.get('hits_identical_to_gpu'), r['kernel'], 'host cpu s/step', d['config']['host_cpu_s_per_step'], d['config'].get('banded_gap_fill'))
        print('   ', {k: v for k, v in sorted((r.get('unoverlapped_ms') or {}).items(), key=lambda kv: -kv[1])[:8]})
    except Exception as e: print(f, 'FAILED', e)
P
       ;;
splice) # the splice gap-fill kernel: its kernel-level cases, the per-class microbenchmark, the splice configuration
       timeout 600 python -m pytest tests/test_gpu_ksw.py -x -q -m gpu -k splice 2>&1 | tail -2
       timeout 300 python tools/ksw_splice_microbench.py 16384 8000 > $O/r06_splice_micro_$V.txt 2>&1; cat $O/r06_splice_micro_$V.txt
       MM2AMD_KSW_CLASS_DEBUG=${SPLICE_CLASS_DEBUG:-0} timeout 500 python bench.py --preset splice --reads 50000 --steps 2 --warmup 1 --cpu-sample 3000 > $O/r06_bench_splice_$V.json 2> $O/r06_bench_splice_$V.log
       python - <<P
import json
d=json.loads(open('$O/r06_bench_splice_$V.json').read().strip().split('\\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']
print('splice', d['value'], d['ms_per_step'], 'cpu', c.get('value'), c.get('hits_identical_to_gpu'), 'host cpu s/step', d['config']['host_cpu_s_per_step'])
print('   ', {k: v for k, v in sorted((r.get('unoverlapped_ms') or {}).items(), key=lambda kv: -kv[1])[:6]})
P
       ;;
srdev) # short reads and pairs through the device region path (chains -> hits per segment -> windows -> DP -> consume -> finish) against the host's plan / consume rounds
       timeout 900 python -m pytest tests/test_gpu_shortreads.py tests/test_gpu_regions.py -x -q -m gpu 2>&1 | tail -3
       for m in dev host; do
         if [ $m = host ]; then export MM2AMD_DEVICE_REGIONS=0; else unset MM2AMD_DEVICE_REGIONS; fi
         cs="--no-cpu-baseline"; [ $m = dev ] && cs="--cpu-sample 100000"
         timeout 600 python bench.py --preset sr --reads 1000000 --steps 3 --warmup 1 $cs > $O/r06_bench_sr_${m}_$V.json 2> $O/r06_bench_sr_${m}_$V.log
         python - <<P
import json
d=json.loads(open('$O/r06_bench_sr_${m}_$V.json').read().strip().split('\\n')[-1]); c=d.get('cpu_baseline') or {}; r=d['roofline']; g=d['config']
print('sr $m', d['value'], d['ms_per_step'], 'cpu', c.get('value'), c.get('hits_identical_to_gpu'), 'host cpu s/step', g['host_cpu_s_per_step'], g.get('device_path_last_batch'))
print('   ', g.get('host_cpu_s_per_stage_one_lane_pass'))
print('   ', {k: v for k, v in sorted((r.get('unoverlapped_ms') or {}).items(), key=lambda kv: -kv[1])[:10]})
P
       done; unset MM2AMD_DEVICE_REGIONS ;;
srtrace) # where a short-read step's time goes: the lanes' timeline
       MM2AMD_TRACE=$O/r06_sr_trace_$V.tsv timeout 600 python bench.py --preset sr --reads 1000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/r06_bench_srt_$V.json 2> $O/r06_bench_srt_$V.log
       python tools/trace_summary.py $O/r06_sr_trace_$V.tsv 0.5 > $O/r06_sr_trace_summary_$V.txt 2>&1; head -60 $O/r06_sr_trace_summary_$V.txt
       python tools/trace_ascii.py $O/r06_sr_trace_$V.tsv --win 0.6 --res 4 2>/dev/null | head -50 > $O/r06_sr_trace_ascii_$V.txt; rm -f $O/r06_sr_trace_$V.tsv ;;
srsubs) # short-read pairs: shares per batch (reads per sub-batch)
       for n in default 100000 62500 41667 31250; do
         if [ $n = default ]; then unset MM2AMD_SUBBATCH_READS; else export MM2AMD_SUBBATCH_READS=$n; fi
         MM2AMD_BENCH_TRACE=1 timeout 600 python bench.py --preset sr --reads 1000000 --steps 6 --warmup 2 --no-cpu-baseline --timed-only > $O/r06_bench_sr_subs${n}_$V.json 2> $O/r06_bench_sr_subs${n}_$V.log
         python -c "
import json; d=json.loads(open('$O/r06_bench_sr_subs${n}_$V.json').read().strip().split('\\n')[-1]); print('sr sub-batch reads $n:', d['value'], d['ms_per_step'], d['config']['host_cpu_s_per_step'])"
         grep -h "stage  batch  4\|map    batch  3\|output batch  3\|free   batch  3" $O/r06_bench_sr_subs${n}_$V.log | tail -4 | cut -c1-100
       done; unset MM2AMD_SUBBATCH_READS ;;
e2esr) # short-read pairs end to end: the reference's fragment reader and writer around the GPU path (dropin_pipeline_gpu), SAM digest vs the minimap2 binary
       timeout 2400 python tools/e2e_wall.py --preset sr --reads ${E2E_PAIRS:-4000000} --dir /tmp/e2esr --out $O/r06_e2e_wall_sr_$V.json > $O/r06_e2e_wall_sr_$V.log 2>&1; tail -c 1800 $O/r06_e2e_wall_sr_$V.log ;;
esac
done
