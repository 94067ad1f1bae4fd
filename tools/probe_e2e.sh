#!/bin/bash
# usage: tools/probe_e2e.sh KIND PRESET REF_MB N_READS  -- end-to-end timing probe on the GPU box (reference CPU vs drop-in GPU)
KIND=$1; PRESET=$2; MB=$3; N=$4
D=/tmp/probe_$KIND
python tests/synth.py $KIND $D --ref-mb $MB --reads $N --seed 21 > /dev/null
NT=$(nproc)
echo "== index build (reference, -t $NT)"
T0=$(date +%s.%N); oracle/_ref/minimap2_ref -x $PRESET -t $NT -d $D/ref.mmi $D/ref.fa 2>&1 | tail -2; T1=$(date +%s.%N); echo "idx wall $(echo "$T1 - $T0" | bc) s"
echo "== reference mapping -t $NT"
T0=$(date +%s.%N); oracle/_ref/minimap2_ref -ax $PRESET -t $NT $D/ref.mmi $D/reads.fa 2> $D/ref.err | grep -v '^@PG' > $D/ref.sam; T1=$(date +%s.%N); echo "ref wall $(echo "$T1 - $T0" | bc) s"
grep -E "mm_idx_stat::|worker_pipeline|Real time" $D/ref.err | tail -6
echo "== drop-in GPU"
T0=$(date +%s.%N); tests/_build/dropin_gpu -x $PRESET -a -t $NT --stats $D/ref.mmi $D/reads.fa 2> $D/gpu.err | grep -v '^@PG' > $D/gpu.sam; T1=$(date +%s.%N); echo "gpu wall $(echo "$T1 - $T0" | bc) s"
grep -E "dropin" $D/gpu.err | tail -5
cmp $D/ref.sam $D/gpu.sam && echo "SAM IDENTICAL ($(wc -l < $D/ref.sam) lines)"
