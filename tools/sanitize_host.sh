#!/bin/bash
# Host pipeline under the sanitizers (dev container only: needs /root/reference headers and oracle/_ref).
# Builds tests/cpucheck (host sources + oracle-backed check backend + the reference-I/O driver) twice, with
# -fsanitize=address,undefined and with -fsanitize=thread, into scratch directories, and runs the driver over synthetic
# inputs of every mode (long reads, splice + junction / jump annotation, short reads single / paired / unpaired, splice:sr, SDUST,
# all-vs-all, staged calls, library formatter).  Prints one line per run; any sanitizer report makes the script fail.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/mm2amd_sanitize}
mkdir -p "$W/data"
python3 - "$ROOT" "$W/data" <<'PY'
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
import synth
d = sys.argv[2]
synth.make("ont", d + "/ont", 2, 40, 23)
synth.make_pairs(d + "/pe")
synth.make_short(d + "/se")
synth.make_overlaps(d + "/ovl")
synth.make_junctions(d + "/jn")
synth.make_rna_pairs(d + "/rna")
synth.make_splice_scores(d + "/jn/ref.fa", d + "/jn/spsc.tsv")
synth.make_weird(d + "/weird")
PY
fail=0
printf "leak:mm_idx_spsc_read2\nleak:ks_getuntil2\n" > "$W/lsan.supp" # the reference's own reader keeps the score table (index.c:1017)
for san in address,undefined thread; do
	O="$W/${san%%,*}"
	mkdir -p "$O"
	make -s -C "$ROOT/tests/cpucheck" OUT="$O" CXXFLAGS="-std=c++17 -O1 -g -fPIC -Wall -ffp-contract=off -fsanitize=$san -fno-omit-frame-pointer -I$ROOT/include" "$O/libmm2amd_check.so" || exit 1
	gcc -O1 -g -DHAVE_KALLOC -I/root/reference -I"$ROOT/include" -c "$ROOT/tests/dropin/dropin_main.c" -o "$O/dropin_main.o" || exit 1
	g++ -fsanitize=$san -o "$O/dropin_check" "$O/dropin_main.o" "$ROOT/oracle/_ref/libminimap2_ref.a" -L"$O" -lmm2amd_check -L"$ROOT/oracle" -loracle -lm -lz -lpthread || exit 1
	export LD_LIBRARY_PATH="$ROOT/oracle:$O"
	D="$W/data"
	while read -r args; do
		envs=""
		while [[ "$args" == *=*\ * && "${args%% *}" == *=* ]]; do envs="$envs ${args%% *}"; args="${args#* }"; done # leading VAR=value words
		env $envs ASAN_OPTIONS=detect_leaks=1 LSAN_OPTIONS=suppressions=$W/lsan.supp:print_suppressions=0 "$O/dropin_check" $args > "$O/out.txt" 2> "$O/err.txt"
		rc=$?
		n=$(grep -c "ERROR: \|runtime error\|WARNING: ThreadSanitizer" "$O/err.txt")
		echo "$san rc=$rc reports=$n :: $args"
		if [ $rc -ne 0 ] || [ "$n" -ne 0 ]; then fail=1; head -30 "$O/err.txt"; fi
	done <<EOF2
-x map-ont -a -t 8 --format-lib $D/ont/ref.fa $D/ont/reads.fa
-x map-ont -c -t 8 --staged $D/ont/ref.fa $D/ont/reads.fa
-x sr -a -t 8 $D/pe/ref.fa $D/pe/r1.fa $D/pe/r2.fa
-x sr -c -t 8 --format-lib $D/pe/ref.fa $D/pe/inter.fa
-x sr -a -t 8 --no-pairing $D/pe/ref.fa $D/pe/r1.fa $D/pe/r2.fa
-x sr -a -t 8 $D/se/ref.fa $D/se/reads.fa
-x ava-ont -c -t 8 $D/ovl/ovl.fa $D/ovl/ovl.fa
-x splice -a -t 8 --junc-bed $D/jn/junc.bed $D/jn/ref.fa $D/jn/reads.fa
-x splice -c -t 8 -j $D/jn/junc.bed $D/jn/ref.fa $D/jn/reads.fa
-x splice -a -t 8 --spsc $D/jn/spsc.tsv $D/jn/ref.fa $D/jn/reads.fa
-x splice:sr -a -t 8 -j $D/rna/introns.bed $D/rna/ref.fa $D/rna/r1.fa $D/rna/r2.fa
-x map-ont -a -t 8 -T 10 $D/weird/ref.fa $D/weird/reads.fa
-x map-ont -c -t 8 --qstrand --cs $D/ont/ref.fa $D/ont/reads.fa
MM2AMD_GPUS=3 -x map-ont -a -t 9 -K 200k $D/ont/ref.fa $D/ont/reads.fa
MM2AMD_GPUS=2 -x sr -a -t 8 --staged --format-lib $D/pe/ref.fa $D/pe/r1.fa $D/pe/r2.fa
-x asm20 -c -t 8 $D/ont/ref.fa $D/ont/reads.fa
-x lr:hqae -a -t 8 $D/weird/ref.fa $D/weird/reads.fa
-x map-ont -a -t 4 --one-by-one $D/ont/ref.fa $D/ont/reads.fa
EOF2
	# the three-step pipeline driver: mm_gpu_format_batch of batch k beside mm_gpu_map_batch of batch k+1
	gcc -O1 -g -DHAVE_KALLOC -I/root/reference -I"$ROOT/include" -c "$ROOT/tests/dropin/dropin_pipeline.c" -o "$O/dropin_pipeline.o" || exit 1
	g++ -fsanitize=$san -o "$O/dropin_pipeline_check" "$O/dropin_pipeline.o" "$ROOT/oracle/_ref/libminimap2_ref.a" -L"$O" -lmm2amd_check -L"$ROOT/oracle" -loracle -lm -lz -lpthread || exit 1
	ASAN_OPTIONS=detect_leaks=1 LSAN_OPTIONS=suppressions=$W/lsan.supp:print_suppressions=0 "$O/dropin_pipeline_check" -x map-ont -a -t 8 -K 100k $D/ont/ref.fa $D/ont/reads.fa > "$O/out.txt" 2> "$O/err.txt"
	rc=$?
	n=$(grep -c "ERROR: \|runtime error\|WARNING: ThreadSanitizer" "$O/err.txt")
	echo "$san rc=$rc reports=$n :: dropin_pipeline -x map-ont -a -K 100k"
	if [ $rc -ne 0 ] || [ "$n" -ne 0 ]; then fail=1; head -30 "$O/err.txt"; fi
done
exit $fail
