R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
MM2AMD_TRACE=/tmp/trace.tsv timeout 200 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/tr.json 2> $O/tr.log
python $R/tools/trace_gantt.py /tmp/trace.tsv 2 > $O/trace_gantt.txt; head -3 $O/trace_gantt.txt; tail -70 $O/trace_gantt.txt
