#!/bin/bash
# one rocprof kernel trace of a short bench run -> step breakdown + timeline under gpurun_out/<tag>/
set -e -o pipefail
O=gpurun_out/${1:-qt}
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_EXTRA > $O/bench_under_rocprof.json 2> $O/prof.err
python3 tools/step_breakdown.py $(ls $O/prof/*/*kernel_trace.csv) > $O/step_breakdown.txt
python3 tools/step_timeline.py $(ls $O/prof/*/*kernel_trace.csv) > $O/step_timeline.txt
rm -rf $O/prof
head -4 $O/step_breakdown.txt
