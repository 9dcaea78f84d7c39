#!/bin/bash
# Modelled data-parallel schedule on one GPU (no multi-GPU node needed): bash tools/dp_timeline.sh <outdir under gpurun_out> [ranks]
# -> <outdir>/dp_timeline_buckets.txt (per-layer buckets) and dp_timeline_round2.txt (two buckets behind the whole backward)
set -e -o pipefail
O=gpurun_out/${1:-dp}
R=${2:-8}
mkdir -p $O
export TMPDIR=/tmp
for plan in 1 0; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof$plan -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-profile --force-dp --tune dp_model=$R --tune dp_buckets=$plan > $O/bench_dp$plan.json 2> $O/prof$plan.err
  name=$([ $plan = 1 ] && echo buckets || echo round2)
  python3 tools/dp_timeline.py $(ls $O/prof$plan/*/*kernel_trace.csv) > $O/dp_timeline_$name.txt
  python3 tools/step_timeline.py $(ls $O/prof$plan/*/*kernel_trace.csv) > $O/dp_step_timeline_$name.txt
  rm -rf $O/prof$plan
  cat $O/dp_timeline_$name.txt
done
