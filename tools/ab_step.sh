#!/bin/bash
# A/B of ss_tune settings on the training step, one process per setting, same box: bash tools/ab_step.sh <outdir> "k=v k=v" "k=v" ...
# prints ms/step per setting (bench.py --no-cpu-baseline --no-extras)
O=gpurun_out/$1
shift
mkdir -p $O
i=0
for cfg in "$@"; do
  args=""
  for kv in $cfg; do [ "$kv" != "-" ] && args="$args --tune $kv"; done
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $args > $O/ab_$i.json 2> $O/ab_$i.err || { echo "FAILED: $cfg"; tail -3 $O/ab_$i.err; }
  python - "$cfg" $O/ab_$i.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    kc = d.get('kernel_classes', {})
    print(f"{sys.argv[1]:40s} {d['ms_per_step']:.3f} ms  " + ' '.join(f"{k}={v['us_per_step']:.0f}" for k, v in kc.items()), flush=True)
except Exception as e:
    print(sys.argv[1], 'no result', e)
P
  i=$((i+1))
done
