#!/bin/bash
# one-GPU step times of the configurations DESIGN.md quotes, one process each:  gpurun -- 'bash tools/config_sweep.sh [all]'
run() { timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --no-profile --steps 30 --warmup 8 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '->', d['ms_per_step'], 'ms', d['value'], d['unit'])"; }
run --workload config5 --frames 192
run --workload config5 --frames 192 --precision bf16
if [ "$1" = all ]; then
run
run --frames 192
run --model G6 --batch 32 --frames 192
run --model G6 --batch 32 --frames 192 --precision bf16
run --batch 32 --precision bf16
run --batch 32 --precision bf16 --tune bf16_img=0
run --batch 64 --precision bf16
run --batch 64 --precision bf16 --tune bf16_img=0
run --model G6 --batch 32 --frames 192 --precision bf16 --tune bf16_img=0
run --batch 16
run --batch 32
run --force-dp
run --force-dp --dp-backend torch
run --tune deterministic=1
fi
