python -m pytest tests -m gpu -q -x -k "fixture or fp32_config or bf16_mode" 2>&1 | tail -2
for rep in 1 2 3; do
for knob in "conv_dw_par=0" "conv_dw_par=1"; do
  python bench.py --no-cpu-baseline --no-profile --no-extras --precision bf16 --batch 32 --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$knob bf16 32x128', d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-profile --no-extras --batch 16 --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$knob f32 16x128', d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-profile --no-extras --batch 32 --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$knob f32 32x128', d['ms_per_step'])"
done; done
