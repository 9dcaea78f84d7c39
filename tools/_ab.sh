for rep in 1 2; do
for knob in "pack_one=0" "pack_one=1"; do
  python bench.py --no-cpu-baseline --no-profile --no-extras --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 64x128 $knob', d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-profile --no-extras --precision bf16 --batch 32 --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 32x128 $knob', d['ms_per_step'])"
done; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "fixture or fp32_config or g6_config4" 2>&1 | tail -2
