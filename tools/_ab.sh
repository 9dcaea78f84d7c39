for knob in "deterministic=0" "deterministic=1"; do
  python bench.py --no-cpu-baseline --no-profile --no-extras --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 64x128 $knob', d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-profile --no-extras --model G6 --batch 32 --frames 192 --tune $knob 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 G6 32x192 $knob', d['ms_per_step'])"
done
python -m pytest tests -m gpu -q -x -k "deterministic or trained" 2>&1 | tail -2
