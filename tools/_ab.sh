python -m pytest tests -m gpu -q -x -k "fused_groupnorm or conv_block or fixture or fp32_config" 2>&1 | tail -2
for rep in 1 2 3; do
for lib in head new; do
  if [ $lib = new ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=$PWD/speechsplit_amd/lib/ab/libss_$lib.so; fi
  python bench.py --no-cpu-baseline --no-profile --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib f32 64x128', d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-profile --no-extras --precision bf16 --batch 32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib bf16 32x128', d['ms_per_step'])"
done; done
