for rep in 1 2; do
for lib in head new; do
  if [ $lib = new ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=$PWD/speechsplit_amd/lib/ab/libss_$lib.so; fi
  python bench.py --no-cpu-baseline --no-profile --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib f32 64x128', d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-profile --no-extras --precision bf16 --batch 32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib bf16 32x128', d['ms_per_step'])"
done; done
for lib in head new; do
  if [ $lib = new ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=$PWD/speechsplit_amd/lib/ab/libss_$lib.so; fi
  echo == $lib; python tools/lstm_nl_error.py 512 192 16 3.0 2>&1 | grep -v amdgpu | head -5
done
unset SS_LIB_PATH
python -m pytest tests -m gpu -q -x -k "blstm or fixture or fp32_config or trained" 2>&1 | tail -2
