#!/bin/bash
# Regenerates everything kept under profiles/<round>/ on a GPU box: gpurun -- 'bash tools/run_profiles.sh r01'
# (GPU tests first; later steps only run if the earlier ones succeeded)
set -e -o pipefail
R=${1:-r03}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
# the ablation benches need the -DSS_DIAG library; build it HERE before the gpurun call (make -C speechsplit_amd/csrc diag), not on the box
test -f speechsplit_amd/lib/libspeechsplit_hip_diag.so || { echo 'libspeechsplit_hip_diag.so is missing: make -C speechsplit_amd/csrc diag'; exit 1; }
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.txt 2>&1 || { tail -30 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
# counter passes first: bench.py reads the HBM traffic of its roofline kernel from profiles/<round>/gemm_pmc.json
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/gemm_pmc.py 1024 > $O/pmc_f.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/gemm_pmc.py 1024 > $O/pmc_w.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- python3 tools/gemm_pmc.py 1024 > $O/pmc_s.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/gemm_pmc
mkdir -p profiles/$R && cp $O/gemm_pmc.json $O/gemm_pmc.txt profiles/$R/
timeout -k 10 400 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err
cat $O/bench_n1.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.err
cp $(ls $O/prof/*/*kernel_stats.csv) $O/bench_kernel_stats.csv
python3 tools/step_breakdown.py $(ls $O/prof/*/*kernel_trace.csv) > $O/step_breakdown.txt
python3 tools/step_timeline.py $(ls $O/prof/*/*kernel_trace.csv) > $O/step_timeline.txt
head -12 $O/step_breakdown.txt
SS_DIAG_LIB=1 timeout -k 10 120 python3 tools/kbench.py seq 2>&1 | grep -v amdgpu.ids > $O/kbench_seq_ablation.txt || true
SS_DIAG_LIB=1 timeout -k 10 120 python3 tools/kbench.py seqtag 2>&1 | grep -v amdgpu.ids > $O/kbench_seq_handoff.txt || true
SS_DIAG_LIB=1 timeout -k 10 120 python3 tools/kbench.py gws gabl 2>&1 | grep -v amdgpu.ids > $O/kbench_gemm_forms.txt || true
timeout -k 10 200 python3 tools/img_bench.py 2>&1 | grep -v amdgpu.ids > $O/img_gemm.txt || true
SS_DIAG_LIB=1 timeout -k 10 200 python3 tools/img_bench.py diag 2>&1 | grep -v amdgpu.ids > $O/img_gemm_ablation.txt || true
bash tools/dp_timeline.sh $R/dp 8 > $O/dp_timeline.log 2>&1 || true
cp $O/dp/dp_timeline_buckets.txt $O/dp/dp_timeline_round2.txt $O/dp/dp_step_timeline_buckets.txt profiles/$R/ 2>/dev/null || true
timeout -k 10 200 python3 tools/seq_stride_probe.py layouts 2>&1 | grep -v amdgpu.ids > $O/seq_operand_temperature.txt || true
timeout -k 10 200 python3 tools/f16x2_error.py 2>&1 | grep -v amdgpu.ids > $O/f16x2_error.txt || true
cp $O/img_gemm.txt $O/img_gemm_ablation.txt $O/bench_kernel_stats.csv $O/step_breakdown.txt $O/kbench_seq_ablation.txt $O/kbench_seq_handoff.txt $O/kbench_gemm_forms.txt $O/seq_operand_temperature.txt $O/f16x2_error.txt $O/bench_n1.json profiles/$R/ 2>/dev/null || true
cp $O/step_timeline.txt profiles/$R/step_timeline.txt
rm -rf $O/prof/*/*.db
