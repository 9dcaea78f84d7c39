// Do VALU instructions of one wave overlap with MFMAs of another wave on the same SIMD?  (gfx950)
// Block = 512 threads = 8 waves = 2 per SIMD.  mode 0: all waves MFMA; 1: all waves VALU; 2: even waves MFMA, odd waves VALU
// (waves w and w+4 share a SIMD: wave id % 4 = SIMD), 3: waves 0-3 MFMA + waves 4-7 VALU (one of each per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    bool do_mfma = mode == 0 || (mode == 2 && (w & 1) == 0) || (mode == 3 && w < 4);
    bool do_valu = mode == 1 || (mode == 2 && (w & 1) == 1) || (mode == 3 && w >= 4);
    float r = 0.f;
    if (do_mfma) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (do_valu) {
        float x0 = threadIdx.x * 1e-3f, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {          // 128 independent-ish VALU ops per iteration (vs 16 MFMAs = 512 cycles)
                x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0001f + 0.5f; x2 = x2 * 1.0001f + 0.5f; x3 = x3 * 1.0001f + 0.5f;
                x4 = x4 * 1.0001f + 0.5f; x5 = x5 * 1.0001f + 0.5f; x6 = x6 * 1.0001f + 0.5f; x7 = x7 * 1.0001f + 0.5f;
            }
        }
        r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[4] = {"8 waves MFMA", "8 waves VALU", "MFMA+VALU waves on the same SIMDs (w%2)", "waves 0-3 MFMA, 4-7 VALU (one each per SIMD)"};
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 100, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d  %-48s %8.3f ms\n", mode, names[mode], ms);
    }
    printf("per iteration and wave: 16 MFMA 32x32x16 (= 512 matrix-pipe cycles) and/or 128 VALU fma (= 512 issue cycles)\n");
    return 0;
}
