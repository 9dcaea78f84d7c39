// Issue cost of the gfx950 16-bit MFMA shapes, one wave alone on a SIMD: N independent accumulators, back to back.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ void rate(long long* out, float* sink, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(0.5f + i);
    f32x4 c4[NACC];
    f32x16 c16[NACC];
    for (int n = 0; n < NACC; ++n) {
        for (int i = 0; i < 4; ++i) c4[n][i] = 0.f;
        for (int i = 0; i < 16; ++i) c16[n][i] = 0.f;
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if constexpr (SHAPE == 16) c4[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[n], 0, 0, 0);
                else c16[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16[n], 0, 0, 0);
            }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += SHAPE == 16 ? c4[n][0] : c16[n][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
void run(const char* name, int threads) {
    long long* d;
    float* sink;
    hipMalloc(&d, 8);
    hipMalloc(&sink, 4 * 1024);
    const int iters = 20000;
    hipLaunchKernelGGL((rate<SHAPE, NACC>), dim3(1), dim3(threads), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d wave(s)/SIMD, %d accumulators: %.1f cycles per MFMA per wave\n", name, threads / 256 ? threads / 256 : 1, NACC,
           (double)h / ((double)iters * 4 * NACC));
}

int main() {
    run<16, 1>("v_mfma_f32_16x16x32_f16 dependent chain", 64);
    run<16, 4>("v_mfma_f32_16x16x32_f16", 64);
    run<16, 4>("v_mfma_f32_16x16x32_f16", 512);
    run<32, 1>("v_mfma_f32_32x32x16_f16 dependent chain", 64);
    run<32, 4>("v_mfma_f32_32x32x16_f16", 64);
    run<32, 4>("v_mfma_f32_32x32x16_f16", 512);
    return 0;
}
