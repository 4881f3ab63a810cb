// Does vector-ALU work hide under fp32 MFMAs on gfx950?  One wave per SIMD (or two) runs a loop of 16 independent-chain
// v_mfma_f32_32x32x2_f32 with NV vector instructions (v_fma_f32 / v_exp_f32 / v_pk_fma_f32) interleaved one by one behind
// the products, and reports shader cycles per loop iteration.  Matrix work alone is 16 x 64 = 1024 cycles.
//   hipcc -O3 --offload-arch=gfx950 -o coexec_probe coexec_probe.hip && ./coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int KIND, bool MFMA, int NACC = 4>
__global__ void __launch_bounds__(512) probe(float* out, long long* cyc, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(i + r);
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[16];
    f32x2 pv[8];
    for (int i = 0; i < 16; ++i) v[i] = seed + (float)i;
    for (int i = 0; i < 8; ++i) pv[i] = (f32x2){seed + (float)i, seed};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MFMA) acc[m & (NACC - 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & (NACC - 1)], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV / 16; ++k) {
                const int j = (m * (NV / 16) + k) & 15;
                if (KIND == 0) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
                if (KIND == 1) v[j] = __builtin_amdgcn_exp2f(v[j]);
                if (KIND == 2) pv[j & 7] = __builtin_elementwise_fma(pv[j & 7], (f32x2){1.0001f, 1.0001f}, (f32x2){0.5f, 0.5f});
            }
            if (MFMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NV >= 16) __builtin_amdgcn_sched_group_barrier(0x002, NV / 16, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += pv[i].x + pv[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int KIND, bool MFMA, int NACC = 4>
static void run(const char* what, int threads) {
    float* out; long long* cyc;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    for (int rep = 0; rep < 2; ++rep) probe<NV, KIND, MFMA, NACC><<<blocks, threads>>>(out, cyc, iters, 1.0f);
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD %d  NV %3d : %7.1f cycles / iteration\n", what, threads / 256, NV, (double)h[0] / iters);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0, true>("16 MFMA alone", 256);
    run<0, 0, true, 1>("16 MFMA, ONE accumulator chain", 256);
    run<0, 0, true, 2>("16 MFMA, two accumulator chains", 256);
    run<16, 0, true>("16 MFMA + v_fma_f32", 256);
    run<64, 0, true>("16 MFMA + v_fma_f32", 256);
    run<128, 0, true>("16 MFMA + v_fma_f32", 256);
    run<192, 0, true>("16 MFMA + v_fma_f32", 256);
    run<64, 0, false>("v_fma_f32 alone", 256);
    run<128, 0, false>("v_fma_f32 alone", 256);
    run<64, 1, true>("16 MFMA + v_exp_f32", 256);
    run<128, 1, true>("16 MFMA + v_exp_f32", 256);
    run<64, 1, false>("v_exp_f32 alone", 256);
    run<64, 2, true>("16 MFMA + v_pk_fma_f32", 256);
    run<128, 2, true>("16 MFMA + v_pk_fma_f32", 256);
    run<64, 2, false>("v_pk_fma_f32 alone", 256);
    run<0, 0, true>("16 MFMA alone", 512);
    run<64, 0, true>("16 MFMA + v_fma_f32", 512);
    run<128, 0, true>("16 MFMA + v_fma_f32", 512);
    run<64, 1, true>("16 MFMA + v_exp_f32", 512);
    return 0;
}
