// Why does a banded head tile of dec_seq_kernel take ~430 cycles per K quad (4 dependent products = 256)?
// One wave per SIMD runs tiles of 16 quads: per quad [LDS reads of the next quad's B values] [scalar guard] [4 products
// on ONE accumulator].  Variants: guard on/off, LDS reads on/off (B from registers), code replicated REP times
// (instruction footprint REP x ~2 KB per tile).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool GUARD, bool LDSB, int REP>
__global__ void __launch_bounds__(256, 1) probe(float* out, long long* cyc, int iters, int nqb, const float* w) {
    __shared__ float col[4][64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* c = col[wave] + lane;
    for (int s = 0; s < 64; ++s) c[s * 64] = 0.001f * (float)(s + lane);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float4 a[16];
    for (int q = 0; q < 16; ++q) a[q] = ((const float4*)w)[q * 64 + lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < REP; ++rep) {
            float bv[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[0][j] = LDSB ? c[j * 64] : a[0].x;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q + 1 < 16) {
                    const int qn = (q + 1 < nqb) ? q + 1 : nqb - 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[(q + 1) & 1][j] = LDSB ? c[(4 * qn + j) * 64] : a[(q + 1) & 15].y;
                }
                if (!GUARD || q < nqb) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bv[q & 1][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bv[q & 1][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bv[q & 1][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bv[q & 1][3], acc, 0, 0, 0);
                }
            }
            asm volatile("" : "+v"(acc));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool GUARD, bool LDSB, int REP>
static void run(const char* what, int nqb) {
    float *out, *w; long long* cyc;
    const int blocks = 256, iters = 400 / REP;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&w, 16 * 64 * 16);
    hipMemset(w, 0, 16 * 64 * 16);
    hipMalloc(&cyc, sizeof(long long) * blocks * 4);
    for (int rep = 0; rep < 2; ++rep) probe<GUARD, LDSB, REP><<<blocks, 256>>>(out, cyc, iters, nqb, w);
    long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s nqb %2d rep %2d : %7.1f cycles / quad executed (256 = matrix time)\n", what, nqb, REP,
           (double)h[0] / ((double)iters * REP * nqb));
    hipFree(out); hipFree(w); hipFree(cyc);
}

int main() {
    run<false, false, 1>("no guard, B in registers", 16);
    run<true, false, 1>("guard, B in registers", 16);
    run<false, true, 1>("no guard, B from LDS", 16);
    run<true, true, 1>("guard, B from LDS", 16);
    run<true, true, 1>("guard, B from LDS", 11);
    run<true, true, 8>("guard, B from LDS", 16);
    run<true, true, 32>("guard, B from LDS", 16);
    run<true, true, 32>("guard, B from LDS", 11);
    run<false, true, 32>("no guard, B from LDS", 16);
    run<false, false, 32>("no guard, B in registers", 16);
    return 0;
}
