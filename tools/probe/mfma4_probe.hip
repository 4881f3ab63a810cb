// Layout probe for v_mfma_f32_4x4x1_16B_f32: prints which (block, row, col) each lane / register of D holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[r * 64 + l] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256];
    // A lane l = 1000 * l marker is too coarse: use A[l] = 1 + l, B[l] = 100 + l and decode products
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    int bad = 0;
    for (int la = 0; la < 64; ++la) {
        for (int i = 0; i < 64; ++i) { ha[i] = (i == la) ? 1.f : 0.f; hb[i] = (float)(i + 1); }
        hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(da, db, dd);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        // expectation: A lane la = (block la/4, row la%4); D[block][row][col j] = B lane (block*4 + j) -> in VGPR row, lane block*4 + j
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                float want = (r == la % 4 && l / 4 == la / 4) ? (float)(l + 1) : 0.f;
                if (hd[r * 64 + l] != want) { if (bad < 10) printf("la %d r %d l %d got %g want %g\n", la, r, l, hd[r * 64 + l], want); ++bad; }
            }
    }
    printf("mfma_f32_4x4x1 layout as assumed: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    return bad != 0;
}
