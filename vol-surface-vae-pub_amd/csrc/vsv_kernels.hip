// gfx950 kernels of libvsv_hip.so and their launchers.
//
// Execution model: one wave owns a tile of 32 independent rows
// (batch index m = lane & 31, lane half hh = lane >> 5).  Every layer is
// y^T[N x 32] = W[N x K] . x^T[K x 32] on v_mfma_f32_32x32x2_f32 (exact f32 FMA
// chain), with the WEIGHTS as the A operand (pre-packed fragments, streamed from
// L2, prefetched one k-quad ahead) and the ACTIVATIONS as the B operand.  The D
// layout of a layer coincides with the B layout of the next one (vsv_layout.h), so
// activations never move between lanes: a lane keeps its slice of every
// activation vector either in registers (decoder head chain, static indexing) or
// in a private LDS column ("col[s*64 + lane]", one slot per k-step).
//
// Schedule of the recurrent kernel: a workgroup of 4 waves owns ONE 32-row tile and
// splits every layer's output tiles between its waves (gate tiles w, w+4, ... of the
// LSTM layers; one tile each of the decoder-head layers); h and the head activations
// are exchanged through LDS with workgroup barriers.  Per wave that is 4 accumulator
// tiles (64 registers), so two workgroups share a CU (2 waves per SIMD): one wave's
// cell / epilogue VALU work overlaps the other's MFMAs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <type_traits>
#include <vector>

#include "vsv_layout.h"

namespace vsv {

// Diagnostic build only (-DVSV_STAMPS, tools/stamps.sh): per-wave s_memtime stamps of the fused decode
// kernel's phases, written to a buffer nothing else reads.
#ifdef VSV_STAMPS
constexpr int kStampWgs = 16384;
__device__ unsigned long long g_stamps[(size_t)kStampWgs * 4 * 32];
#define VSV_STAMP(i)                                                                                          \
    do {                                                                                                      \
        if (XMODE == 1 && HEAD && lane == 0 && blockIdx.x < kStampWgs)                                        \
            g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime();               \
    } while (0)
// (mc_decode1_kernel, tools/stamps_mc1.py: 0 start, 1 latents, 2s+2 / 2s+3 GEMM / cell of pass s, 18+2li / 19+2li
// matrix part / epilogue of head layer li)
#define MC1_STAMP(i)                                                                                          \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (lane == 0 && blockIdx.x < kStampWgs)                                                              \
            g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
// (dec_seq_kernel, tools/stamps_dseq.py: the phases of time step 1 -- 0 start, 1 + 2s / 2 + 2s GEMM / cell of pass s,
// 17 columns written, 18 + 2li / 19 + 2li matrix part / epilogue of head layer li, 26 end of step)
#define DSQ_STAMP(i)                                                                                          \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (dsq_t == 1 && lane == 0 && blockIdx.x < kStampWgs)                                                \
            g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
// (lstm_rec4_kernel / lstm_rec16_kernel, tools/stamps_rec.py: two consecutive time steps from the middle of the
// sequence -- 8s + 0 top of step, + 1 products issued, + 2 cell and stores issued, + 3 past the barrier; lstm_rec16_kernel:
// + 1 first half issued, + 4 past its barrier, + 5 second half issued, + 2 early tiles published, + 3 past the barrier)
#ifndef VSV_STAMP_NT
#define VSV_STAMP_NT 16      /* gate tiles of the launches that write recurrent-step stamps (-DVSV_STAMP_NT=13: the encoders') */
#endif
#define REC_STAMP(i)                                                                                          \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (NT == VSV_STAMP_NT && (t == (a.T >> 1) || t == (a.T >> 1) + 1) && lane == 0 && wave < 8 && blockIdx.x < kStampWgs / 2) \
            g_stamps[((size_t)blockIdx.x * 8 + wave) * 32 + 8 * (t - (a.T >> 1)) + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
// (lstm_bwd4_kernel<16>, tools/stamps_bwd.py: one step from the middle of the sequence -- 0 top, 1 products issued,
// 2 partial tiles published, 3 past the first barrier, 4 cell backward and stores issued, 5 past the second barrier)
#define BWD_STAMP(i)                                                                                          \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (NT == 16 && t == (a.T >> 1) && lane == 0 && wave < 4 && blockIdx.x < kStampWgs)                    \
            g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#else
#define VSV_STAMP(i) do {} while (0)
#define MC1_STAMP(i) do {} while (0)
#define DSQ_STAMP(i) do {} while (0)
#define REC_STAMP(i) do {} while (0)
#define BWD_STAMP(i) do {} while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// feature index held by lane half hh in accumulator register reg of tile t (NAT32 out order)
__device__ __forceinline__ int feat_of(int t, int reg, int hh) { return 32 * t + 8 * (reg >> 2) + 4 * hh + (reg & 3); }

// acc[i] += W[tile toff[i]] . B over nq k-quads.  w: this lane's float4 at quad 0, tile 0 of
// the segment; toff[i]: tile offset in float4 units (tile * 64); qstride: float4 per quad
// (layer tiles * 64); b: this lane's activation column at the segment's first k-step
// (LDS or global, 64 floats per k-step).  The weights of quad q+1 are in flight while
// quad q is multiplied (two named buffers, loop unrolled by two).
template <int NC>
__device__ __forceinline__ void quad_mfma(f32x16 (&acc)[NC], const float4 (&a)[NC], const float* b, int q) {
    const float b0 = b[(4 * q + 0) * 64], b1 = b[(4 * q + 1) * 64], b2 = b[(4 * q + 2) * 64], b3 = b[(4 * q + 3) * 64];
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].x, b0, acc[i]);
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].y, b1, acc[i]);
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].z, b2, acc[i]);
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].w, b3, acc[i]);
}

// NC = 1: a quad is only 4 MFMAs (256 cycles = 0.1 us), so the weights run VSV_SEG1_DEPTH quads ahead
// (an L2 hit is 0.5-0.8 us away: four quads ahead left the one-tile-per-wave kernels waiting on it).
#ifndef VSV_SEG1_DEPTH
#define VSV_SEG1_DEPTH 8
#endif
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[1], const float4* __restrict__ w, unsigned lane,
                                         const int (&toff)[1], int qstride, const float* b, int nq) {
    constexpr int D = VSV_SEG1_DEPTH;
    // w is wave-uniform: indexing it with 32-bit offsets keeps the address in SGPRs + one VGPR
    const unsigned o = lane + (unsigned)toff[0];
    float4 a[D];
#pragma unroll
    for (int u = 0; u < D; ++u) a[u] = w[o + (unsigned)(((u < nq) ? u : nq - 1) * qstride)];
    int q = 0;
#pragma unroll 1
    for (; q + D <= nq; q += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const float4 cur[1] = {a[u]};
            const int qn = (q + D + u < nq) ? q + D + u : nq - 1;
            a[u] = w[o + (unsigned)(qn * qstride)];
            quad_mfma<1>(acc, cur, b, q + u);
        }
    }
#pragma unroll
    for (int u = 0; u < D - 1; ++u) {
        if (q + u < nq) {
            const float4 cur[1] = {a[u]};
            quad_mfma<1>(acc, cur, b, q + u);
        }
    }
}

template <int NC>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NC], const float4* __restrict__ w, unsigned lane,
                                         const int (&toff)[NC], int qstride, const float* b, int nq) {
    float4 a0[NC], a1[NC];
    unsigned o[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        o[i] = lane + (unsigned)toff[i];
        a0[i] = w[o[i]];
    }
    int q = 0;
#pragma unroll 1
    for (; q + 1 < nq; q += 2) {
        const unsigned q1 = (unsigned)((q + 1) * qstride);
#pragma unroll
        for (int i = 0; i < NC; ++i) a1[i] = w[o[i] + q1];
        quad_mfma<NC>(acc, a0, b, q);
        const unsigned q2 = (unsigned)(((q + 2 < nq) ? q + 2 : q + 1) * qstride);
#pragma unroll
        for (int i = 0; i < NC; ++i) a0[i] = w[o[i] + q2];
        quad_mfma<NC>(acc, a1, b, q + 1);
    }
    if (q < nq) quad_mfma<NC>(acc, a0, b, q);
}

// gemm_seg with the A fragments requested four quads ahead: for weights that are not cache-warm (the
// transposed blob of a training step is gathered anew every step) one quad of MFMAs (0.4 us) does not
// cover a fetch, four do.
template <int NC>
__device__ __forceinline__ void gemm_seg_deep(f32x16 (&acc)[NC], const float4* __restrict__ w, unsigned lane,
                                              const int (&toff)[NC], int qstride, const float* b, int nq) {
    float4 a[4][NC];
    unsigned o[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) o[i] = lane + (unsigned)toff[i];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NC; ++i) a[u][i] = w[o[i] + (unsigned)(((u < nq) ? u : nq - 1) * qstride)];
    int q = 0;
#pragma unroll 1
    for (; q + 4 <= nq; q += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float4 cur[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) cur[i] = a[u][i];
            const int qn = (q + 4 + u < nq) ? q + 4 + u : nq - 1;
#pragma unroll
            for (int i = 0; i < NC; ++i) a[u][i] = w[o[i] + (unsigned)(qn * qstride)];
            quad_mfma<NC>(acc, cur, b, q + u);
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        if (q + u < nq) quad_mfma<NC>(acc, a[u], b, q + u);
    }
}

// Pinned software pipeline for the 4-tiles-per-wave schedule: the A fragments (global, L2-resident) and the
// B values (bl(q, bv): 4 k-steps of this lane's activation column) of quad q+1 are issued BEFORE the 16
// MFMAs of quad q and the scheduler may not move anything across the fences, so a load has a full quad
// (>= 1024 cycles of matrix work) to land.  a0 holds quad 0 on entry (wload by the caller, as early as
// it likes); on exit a0 is free again.
template <int NC>
__device__ __forceinline__ void wload(float4 (&a)[NC], const float4* __restrict__ w, const unsigned (&o)[NC], unsigned qoff) {
#pragma unroll
    for (int i = 0; i < NC; ++i) a[i] = w[o[i] + qoff];
}
template <int NC>
__device__ __forceinline__ void mfma_quad(f32x16 (&acc)[NC], const float4 (&a)[NC], const float (&b)[4]) {
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].x, b[0], acc[i]);
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].y, b[1], acc[i]);
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].z, b[2], acc[i]);
#pragma unroll
    for (int i = 0; i < NC; ++i) acc[i] = mfma32(a[i].w, b[3], acc[i]);
}
template <int NC, typename BL>
__device__ __forceinline__ void gemm_stream(f32x16 (&acc)[NC], float4 (&a0)[NC], const float4* __restrict__ w,
                                            const unsigned (&o)[NC], int qstride, int nq, BL bl) {
    float4 a1[NC];
    float b0[4], b1[4];
    bl(0, b0);
    int q = 0;
#pragma unroll 1
    for (; q + 1 < nq; q += 2) {
        wload<NC>(a1, w, o, (unsigned)((q + 1) * qstride));
        bl(q + 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad<NC>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int q2 = (q + 2 < nq) ? q + 2 : q + 1;
        wload<NC>(a0, w, o, (unsigned)(q2 * qstride));
        bl(q2, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad<NC>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (q < nq) mfma_quad<NC>(acc, a0, b0);
}

// One output tile per wave (decoder head): the weights of the first kHalfQ quads are fetched by
// the caller before it waits at the previous layer's barrier; while they are multiplied each
// register is refilled with the quad kHalfQ further on, so the rest is in flight behind the MFMAs.
constexpr int kHalfQ = 8;
template <int NQ>
__device__ __forceinline__ void tile_load(float4 (&a)[kHalfQ], const float4* __restrict__ w, unsigned off, int qstride) {
#pragma unroll
    for (int q = 0; q < kHalfQ; ++q) a[q] = w[off + (unsigned)(((q < NQ) ? q : NQ - 1) * qstride)];
}
template <int NQ>
__device__ __forceinline__ void tile_mfma(f32x16& acc, float4 (&a)[kHalfQ], const float4* __restrict__ w, unsigned off,
                                          int qstride, const float* b) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 cur = a[q % kHalfQ];
        if (q + kHalfQ < NQ) a[q % kHalfQ] = w[off + (unsigned)((q + kHalfQ) * qstride)];
        acc = mfma32(cur.x, b[(4 * q + 0) * 64], acc);
        acc = mfma32(cur.y, b[(4 * q + 1) * 64], acc);
        acc = mfma32(cur.z, b[(4 * q + 2) * 64], acc);
        acc = mfma32(cur.w, b[(4 * q + 3) * 64], acc);
    }
}

// The same for a tile whose non-zero weights are the K quads [q0, q0 + nqb) (a conv layer in pixel-major
// order): only those are fetched and multiplied.  nqb is wave-uniform, so the guards are scalar branches.
__device__ __forceinline__ int pick4(const int (&v)[4], int w) { return w == 0 ? v[0] : (w == 1 ? v[1] : (w == 2 ? v[2] : v[3])); }
__device__ __forceinline__ void tile_load_band(float4 (&a)[kHalfQ], const float4* __restrict__ w, unsigned off, int qstride,
                                               int q0, int nqb) {
#pragma unroll
    for (int q = 0; q < kHalfQ; ++q) a[q] = w[off + (unsigned)((q0 + ((q < nqb) ? q : nqb - 1)) * qstride)];
}
template <int NQ>
__device__ __forceinline__ void tile_mfma_band(f32x16& acc, float4 (&a)[kHalfQ], const float4* __restrict__ w, unsigned off,
                                               int qstride, const float* b, int q0, int nqb) {
    const float4* wb = w + (size_t)q0 * qstride;
    const float* bb = b + (size_t)q0 * 4 * 64;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 cur = a[q % kHalfQ];
        // (the refill is unconditional and clamped to the band's last quad: under the scalar branch it was loaded
        // into a spare tuple, waited for with vmcnt(0) right after the quad's products and copied over)
        if (q + kHalfQ < NQ) {
            const int qn = (q + kHalfQ < nqb) ? q + kHalfQ : nqb - 1;
            a[q % kHalfQ] = wb[off + (unsigned)(qn * qstride)];
        }
        if (q < nqb) {
            acc = mfma32(cur.x, bb[(4 * q + 0) * 64], acc);
            acc = mfma32(cur.y, bb[(4 * q + 1) * 64], acc);
            acc = mfma32(cur.z, bb[(4 * q + 2) * 64], acc);
            acc = mfma32(cur.w, bb[(4 * q + 3) * 64], acc);
        }
    }
}

// The same streams with BUFFER-addressed weight loads (mc_decode1_kernel): the fragment address is
// descriptor base (SGPRs) + lane offset (one VGPR per phase) + a scalar byte offset that carries tile, quad and
// band, so a load costs no vector instruction (the flat form above spends one 32-bit add and one 64-bit
// shift-add per load: nine vector instructions per quad of sixteen products).  The descriptor has no stride and
// no bound: offsets are plain byte offsets from the section's base, which the packer keeps below 4 GiB.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
__device__ __forceinline__ wrsrc_t make_wrsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ float4 bload(wrsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    // (the whole vector is reinterpreted: __builtin_bit_cast of ONE ELEMENT of a vector reads element 0 whatever
    // the index with this compiler)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v g = __builtin_bit_cast(f32x4v, v);
    float4 f;
    f.x = g.x; f.y = g.y; f.z = g.z; f.w = g.w;
    return f;
}
__device__ __forceinline__ float bload1(wrsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bstore1(wrsrc_t r, float v, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
// NC consecutive tiles (1 KiB of fragments each) of the quad at scalar byte offset soff
template <int NC>
__device__ __forceinline__ void wload_buf(float4 (&a)[NC], wrsrc_t r, unsigned voff, unsigned soff) {
#pragma unroll
    for (int i = 0; i < NC; ++i) a[i] = bload(r, voff, soff + (unsigned)i * 1024u);
}
template <int NC, typename BL>
__device__ __forceinline__ void gemm_stream_buf(f32x16 (&acc)[NC], float4 (&a0)[NC], wrsrc_t r, unsigned voff, unsigned sbase,
                                                unsigned qsb, int nq, BL bl) {
    float4 a1[NC];
    float b0[4], b1[4];
    bl(0, b0);
    int q = 0;
#pragma unroll 1
    for (; q + 1 < nq; q += 2) {
        wload_buf<NC>(a1, r, voff, sbase + (unsigned)(q + 1) * qsb);
        bl(q + 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad<NC>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int q2 = (q + 2 < nq) ? q + 2 : q + 1;
        wload_buf<NC>(a0, r, voff, sbase + (unsigned)q2 * qsb);
        bl(q2, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_quad<NC>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (q < nq) mfma_quad<NC>(acc, a0, b0);
}
// gemm_stream_buf for ONE wave per SIMD (dec_seq_kernel): nobody else fills the gap between two quads of products,
// where the fenced form issues the next quad's four fragment loads and LDS reads back to back (~100 of a quad's 1024
// cycles).  Here they go out one by one in the shadow of the quad's first products (memory and LDS instructions issue
// while a product executes; vector instructions do not).
template <int NC, typename BL>
__device__ __forceinline__ void gemm_stream_buf1(f32x16 (&acc)[NC], float4 (&a0)[NC], wrsrc_t r, unsigned voff, unsigned sbase,
                                                 unsigned qsb, int nq, BL bl) {
    static_assert(NC == 4, "interleave pattern written for four tiles");
    float4 a1[NC];
    float b0[4], b1[4];
    bl(0, b0);
    auto pattern = [] {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one product
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one fragment load
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    int q = 0;
#pragma unroll 1
    for (; q + 1 < nq; q += 2) {
        wload_buf<NC>(a1, r, voff, sbase + (unsigned)(q + 1) * qsb);
        bl(q + 1, b1);
        mfma_quad<NC>(acc, a0, b0);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
        const int q2 = (q + 2 < nq) ? q + 2 : q + 1;
        wload_buf<NC>(a0, r, voff, sbase + (unsigned)q2 * qsb);
        bl(q2, b0);
        mfma_quad<NC>(acc, a1, b1);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (q < nq) mfma_quad<NC>(acc, a0, b0);
}
// The same with the fragments and B values requested TWO quads ahead (three buffers, loop unrolled by three): one quad
// of products (1024 cycles, 0.43 us) does not cover an L2 round trip under load.  a0 holds quad 0 on entry.
template <int NC, typename BL>
__device__ __forceinline__ void gemm_stream_buf2(f32x16 (&acc)[NC], float4 (&a0)[NC], wrsrc_t r, unsigned voff, unsigned sbase,
                                                 unsigned qsb, int nq, BL bl) {
    static_assert(NC >= 2 && NC <= 5, "interleave pattern written for 2 - 5 tiles");
    float4 a1[NC], a2[NC];
    float b0[4], b1[4], b2[4];
    auto pattern = [] {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one product
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one fragment load
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NC - 4, 0);
    };
    const int last = nq - 1;
    bl(0, b0);
    {
        const int q1 = last < 1 ? last : 1;
        wload_buf<NC>(a1, r, voff, sbase + (unsigned)q1 * qsb);
        bl(q1, b1);
    }
    __builtin_amdgcn_sched_barrier(0);
    int q = 0;
#pragma unroll 1
    for (; q + 2 < nq; q += 3) {
        wload_buf<NC>(a2, r, voff, sbase + (unsigned)(q + 2) * qsb);
        bl(q + 2, b2);
        mfma_quad<NC>(acc, a0, b0);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
        const int q3 = (q + 3 < last) ? q + 3 : last;
        wload_buf<NC>(a0, r, voff, sbase + (unsigned)q3 * qsb);
        bl(q3, b0);
        mfma_quad<NC>(acc, a1, b1);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
        const int q4 = (q + 4 < last) ? q + 4 : last;
        wload_buf<NC>(a1, r, voff, sbase + (unsigned)q4 * qsb);
        bl(q4, b1);
        mfma_quad<NC>(acc, a2, b2);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (q < nq) mfma_quad<NC>(acc, a0, b0);
    if (q + 1 < nq) mfma_quad<NC>(acc, a1, b1);
}
__device__ __forceinline__ void tile_load_band_buf(float4 (&a)[kHalfQ], wrsrc_t r, unsigned voff, unsigned sbase, unsigned qsb,
                                                   int q0, int nqb) {
#pragma unroll
    for (int q = 0; q < kHalfQ; ++q) a[q] = bload(r, voff, sbase + (unsigned)(q0 + ((q < nqb) ? q : nqb - 1)) * qsb);
}
template <int NQ>
__device__ __forceinline__ void tile_mfma_band_buf(f32x16& acc, float4 (&a)[kHalfQ], wrsrc_t r, unsigned voff, unsigned sbase,
                                                   unsigned qsb, const float* b, int q0, int nqb) {
    const unsigned sb = sbase + (unsigned)q0 * qsb;
    const float* bb = b + (size_t)q0 * 4 * 64;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q < nqb) {
            const float4 cur = a[q % kHalfQ];
            if (q + kHalfQ < nqb) a[q % kHalfQ] = bload(r, voff, sb + (unsigned)(q + kHalfQ) * qsb);
            acc = mfma32(cur.x, bb[(4 * q + 0) * 64], acc);
            acc = mfma32(cur.y, bb[(4 * q + 1) * 64], acc);
            acc = mfma32(cur.z, bb[(4 * q + 2) * 64], acc);
            acc = mfma32(cur.w, bb[(4 * q + 3) * 64], acc);
        }
    }
}

// A banded tile with ALL of its K quads requested up front (mc_decode1_kernel; a band has at most NQ = 16 quads):
// the refill scheme above keeps eight quads in flight and re-requests a register when its quad has been multiplied,
// but a refill under a scalar branch makes the compiler load into a spare tuple, wait for it with vmcnt(0) after
// the quad's four products and copy it over -- every refill exposed most of an L2 round trip.  Here the loads are
// unconditional (quads past the band re-request its last quad: the compiler can count them, so the products wait
// quad by quad with vmcnt(15), vmcnt(14), ...) and only the products are guarded.
template <int NQ>
__device__ __forceinline__ void tile_band_all_buf(f32x16& acc, wrsrc_t r, unsigned voff, unsigned sbase, unsigned qsb,
                                                  const float* b, int q0, int nqb) {
    const unsigned sb = sbase + (unsigned)q0 * qsb;
    const float* bb = b + (size_t)q0 * 4 * 64;
    float4 a[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = bload(r, voff, sb + (unsigned)(q < nqb ? q : nqb - 1) * qsb);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q < nqb) {
            acc = mfma32(a[q].x, bb[(4 * q + 0) * 64], acc);
            acc = mfma32(a[q].y, bb[(4 * q + 1) * 64], acc);
            acc = mfma32(a[q].z, bb[(4 * q + 2) * 64], acc);
            acc = mfma32(a[q].w, bb[(4 * q + 3) * 64], acc);
        }
    }
}

// The two halves of tile_band_all_buf, for callers that request a tile's fragments while the tile before it is
// multiplied (dec_seq_kernel: one wave per SIMD, nobody else covers the fetch)
template <int NQ>
__device__ __forceinline__ void band_load_all(float4 (&a)[NQ], wrsrc_t r, unsigned voff, unsigned sbase, unsigned qsb, int q0,
                                              int nqb) {
    const unsigned sb = sbase + (unsigned)q0 * qsb;
#pragma unroll
    for (int q = 0; q < NQ; ++q) a[q] = bload(r, voff, sb + (unsigned)(q < nqb ? q : nqb - 1) * qsb);
}
// (the B values of quad q + 1 are read from LDS -- unconditionally, clamped to the band -- in front of quad q's
// products: read inside the band's scalar guards they were waited for quad by quad)
template <int NQ>
__device__ __forceinline__ void band_mfma_all(f32x16& acc, const float4 (&a)[NQ], const float* b, int q0, int nqb) {
    const float* bb = b + (size_t)q0 * 4 * 64;
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[0][j] = bb[j * 64];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (q + 1 < NQ) {
            const int qn = (q + 1 < nqb) ? q + 1 : nqb - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[(q + 1) & 1][j] = bb[(4 * qn + j) * 64];
        }
        if (q < nqb) {
            acc = mfma32(a[q].x, bv[q & 1][0], acc);
            acc = mfma32(a[q].y, bv[q & 1][1], acc);
            acc = mfma32(a[q].z, bv[q & 1][2], acc);
            acc = mfma32(a[q].w, bv[q & 1][3], acc);
        }
    }
}

// A value parked in the accumulator half of the register file (gfx950: one 512-entry file per SIMD, split into
// vector and accumulator registers).  dec_seq_kernel keeps the cell state of both memory layers there for the
// whole time loop and takes sixteen values at a time into vector registers for a pass's cell math: left to the
// register allocator the state occupied half of the vector registers and everything else spilled around it.
__device__ __forceinline__ float agpr_put(float v) {
    float a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
__device__ __forceinline__ float agpr_get(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// The bias of output tile `tile` (compact [nt][2][16] layout) as an A fragment: lane l holds the bias of tile row
// l & 31 (both lane halves, i.e. both k-steps, the same value).  One product with a B operand that is 1 in the lower
// lanes (k-step 0) and 0 in the upper turns it into the tile's initial accumulator, exactly (0 + b * 1 + b * 0): one
// register per lane and tile to fetch ahead instead of sixteen, and nothing to do to the value once it has landed
// (dec_seq_kernel).
__device__ __forceinline__ float bias_frag(const float* __restrict__ b, int tile, int lane) {
    const int i = lane & 31;
    return b[tile * 32 + ((i >> 2) & 1) * 16 + 4 * (i >> 3) + (i & 3)];
}
__device__ __forceinline__ f32x16 bias_acc(float bf, float one_lo) {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return mfma32(bf, one_lo, z);
}

// acc[i] = bias of tile tile[i] (compact [nt][2][16] layout)
template <int NC>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NC], const float* __restrict__ b, const int (&toff)[NC], int hh) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const float4* p = (const float4*)(b + (toff[i] >> 6) * 32 + hh * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 x = p[v];
            acc[i][4 * v + 0] = x.x; acc[i][4 * v + 1] = x.y; acc[i][4 * v + 2] = x.z; acc[i][4 * v + 3] = x.w;
        }
    }
}

// A copy of x the optimiser cannot see through.  Lane-dependent addresses derived from it are computed
// in the phase that uses them instead of being hoisted above the time loop and parked in scratch.
template <bool ON = true>
__device__ __forceinline__ int opaque(int x) {
    if (ON) asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// max(x, 0) as ONE instruction (fmaxf first canonicalises an operand it cannot prove quiet: a second v_max)
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// tanh(x) = 2 sigmoid(2x) - 1: five instructions (exp(-2x) overflows to +inf -> -1, underflows to 0 -> +1);
// absolute error ~1e-7 around 0, where the quotient form it replaces (eight instructions) was relatively exact
__device__ __forceinline__ float fast_tanh(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.885390081777927f)) - 1.0f;
}

// ---------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: one N(0,1) value per (sample, day, step, f).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t& c0, uint32_t& c1, uint32_t& c2,
                                              uint32_t& c3) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// nn.LSTM's inter-layer dropout (train mode): the factor (0 or 1/(1-p)) applied to the four hidden
// units 8*tile + 2*jj + hh (jj = 0..3) of `row` at step t of layer `layer` of LSTM `id` (0 posterior
// encoder, 1 context encoder, 2 decoder).  One Philox block per (row, t, tile, hh); unit u keeps its
// value iff the top 24 bits of word jj, scaled to [0,1), are >= p.  Forward and backward call this
// with the same arguments, so the mask is never stored.
struct DropArgs {
    float p, scale;             // p = 0: no dropout
    uint32_t k0, k1;            // seed
    uint32_t id;
    const uint32_t* key_dev;    // non-null: the seed's two words are read from device memory at kernel start
};
__device__ __forceinline__ DropArgs resolve_drop(DropArgs d) {
    if (d.key_dev) { d.k0 = d.key_dev[0]; d.k1 = d.key_dev[1]; }       // uniform address: scalar loads
    return d;
}
__device__ __forceinline__ void dropout_factors(const DropArgs& d, uint32_t row, uint32_t t, int tile, int hh, int layer,
                                                float (&f)[4]) {
    uint32_t c0 = row, c1 = t, c2 = (uint32_t)(2 * tile + hh), c3 = 0x44520000u | (d.id << 8) | (uint32_t)layer;
    philox4x32_10(d.k0, d.k1, c0, c1, c2, c3);
    const uint32_t x[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = ((float)(x[j] >> 8) * 5.9604644775390625e-08f >= d.p) ? d.scale : 0.0f;
}

__device__ __forceinline__ float philox_normal_elem(uint32_t k0, uint32_t k1, uint32_t sample,
                                                    uint32_t day, uint32_t step, int f) {
    uint32_t c0 = sample, c1 = day, c2 = step, c3 = (uint32_t)(f >> 2);
    philox4x32_10(k0, k1, c0, c1, c2, c3);
    const int p = (f & 3) >> 1;
    const uint32_t xa = p ? c2 : c0, xb = p ? c3 : c1;
    const float u1 = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-07f;   // 2^-23
    const float u2 = ((float)(xb >> 9) + 0.5f) * 1.1920928955078125e-07f;
    // hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in turns, so
    // u2 goes in unscaled); absolute error ~1e-6 on an N(0,1) draw, far below the KS test's resolution
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    return r * ((f & 1) ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
}

// ---------------------------------------------------------------------------
// dense_chain: a stack of dense layers (conv embeddings, linear heads, decoder
// head) applied to independent (row tile, t) instances.  One wave per instance;
// activations live in ONE private LDS column per wave, updated in place (a layer's
// outputs stay in the accumulators until all of its inputs have been read).
// ---------------------------------------------------------------------------
struct DenseLayerArg {
    const float* w;
    const float* b;
    int nq, nt, n_relu;
    int q0[4], nqb[4];      // output tile t reads K quads [q0[t], q0[t] + nqb[t]) (Sec::qlo / qhi); fused head only
};

struct DenseChainArgs {
    int n_layers;
    DenseLayerArg layer[kMaxDenseLayers];
    long n_rows;
    long n_rowtiles;
    int T;
    // input
    int in_mode;            // 0 row-major gather (natural order), 1 tile-B
    const float* in;
    long in_rs, in_ts;      // row-major strides (floats)
    int in_f;               // row-major: real features
    int in_ks;              // k-steps to fill (mode 0: multiple of 4; mode 1: k-steps per instance in `in`)
    // injection into layer inj_layer's output features [inj_f0, inj_f0+inj_n) (the ex-features joining
    // the conv embedding, torch.cat at cvae_with_mem_randomized.py:188; interaction Linears may follow)
    const float* inj;
    long inj_rs, inj_ts;
    int inj_f0, inj_n, inj_layer;
    // output of the last layer
    int out_mode;           // 0 tile-B, 1 row-major (row, t, f), 2 sample-minor (row/S, t, f, row%S)
    float* out;
    int out_f;
    long S;
    // aux output: features [aux_f0, aux_f0+aux_n) of layer aux_layer -> aux (row, t, aux_n)
    int aux_layer;
    float* aux;
    int aux_f0, aux_n;
    // training: the INPUT of every layer (64 slots) -> save[(li*n_inst + inst)*64 + s][64]
    float* save;
};

constexpr int kDenseSlots = 16 * kHeadTiles;     // LDS k-step slots per wave

// layer[li] of dense_chain_kernel's by-value argument, read from the kernarg segment itself (scalar
// loads at a run-time offset): indexing the by-value copy with a run-time li makes the compiler keep
// the whole struct in scratch (1 KiB per lane).  The struct is the kernel's only explicit argument,
// so it sits at offset 0 of the segment.
__device__ __forceinline__ DenseLayerArg chain_layer_of_kernarg(int li, unsigned koff = 0) {
    // (koff: byte offset of the DenseChainArgs inside the kernarg segment -- 0 for the kernels whose only argument
    // it is, sizeof(DenseChainArgs) for the second set of dense_chain_wg_pair_kernel)
    typedef const __attribute__((address_space(4))) DenseChainArgs* KP;
    typedef const __attribute__((address_space(4))) char* KC;
    const KP kp = (KP)((KC)__builtin_amdgcn_kernarg_segment_ptr() + koff);
    DenseLayerArg r;
    r.w = kp->layer[li].w;
    r.b = kp->layer[li].b;
    r.nq = kp->layer[li].nq;
    r.nt = kp->layer[li].nt;
    r.n_relu = kp->layer[li].n_relu;
    return r;
}

// Per-(row, t) pointers of a chain's side effects, hoisted out of the per-register epilogue:
// out + f*fstride is the final store (row-major or sample-minor), inj/aux are pre-offset by
// their first feature so they are indexed by the feature number itself.  inj and aux features
// always live in the LAST tile of their layer (make_layout checks c_last*25 >= 96 when X > 0).
struct RowIo {
    float* out;
    long fstride;
    const float* inj;
    float* aux;
};

__device__ __forceinline__ RowIo make_rowio(const DenseChainArgs& a, long row, bool row_ok, int t) {
    RowIo io = {nullptr, 1, nullptr, nullptr};
    if (a.out_mode == 1 && row_ok) {
        io.out = a.out + ((size_t)row * a.T + t) * a.out_f;
    } else if (a.out_mode == 2 && row_ok) {
        const long d = row / a.S;
        io.out = a.out + ((size_t)d * a.T + t) * a.out_f * a.S + (row - d * a.S);
        io.fstride = a.S;
    }
    if (a.inj) io.inj = a.inj + row * a.inj_rs + (long)t * a.inj_ts - a.inj_f0;
    if (a.aux && row_ok) io.aux = a.aux + ((size_t)row * a.T + t) * a.aux_n - a.aux_f0;
    return io;
}

// The row's output pointers, computed where they are used (the epilogue of the layer that stores) from
// the caller's opaque lane index: hoisted above the time loop, sixteen 64-bit store addresses per lane
// were kept alive -- in scratch -- through every GEMM.
__device__ __forceinline__ RowIo late_rowio(const DenseChainArgs& hd, long rowtile, int lane, int to) {
    const int mv = lane & 31;
    const long rr = rowtile * 32 + mv;
    const bool ok = rr < hd.n_rows;
    return make_rowio(hd, ok ? rr : hd.n_rows - 1, ok, to);
}

// ReLU mask, injection and aux output for the NT tiles of layer li (in place)
template <int NT>
__device__ __forceinline__ void dense_epilogue(const DenseChainArgs& a, const RowIo& io, int li, bool last,
                                               f32x16 (&acc)[NT], int hh, int n_relu) {
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = (feat_of(tt, r, hh) < n_relu) ? fmaxf(acc[tt][r], 0.0f) : acc[tt][r];
    const bool do_inj = (li == a.inj_layer) && io.inj != nullptr, do_aux = (li == a.aux_layer) && io.aux != nullptr;
    if (do_inj) {
        float iv[16];                                              // (clamped addresses: the loads go out together)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = feat_of(NT - 1, r, hh);
            iv[r] = io.inj[f < a.inj_f0 ? a.inj_f0 : (f >= a.inj_f0 + a.inj_n ? a.inj_f0 + a.inj_n - 1 : f)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = feat_of(NT - 1, r, hh);
            if (f >= a.inj_f0 && f < a.inj_f0 + a.inj_n) acc[NT - 1][r] = iv[r];
        }
    }
    if (do_aux) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = feat_of(NT - 1, r, hh);
            if (f >= a.aux_f0 && f < a.aux_f0 + a.aux_n) io.aux[f] = acc[NT - 1][r];
        }
    }
}

template <int NT>
__device__ __forceinline__ void dense_store(const DenseChainArgs& a, const RowIo& io, const f32x16 (&acc)[NT], int hh) {
    if (!io.out) return;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = feat_of(tt, r, hh);
            if (f < a.out_f) io.out[(size_t)f * io.fstride] = acc[tt][r];
        }
}

template <int NT>
__device__ __forceinline__ void dense_layer(const DenseChainArgs& a, const DenseLayerArg& ly, long rowtile, int t, int li,
                                            float* col0, int lane0, long inst) {
    // every lane-dependent value of the layer (feature masks, LDS and global addresses) is derived from an
    // opaque copy of the lane index, so that it is computed here and not hoisted above the layer loop,
    // where the masks and addresses of all layers together overflowed the register file into scratch
    const int lane = opaque(lane0);
    float* const col = col0 + lane;
    const int hh = lane >> 5;
    f32x16 acc[NT];
    int toff[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) toff[i] = i * 64;
    load_bias<NT>(acc, ly.b, toff, hh);
    gemm_seg<NT>(acc, (const float4*)ly.w, lane, toff, NT * 64, col, ly.nq);
    const bool last = (li == a.n_layers - 1);
    // row pointers only where a layer touches global memory, from an opaque lane (see late_rowio)
    RowIo io = {nullptr, 1, nullptr, nullptr};
    if ((last && a.out_mode != 0) || (li == a.inj_layer && a.inj != nullptr) || li == a.aux_layer)
        io = late_rowio(a, rowtile, lane, t);
    dense_epilogue<NT>(a, io, li, last, acc, hh, ly.n_relu);
    if (!last) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) col[(16 * tt + r) * 64] = acc[tt][r];
        // zero the slots a narrower layer leaves behind so the next layer's padded k-steps read 0
        for (int s = 16 * NT; s < kDenseSlots; ++s) col[s * 64] = 0.0f;
    } else if (a.out_mode == 0) {
        float* o = a.out + ((size_t)inst * (16 * NT)) * 64 + lane;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(16 * tt + r) * 64] = acc[tt][r];
    } else {
        dense_store<NT>(a, io, acc, hh);
    }
}

__global__ void __launch_bounds__(256, 2) dense_chain_kernel(const DenseChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long inst = (long)blockIdx.x * 4 + wave;
    if (inst >= a.n_rowtiles * a.T) return;
    const long rowtile = inst / a.T;
    const int t = (int)(inst % a.T);
    const int hh = lane >> 5, m = lane & 31;
    long row = rowtile * 32 + m;
    const bool row_ok = row < a.n_rows;
    if (!row_ok) row = a.n_rows - 1;
    float* col = smem + (size_t)wave * (kDenseSlots * 64) + lane;

    // input column, sixteen k-steps at a time: the group's loads are issued together from clamped addresses and the
    // values selected afterwards (one conditional load per k-step meant one exposed L2 round trip per k-step)
    if (a.in_mode == 0) {
        const float* src = a.in + row * a.in_rs + (long)t * a.in_ts;
#pragma unroll 1
        for (int s0 = 0; s0 < kDenseSlots; s0 += 16) {
            if (s0 < a.in_ks) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int f = 2 * (s0 + k) + hh;
                    v[k] = src[f < a.in_f ? f : a.in_f - 1];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    col[(s0 + k) * 64] = (s0 + k < a.in_ks && 2 * (s0 + k) + hh < a.in_f) ? v[k] : 0.0f;
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) col[(s0 + k) * 64] = 0.0f;
            }
        }
    } else {
        const float* src = a.in + (size_t)inst * a.in_ks * 64 + lane;
#pragma unroll 1
        for (int s0 = 0; s0 < kDenseSlots; s0 += 16) {
            if (s0 < a.in_ks) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = src[(s0 + k < a.in_ks ? s0 + k : a.in_ks - 1) * 64];
#pragma unroll
                for (int k = 0; k < 16; ++k) col[(s0 + k) * 64] = (s0 + k < a.in_ks) ? v[k] : 0.0f;
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) col[(s0 + k) * 64] = 0.0f;
            }
        }
    }
#pragma unroll 1
    for (int li = 0; li < a.n_layers; ++li) {
        if (a.save) {
            float* sv = a.save + ((size_t)li * (a.n_rowtiles * a.T) + inst) * kDenseSlots * 64 + lane;
#pragma unroll 16
            for (int s = 0; s < kDenseSlots; ++s) sv[s * 64] = col[s * 64];
        }
        const DenseLayerArg ly = chain_layer_of_kernarg(li);
        switch (ly.nt) {
            case 1: dense_layer<1>(a, ly, rowtile, t, li, col - lane, lane, inst); break;
            case 2: dense_layer<2>(a, ly, rowtile, t, li, col - lane, lane, inst); break;
            case 3: dense_layer<3>(a, ly, rowtile, t, li, col - lane, lane, inst); break;
            default: dense_layer<4>(a, ly, rowtile, t, li, col - lane, lane, inst); break;
        }
    }
}

// dense_chain_wg_kernel: the same chain with one WORKGROUP per instance, for few instances (the decoder head of a
// training batch of 64 at horizon 1 is two instances; the latent heads and small date ranges likewise): wave w
// computes output tile w of every layer from the column shared in LDS.  Per-tile product chains as in
// dense_chain_kernel: results are bit-equal.
__device__ __forceinline__ void dense_chain_wg_body(const DenseChainArgs& a, const unsigned koff) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long inst = (long)blockIdx.x;
    const long rowtile = inst / a.T;
    const int t = (int)(inst % a.T);
    const int hh = lane >> 5, m = lane & 31;
    long row = rowtile * 32 + m;
    const bool row_ok = row < a.n_rows;
    if (!row_ok) row = a.n_rows - 1;
    float* col = smem + lane;
    const int s_lo = 16 * wave;
    {
        // (loads from clamped addresses, values selected afterwards: see dense_chain_kernel)
        float v[16];
        if (a.in_mode == 0) {
            const float* src = a.in + row * a.in_rs + (long)t * a.in_ts;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int f = 2 * (s_lo + k) + hh;
                v[k] = src[f < a.in_f ? f : a.in_f - 1];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                col[(s_lo + k) * 64] = (s_lo + k < a.in_ks && 2 * (s_lo + k) + hh < a.in_f) ? v[k] : 0.0f;
        } else {
            const float* src = a.in + (size_t)inst * a.in_ks * 64 + lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[(s_lo + k < a.in_ks ? s_lo + k : a.in_ks - 1) * 64];
#pragma unroll
            for (int k = 0; k < 16; ++k) col[(s_lo + k) * 64] = (s_lo + k < a.in_ks) ? v[k] : 0.0f;
        }
    }
    // A layer's fragments (this wave's tile, all of its <= 16 K quads) and bias are requested one layer AHEAD, behind
    // the products of the layer before: with a handful of instances per launch (the decoder head of a training batch of
    // 64 is two) a layer is a chain of 64 dependent products (1.7 us) and used to begin with two exposed round trips
    // (bias, first fragments) of about that length.
    float4 wq[4 * kHeadTiles];
    f32x16 bn[1];
    auto prefetch = [&](const DenseLayerArg& ly) {
        const int tw = wave < ly.nt ? wave : ly.nt - 1;          // (waves without a tile fetch the last one's: no branch)
        const float4* w = (const float4*)ly.w + (unsigned)(tw * 64) + lane;
#pragma unroll
        for (int q = 0; q < 4 * kHeadTiles; ++q) wq[q] = w[(unsigned)((q < ly.nq ? q : ly.nq - 1) * ly.nt * 64)];
        const int toff1[1] = {tw * 64};
        load_bias<1>(bn, ly.b, toff1, hh);
    };
    DenseLayerArg ly = chain_layer_of_kernarg(0, koff);
    prefetch(ly);
    __syncthreads();
#pragma unroll 1
    for (int li = 0; li < a.n_layers; ++li) {
        if (a.save) {
            float* sv = a.save + ((size_t)li * (a.n_rowtiles * a.T) + inst) * kDenseSlots * 64 + lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) sv[(s_lo + k) * 64] = col[(s_lo + k) * 64];
        }
        const bool last = (li == a.n_layers - 1);
        const bool mine = wave < ly.nt;
        f32x16 acc[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = mine ? bn[0][r] : 0.0f;
        if (mine) {
#pragma unroll
            for (int q = 0; q < 4 * kHeadTiles; ++q) {
                if (q < ly.nq) {
                    const float4 cur[1] = {wq[q]};
                    quad_mfma<1>(acc, cur, col, q);
                }
            }
        }
        const DenseLayerArg lyn = chain_layer_of_kernarg(last ? li : li + 1, koff);
        if (!last) prefetch(lyn);
        if (mine) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[0][r] = (feat_of(wave, r, hh) < ly.n_relu) ? fmaxf(acc[0][r], 0.0f) : acc[0][r];
            // injection / aux features live in the layer's LAST tile
            RowIo io = {nullptr, 1, nullptr, nullptr};
            if ((last && a.out_mode != 0) || (li == a.inj_layer && a.inj != nullptr) || li == a.aux_layer)
                io = make_rowio(a, row, row_ok, t);
            const bool do_inj = (li == a.inj_layer) && io.inj != nullptr, do_aux = (li == a.aux_layer) && io.aux != nullptr;
            if (do_inj && wave == ly.nt - 1) {
                float iv[16];                                      // (clamped addresses: the loads go out together)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = feat_of(wave, r, hh);
                    iv[r] = io.inj[f < a.inj_f0 ? a.inj_f0 : (f >= a.inj_f0 + a.inj_n ? a.inj_f0 + a.inj_n - 1 : f)];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = feat_of(wave, r, hh);
                    if (f >= a.inj_f0 && f < a.inj_f0 + a.inj_n) acc[0][r] = iv[r];
                }
            }
            if (do_aux && wave == ly.nt - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = feat_of(wave, r, hh);
                    if (f >= a.aux_f0 && f < a.aux_f0 + a.aux_n) io.aux[f] = acc[0][r];
                }
            }
            if (last) {
                if (a.out_mode == 0) {
                    float* o = a.out + ((size_t)inst * (16 * ly.nt)) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[(s_lo + r) * 64] = acc[0][r];
                } else if (io.out) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = feat_of(wave, r, hh);
                        if (f < a.out_f) io.out[(size_t)f * io.fstride] = acc[0][r];
                    }
                }
            }
        }
        if (!last) {
            __syncthreads();                                       // every wave has read this layer's input
#pragma unroll
            for (int r = 0; r < 16; ++r) col[(s_lo + r) * 64] = acc[0][r];        // (tiles >= nt: zeros)
            __syncthreads();
        }
        ly = lyn;
    }
}
__global__ void __launch_bounds__(256) dense_chain_wg_kernel(const DenseChainArgs a) { dense_chain_wg_body(a, 0u); }
// two chains as one launch (blockIdx.y picks the argument set; see lstm_rec4_pair_kernel)
__global__ void __launch_bounds__(256) dense_chain_wg_pair_kernel(const DenseChainArgs a0, const DenseChainArgs a1) {
    if (blockIdx.y == 0) {
        if ((long)blockIdx.x < a0.n_rowtiles * a0.T) dense_chain_wg_body(a0, 0u);
    } else {
        if ((long)blockIdx.x < a1.n_rowtiles * a1.T) dense_chain_wg_body(a1, (unsigned)sizeof(DenseChainArgs));
    }
}

// ---------------------------------------------------------------------------
// lstm_kernel: a 1- or 2-layer LSTM walked over T steps for a tile of 32 rows
// per workgroup, with (XMODE 0) the x part read from a tile-B tensor written by
// dense_chain (encoders) or (XMODE 1) assembled from ctx / z / noise (decoder);
// optionally (HEAD) followed at every output step by the decoder head chain.
// ---------------------------------------------------------------------------
struct LstmSeqArgs {
    const float* w[2];
    const float* b[2];
    int kx;                 // k-steps of the x part (multiple of 4)
    long n_rows, n_rowtiles;
    int T;
    // XMODE 0
    const float* x_tileb;   // [(rowtile*T + t)*kx + s][64]
    // XMODE 1: decoder input [ctx (Lc features: L with compress_context, else the context memory's width) | z (L)]
    int L, Lc;
    vsv_decode_in din;
    float* z_out;           // (n_rows, T, L) or null
    // state
    float* state_out;       // (n_rows, NL, 2, 8*NT) or null
    // per-step top-layer h in tile-B layout for t >= t_out0: [(rowtile*(T-t_out0) + t-t_out0)*4*NT + s][64]
    float* hout;
    int t_out0;
    // fused decoder head (HEAD): layers, outputs and aux as in DenseChainArgs (in_* unused, T = T - t_out0)
    DenseChainArgs head;
    // training (SAVE): activated gates [inst][l][16*NT][64] (slot 16*tile + 4*jj + gate), cell and hidden
    // state after the step [inst][l][4*NT][64], decoder x part [inst][kx][64]; inst = rowtile*T + t
    float *sv_gates, *sv_c, *sv_h, *sv_x;
    // training with mem_dropout > 0 (SAVE, NL == 2): layer 1 reads the dropped copy of layer 0's h,
    // which is also saved ([inst][4*NT][64]) as the X of layer 1's weight gradient
    DropArgs drop;
    float* sv_hd;
    int gru;                // memory cell: 0 nn.LSTM, 1 nn.GRU, 2 nn.RNN (tanh)
    // training, one launch per layer (NL = 1): the saved arrays hold sv_nl layers per instance and this
    // launch is layer sv_l0 (w[0] / b[0] are that layer's); rec: the gate accumulators start from the
    // pre-activations b + W_ih . x that lstm_pre_kernel left in sv_gates for every step, and only the
    // recurrent product stays on the time loop
    int rec, sv_nl, sv_l0;
    // XMODE 0, inference over overlapping windows (row r, step t <-> calendar day r + t): layer 0's gate
    // pre-activations b + W_ih . x per CALENDAR DAY, transposed ([32*NT gate values][pre_stride days], gate
    // value j = 2*(16*tile + register) + lane half), computed once per day instead of once per
    // (window, step); layer 0 starts from them and skips its input product
    // Step t of row r reads column r + t * pre_tstride: 1 for calendar windows; for autoregressive
    // rollouts the table is [generated day][gate value][row] and pre_tstride = one day's table.
    const float* pre_t;
    long pre_stride, pre_tstride;
    // XMODE 0, inference, one launch per layer (NL = 1; the date-range prefix as a pipeline of layer stages):
    // the gate pre-activations b + W_ih . x of every step, [inst][16*NT][64] as lstm_pre_kernel writes them
    // (inst = rowtile*T + t); the launch keeps only the recurrent product on its time loop
    const float* pre_g;
    // the state arrays (state_in / state_out / gate_out) hold st_nl layers per row and this launch's layer 0
    // is layer st_l0 of them (0: NL layers, this launch's own)
    int st_nl, st_l0;
    // lstm_rec16_kernel<13>, inference: a Linear on the layer's h of every step rides as the spare gate tile of
    // wave 2 (the posterior encoder's z_mean in the date-range prefix).  hd_w / hd_b: one packed dense layer of one
    // output tile over the layer's 4*NT k-steps; hd_out: tile-B [inst][hd_slots][64] in natural feature order
    // (feature j: slot j >> 1, lane half j & 1), features >= hd_n written as zeros; hd_slots <= 8
    const float *hd_w, *hd_b;
    float* hd_out;
    int hd_n, hd_slots;
};

// per-thread constants of the decoder input assembly
struct DecIn {
    const float* ctx;       // this row's ctx block or null
    long z_idx;             // this row's offset into zb / lv
    const float* eps;       // this row's eps block or null
    uint32_t sample, day;
};

__device__ __forceinline__ float decoder_input(const LstmSeqArgs& a, const DecIn& di, long row, bool row_ok, bool writer,
                                               int t, int f) {
    const vsv_decode_in& d = a.din;
    const int L = a.L, Lc = a.Lc;
    if (f < Lc) return (t < d.ctx_T) ? di.ctx[(long)t * Lc + f] : 0.0f;
    f -= Lc;
    if (f >= L) return 0.0f;
    float base, scale;
    if (t < d.z_T) {
        const long idx = di.z_idx + (long)t * L + f;
        base = d.zb[idx];
        scale = d.lv ? expf(0.5f * d.lv[idx]) : 0.0f;
    } else {
        base = d.mu0;
        scale = d.std0;
    }
    float v = base;
    if (scale != 0.0f) {
        const int te = t - d.eps_t0;
        float eps;
        if (d.eps_in)
            eps = di.eps[(size_t)te * L + f];
        else
            // (opaque: otherwise the first Philox round's products of the sample index are hoisted out of
            // the time loop and live in scratch)
            eps = philox_normal_elem((uint32_t)d.seed, (uint32_t)(d.seed >> 32), (uint32_t)opaque((int)di.sample), di.day,
                                     (uint32_t)(d.step0 + te), f);
        v = base + scale * eps;
    }
    if (a.z_out && row_ok && writer) a.z_out[((size_t)row * a.T + t) * L + f] = v;
    return v;
}

// OWN = gate tiles per wave: 4 (4 waves per row tile: tiles w, w+4, w+8, w+12; many row tiles) or
// 1 (one wave per gate tile, NT waves per row tile: few row tiles, shortest time per step)

// LSTM cell: lane holds gates (i,f,g,o) of unit 8t + 2jj + hh in acc[i][4jj .. 4jj+3]
template <bool KEEP, int OWN>
__device__ __forceinline__ void lstm_cell(f32x16 (&acc)[OWN], float (&c)[OWN][4], float (&hn)[OWN][4], int kind) {
    if (kind == 2) {
        // nn.RNN: slot 0 holds W_ih x + b_ih + W_hh h + b_hh, h' = tanh of it (slots 1-3 are zero rows).  The `c`
        // registers carry h like the GRU's; KEEP leaves h' in slot 0 for the backward pass.
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float h = fast_tanh(acc[i][4 * jj + 0]);
                c[i][jj] = h;
                hn[i][jj] = h;
                if (KEEP) acc[i][4 * jj + 0] = h;
            }
        }
        return;
    }
    if (kind == 1) {
        // nn.GRU: slots (r, z, n_x, n_h); n = tanh(n_x + r * n_h), h' = (1 - z) * n + z * h.  The `c`
        // registers carry this lane's units of h.  KEEP leaves (r, z, n, n_h) for the backward pass.
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float r = fast_sigmoid(acc[i][4 * jj + 0]);
                const float z = fast_sigmoid(acc[i][4 * jj + 1]);
                // (explicit fused multiply-adds: left to -ffp-contract the template instances of a kernel may
                // fuse these differently, and launches that should agree bit for bit differ in the last place)
                const float n = fast_tanh(__builtin_fmaf(r, acc[i][4 * jj + 3], acc[i][4 * jj + 2]));
                const float h = __builtin_fmaf(z, c[i][jj] - n, n);
                c[i][jj] = h;
                hn[i][jj] = h;
                if (KEEP) {
                    acc[i][4 * jj + 0] = r; acc[i][4 * jj + 1] = z; acc[i][4 * jj + 2] = n;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float ig = fast_sigmoid(acc[i][4 * jj + 0]);
            const float fg = fast_sigmoid(acc[i][4 * jj + 1]);
            const float gg = fast_tanh(acc[i][4 * jj + 2]);
            const float og = fast_sigmoid(acc[i][4 * jj + 3]);
            const float cn = __builtin_fmaf(fg, c[i][jj], ig * gg);
            c[i][jj] = cn;
            hn[i][jj] = og * fast_tanh(cn);
            if (KEEP) {
                acc[i][4 * jj + 0] = ig; acc[i][4 * jj + 1] = fg; acc[i][4 * jj + 2] = gg; acc[i][4 * jj + 3] = og;
            }
        }
    }
}

// The LSTM cell of lstm_cell<false, 4> (kind 0) on PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two values per lane and issue slot).  The (i, f) and (g, o) gate pairs of a unit are neighbours in
// the accumulator; from the cell update on, the units 2u and 2u + 1 of a tile are paired.  Every value goes through
// the same IEEE operations in the same order as in lstm_cell (x * -log2e, exp2, 1 +, rcp, fma), so the results are
// bit-equal; what changes is that 15 non-transcendental instructions per unit become 7.5.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sig_of_scaled(f32x2 t) {       // 1 / (1 + 2^t), per half
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    e = e + (f32x2){1.0f, 1.0f};
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(e.x);
    r.y = __builtin_amdgcn_rcpf(e.y);
    return r;
}
template <int NC = 4>
__device__ __forceinline__ void lstm_cell_pk(f32x16 (&acc)[NC], float (&c)[NC][4], float (&hn)[NC][4]) {
    constexpr float kS = -1.4426950408889634f, kT = -2.885390081777927f;   // sigmoid / tanh exponent scales
    const f32x2 one = {1.0f, 1.0f}, two = {2.0f, 2.0f};
#pragma unroll
    for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x2 ig, fg, rg, og;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int jj = 2 * u + v;
                const f32x2 sif = pk_sig_of_scaled((f32x2){acc[i][4 * jj + 0], acc[i][4 * jj + 1]} * (f32x2){kS, kS});
                const f32x2 sgo = pk_sig_of_scaled((f32x2){acc[i][4 * jj + 2], acc[i][4 * jj + 3]} * (f32x2){kT, kS});
                ig[v] = sif.x; fg[v] = sif.y; rg[v] = sgo.x; og[v] = sgo.y;
            }
            const f32x2 gg = __builtin_elementwise_fma(rg, two, -one);            // tanh = 2 sigmoid(2x) - 1
            const f32x2 cn = __builtin_elementwise_fma(fg, (f32x2){c[i][2 * u], c[i][2 * u + 1]}, ig * gg);
            const f32x2 th = __builtin_elementwise_fma(pk_sig_of_scaled(cn * (f32x2){kT, kT}), two, -one);
            const f32x2 h = og * th;
            c[i][2 * u] = cn.x; c[i][2 * u + 1] = cn.y;
            hn[i][2 * u] = h.x; hn[i][2 * u + 1] = h.y;
        }
    }
}

// Decoder head on the top-layer h of this step: layer li's output tile `wave` is computed by
// wave `wave`; activations ping-pong between two LDS buffers (64 slots each) with one barrier
// per layer; the next layer's weights are fetched before that barrier.  Every wave of the
// workgroup must call this.
template <int NT>
__device__ __forceinline__ void head_preload(const LstmSeqArgs& a, float4 (&wq)[kHalfQ], float4 (&bq)[4], int lane, int wave) {
    const DenseChainArgs& hd = a.head;
    const int hh = lane >> 5;
    const int tw = wave < hd.layer[0].nt ? wave : hd.layer[0].nt - 1;
    tile_load<NT>(wq, (const float4*)hd.layer[0].w, tw * 64 + lane, hd.layer[0].nt * 64);
    const float4* bp = (const float4*)(hd.layer[0].b + tw * 32 + hh * 16);
#pragma unroll
    for (int v = 0; v < 4; ++v) bq[v] = bp[v];
}

// wq / bq: weights and bias of the first head layer, fetched by head_preload (before the last LSTM
// layer's cell and barriers, so they have landed by now).  htop0 / ha0: LDS bases WITHOUT the lane
// offset; every layer re-derives its lane-dependent addresses from an opaque copy of the lane index so
// that they are not kept alive (in scratch) across the whole time loop.
template <int NT>
__device__ __forceinline__ void fused_head(const LstmSeqArgs& a, float4 (&wq)[kHalfQ], float4 (&bq)[4], const float* htop0,
                                           float* ha0, int lane, int wave, long rowtile, int to) {
    const DenseChainArgs& hd = a.head;
    const int nl = hd.n_layers;
#pragma unroll 1
    for (int li = 0; li < nl; ++li) {
        int lv = lane;
        asm volatile("" : "+v"(lv));
        const int hh = lv >> 5;
        const DenseLayerArg& ly = hd.layer[li];
        const bool last = (li == nl - 1);
        const bool mine = wave < ly.nt;
        const float* src = (li == 0 ? htop0 : ha0 + (size_t)((li - 1) & 1) * kDenseSlots * 64) + lv;
        float* dst = ha0 + (size_t)(li & 1) * kDenseSlots * 64 + lv;
        f32x16 acc[1];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            acc[0][4 * v + 0] = bq[v].x; acc[0][4 * v + 1] = bq[v].y; acc[0][4 * v + 2] = bq[v].z; acc[0][4 * v + 3] = bq[v].w;
        }
        if (mine) {
            const unsigned off = wave * 64 + lv;
            if (li == 0) tile_mfma<NT>(acc[0], wq, (const float4*)ly.w, off, ly.nt * 64, src);
            else tile_mfma_band<4 * kHeadTiles>(acc[0], wq, (const float4*)ly.w, off, ly.nt * 64, src,
                                                   pick4(ly.q0, wave), pick4(ly.nqb, wave));
        }
        if (!last) {
            const DenseLayerArg& nx = hd.layer[li + 1];
            const int tw = wave < nx.nt ? wave : nx.nt - 1;
            tile_load_band(wq, (const float4*)nx.w, tw * 64 + lv, nx.nt * 64, pick4(nx.q0, tw), pick4(nx.nqb, tw));
            const float4* bp = (const float4*)(nx.b + tw * 32 + hh * 16);
#pragma unroll
            for (int v = 0; v < 4; ++v) bq[v] = bp[v];
        }
        if (mine) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (feat_of(wave, r, hh) < ly.n_relu) acc[0][r] = fmaxf(acc[0][r], 0.0f);
            if (!last) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(16 * wave + r) * 64] = acc[0][r];
            }
            const bool do_aux = (li == hd.aux_layer) && hd.aux != nullptr;
            if (do_aux || last) {
                const RowIo io = late_rowio(hd, rowtile, lv, to);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 32 * wave + 8 * g;                     // wave-uniform group of 8 features
                    if (do_aux && io.aux && f0 + 8 > hd.aux_f0 && f0 < hd.aux_f0 + hd.aux_n) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int f = f0 + 4 * hh + k;
                            if (f >= hd.aux_f0 && f < hd.aux_f0 + hd.aux_n) io.aux[f] = acc[0][4 * g + k];
                        }
                    }
                    if (last && io.out && f0 < hd.out_f) {
                        float* po = io.out + (size_t)(f0 + 4 * hh) * io.fstride;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (f0 + 4 * hh + k < hd.out_f) po[(size_t)k * io.fstride] = acc[0][4 * g + k];
                    }
                }
            }
        }
        if (!last) __syncthreads();
#ifdef VSV_STAMPS
        if (lane == 0 && blockIdx.x < kStampWgs) g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 8 + li] = __builtin_amdgcn_s_memtime();
#endif
    }
}

#ifndef VSV_LSTM_OCC
#define VSV_LSTM_OCC 2
#endif
template <int NT, int NL, int XMODE, bool HEAD, bool SAVE, int OWN>
__global__ void __launch_bounds__((OWN == 4) ? 256 : 64 * NT, (OWN == 4) ? VSV_LSTM_OCC : (NT + 3) / 4)
lstm_kernel(const LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(NT <= 16 && (OWN == 4 || OWN == 1), "hidden size <= 128");
    static_assert(!HEAD || OWN == 4, "the fused head runs on the 4-wave schedule");
    constexpr int NW = (OWN == 4) ? 4 : NT;                       // waves per workgroup
    constexpr int HPK = 4 * NT;                                   // k-steps of one layer's h
    constexpr int HP = 8 * NT;                                    // padded hidden units
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    VSV_STAMP(0);
#ifdef VSV_STAMPS
    if (XMODE == 1 && HEAD && lane == 0 && blockIdx.x < kStampWgs)
        g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 15] =
            (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#endif
    const long rowtile = (long)blockIdx.x;
    const int hh = lane >> 5, m = lane & 31;
    const long row_raw = rowtile * 32 + m;
    const bool row_ok = row_raw < a.n_rows;
    const long row = row_ok ? row_raw : a.n_rows - 1;
    const int kx = a.kx;
    const int xk = (XMODE == 0) ? 0 : kx;                         // x k-steps staged in LDS (decoder only)
    float* const xs0 = smem;                                      // LDS bases without the lane offset
    float* const hb0 = xs0 + (size_t)xk * 64;                     // h of layer l: hb + l*HPK*64
    float* const ha0 = hb0 + (size_t)NL * HPK * 64;               // head activations (HEAD only)
    float* const xs = xs0 + lane;                                 // prologue / epilogue views
    float* const hb = hb0 + lane;
    const int SL = (SAVE && a.sv_nl > 0) ? a.sv_nl : NL;          // layers per instance in the saved arrays
    const int L0 = (SAVE && a.sv_nl > 0) ? a.sv_l0 : 0;           // the model's layer that is l = 0 here
    const bool rec = SAVE && a.rec != 0;                          // wave-uniform
    const int SNL = a.st_nl > 0 ? a.st_nl : NL;                   // layers per row in the state arrays
    const bool dropping = SAVE && SL > 1 && a.drop.p > 0.0f;      // wave-uniform; the dropped h lives at ha0
    const DropArgs drop = (SAVE && dropping) ? resolve_drop(a.drop) : a.drop;

    // gate tiles of this wave; out-of-range ones are computed on a clamped tile and never stored
    // (opaque<OWN == 4>: the one-wave-per-gate-tile schedule has registers to spare, and pinning its
    // address arithmetic to the phases measured 60% slower on the date-range prefix)
    constexpr bool ALLV = (OWN * NW == NT);                       // every wave owns OWN existing tiles
    int toff[OWN];
    unsigned wo[OWN];
    bool tval[OWN];
#pragma unroll
    for (int i = 0; i < OWN; ++i) {
        const int tl = wave + NW * i;
        tval[i] = ALLV || tl < NT;
        toff[i] = (tval[i] ? tl : NT - 1) * 64;
        wo[i] = (unsigned)lane + (unsigned)toff[i];
    }

    DecIn di = {};
    const float* gate_in = nullptr;
    // rows fit 31 bits (checked by the launcher): 32-bit divisions; a divisor beyond that gives quotient 0
    const uint32_t row32 = (uint32_t)row;
    auto div32 = [](long dv) { return (uint32_t)(dv > 0x7fffffffL ? 0x7fffffffL : dv); };
    uint32_t state_row = 0;
    if (XMODE == 1) {
        const vsv_decode_in& d = a.din;
        const uint32_t day_rel = row32 / div32(d.S);
        // (wave-uniform tests: a division costs ~25 instructions, and instruction issue is what the
        // prologue is made of)
        if (d.ctx) di.ctx = d.ctx + (long)(d.ctx_div == 1 ? row32 : row32 / div32(d.ctx_div)) * d.ctx_rs;
        if (d.zb) di.z_idx = (long)(d.z_div == 1 ? row32 : row32 / div32(d.z_div)) * d.z_rs;
        di.eps = d.eps_in ? d.eps_in + (size_t)row * d.eps_T * a.L : nullptr;
        di.sample = row32 - day_rel * div32(d.S);
        di.day = (uint32_t)d.day0 + day_rel;
        state_row = (d.state_div == d.S) ? day_rel : row32 / div32(d.state_div);
        if (d.gate_in) gate_in = d.gate_in + (size_t)state_row * NL * NT * 32;
    } else if (a.din.state_in) {
        // encoders continuing from a shared state (autoregressive rollouts: the part of a row's window
        // that is still the day's original context was run once per day)
        state_row = a.din.state_div == 1 ? row32 : row32 / div32(a.din.state_div);
    }

    // Gate accumulators start from the bias, or (first step from a shared state) from bias + W_hh . h_prev,
    // computed once per state row.  The loads are issued one stage early -- at kernel entry, or right
    // after the previous layer's cell has consumed the accumulators -- so the barriers hide them.
    f32x16 acc[OWN];
    float4 a0[OWN];                                               // quad 0 of the next gate GEMM (OWN == 4)
    auto acc_init = [&](int l, bool shared, int lv, int t) {
        if (XMODE == 0 && !SAVE && l == 0 && a.pre_t) {
            const float* p = a.pre_t + (size_t)(lv >> 5) * a.pre_stride + row32 + (size_t)t * a.pre_tstride;
#pragma unroll
            for (int i = 0; i < OWN; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = p[(size_t)(2 * ((toff[i] >> 2) + r)) * a.pre_stride];
            return;
        }
        if (XMODE == 0 && !SAVE && l == 0 && a.pre_g) {
            const float* p = a.pre_g + (size_t)(rowtile * a.T + t) * (16 * NT) * 64 + lv;
#pragma unroll
            for (int i = 0; i < OWN; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = p[((toff[i] >> 2) + r) * 64];
            return;
        }
        if (SAVE && rec) {
            const float* p = a.sv_gates + ((size_t)(rowtile * a.T + t) * SL + L0 + l) * (16 * NT) * 64 + lv;
#pragma unroll
            for (int i = 0; i < OWN; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = p[((toff[i] >> 2) + r) * 64];
            return;
        }
        const int hh = lv >> 5;
        const float* p0 = shared ? gate_in + (size_t)l * NT * 32 : a.b[l];
#pragma unroll
        for (int i = 0; i < OWN; ++i) {
            const float4* p = (const float4*)(p0 + (toff[i] >> 6) * 32 + hh * 16);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 x = p[v];
                acc[i][4 * v + 0] = x.x; acc[i][4 * v + 1] = x.y; acc[i][4 * v + 2] = x.z; acc[i][4 * v + 3] = x.w;
            }
        }
    };
    const bool shared0 = (XMODE == 1) && a.din.gate_in != nullptr;      // wave-uniform
    VSV_STAMP(16);
    acc_init(0, shared0, lane, 0);
    // Monte-Carlo first step from a shared state (a prior draw): the context half of [ctx | z] is the zero
    // padding and what follows z is alignment padding, so of layer 0's input quads only those that hold z
    // are multiplied (L = 12: quads 1-2 of 4)
    int qx0 = 0, qxn = kx >> 2;
    if (XMODE == 1 && !SAVE && OWN == 4 && shared0 && a.din.ctx_T <= 0 && a.din.z_T <= 0 && !a.din.eps_in && !a.z_out) {
        qx0 = a.Lc >> 3;
        qxn = ((a.Lc + a.L + 7) >> 3) - qx0;
    }
    // (with a pre-activation table layer 0 streams its recurrent quads only)
    const size_t w0_skip = (XMODE == 0 && !SAVE && (a.pre_t || a.pre_g)) ? (size_t)(kx >> 2) * NT * 64 : 0;
    if (OWN == 4) wload<OWN>(a0, (const float4*)a.w[0] + w0_skip + (size_t)qx0 * NT * 64, wo, 0u);

    // state: (h | c) per layer, each HP floats in lane order (position 8*tile + 4*hh + jj holds unit
    // 8*tile + 2*jj + hh), so a lane's four units of a tile are one float4
    float c[NL][OWN][4];
    {
        const bool has_state = a.din.state_in != nullptr;                  // wave-uniform
        const float* st = has_state ? a.din.state_in + ((size_t)state_row * SNL + a.st_l0) * 2 * HP : nullptr;
        float4 hv[NL][OWN], cv[NL][OWN];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                hv[l][i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                cv[l][i] = hv[l][i];
            }
        if (has_state) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int i = 0; i < OWN; ++i) {
                    const float* ps = st + (size_t)l * 2 * HP + (toff[i] >> 3) + 4 * hh;
                    if (!shared0) hv[l][i] = *(const float4*)ps;      // with gate_in the old h is never read
                    cv[l][i] = *(const float4*)(ps + HP);
                }
        }
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                const float hq[4] = {hv[l][i].x, hv[l][i].y, hv[l][i].z, hv[l][i].w};
                const float cq[4] = {cv[l][i].x, cv[l][i].y, cv[l][i].z, cv[l][i].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    // (a shared first step never reads the old h: no need to stage it)
                    if (tval[i] && !shared0) hb[(l * HPK + (toff[i] >> 4) + jj) * 64] = hq[jj];
                    c[l][i][jj] = tval[i] ? cq[jj] : 0.0f;
                }
            }
    }
    VSV_STAMP(17);
    // x part of layer 0 for step 0 (slots wave, wave+4, ...: the waves share the assembly)
    if (XMODE == 1) {
        const vsv_decode_in& d = a.din;
        if (!SAVE && d.ctx_T <= 0 && d.z_T <= 0 && !d.eps_in && !a.z_out) {
            // Monte-Carlo fast path (prior draw from a day's prefix state): the context half of the input
            // is the zero padding, the latent half mu0 + std0 * Philox
            for (int s = wave; s < kx; s += NW) {
                const int f = 2 * s + hh - a.Lc;
                float v = 0.0f;
                if (f >= 0 && f < a.L) {
                    v = d.mu0;
                    if (d.std0 != 0.0f)
                        v += d.std0 * philox_normal_elem((uint32_t)d.seed, (uint32_t)(d.seed >> 32), di.sample, di.day,
                                                         (uint32_t)(d.step0 - d.eps_t0), f);
                }
                xs[s * 64] = v;
            }
        } else {
            for (int s = wave; s < kx; s += NW) {
                const float v = decoder_input(a, di, row, row_ok, true, 0, 2 * s + hh);
                xs[s * 64] = v;
                if (SAVE) a.sv_x[((size_t)(rowtile * a.T) * kx + s) * 64 + lane] = v;
            }
        }
    }
    VSV_STAMP(18);
    __syncthreads();
    VSV_STAMP(1);

    const int t_outs = a.T - a.t_out0;
    float4 wq[HEAD ? kHalfQ : 1], bq[HEAD ? 4 : 1];               // first head layer, fetched ahead
    for (int t = 0; t < a.T; ++t) {
        const bool shared = shared0 && t == 0;
        const bool head_now = HEAD && t >= a.t_out0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            float hn[OWN][4];
            const float4* w = (const float4*)a.w[l];
            const int nq_in = (l == 0) ? (kx >> 2) : NT;          // quads of the input segment
            const int lv = opaque<OWN == 4>(lane);
            float* const xs = xs0 + lv;
            float* const hb = hb0 + lv;
            float* hl = hb + (size_t)l * HPK * 64;
            unsigned wo[OWN];
#pragma unroll
            for (int i = 0; i < OWN; ++i) wo[i] = (unsigned)lv + (unsigned)toff[i];
            if constexpr (OWN == 4) {
                // one A stream over [input part | recurrent part]; a shared first step has no recurrent part
                const int nq = shared ? nq_in : nq_in + NT;
                if constexpr (XMODE == 0) {
                    if (l == 0) {
                        const float* xg = a.x_tileb + ((size_t)(rowtile * a.T + t) * kx) * 64 + lv;
                        const int qp = w0_skip ? nq_in : 0;          // input quads already in the accumulators
                        gemm_stream<OWN>(acc, a0, w + w0_skip, wo, NT * 64, nq - qp, [&](int q, float (&bv)[4]) {
                            const int qq = q + qp;
                            if (qq < nq_in) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) bv[j] = xg[(4 * qq + j) * 64];
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) bv[j] = hl[(4 * (qq - nq_in) + j) * 64];
                            }
                        });
                    }
                }
                if (XMODE == 1 || l > 0) {
                    // (a shared Monte-Carlo first step streams only layer 0's input quads that hold z)
                    const bool zq = XMODE == 1 && l == 0 && shared;
                    const int q0 = zq ? qx0 : 0, nqs = zq ? qxn : nq, nqi = nq_in - q0;
                    const float* bin = ((l == 0) ? xs : ((SAVE && dropping) ? ha0 + lv : hb + (size_t)(l - 1) * HPK * 64)) +
                                       (size_t)(4 * q0) * 64;
                    gemm_stream<OWN>(acc, a0, w + (size_t)q0 * NT * 64, wo, NT * 64, nqs, [&](int q, float (&bv)[4]) {
                        const float* bp = (q < nqi) ? bin + (size_t)(4 * q) * 64 : hl + (size_t)(4 * (q - nqi)) * 64;
#pragma unroll
                        for (int j = 0; j < 4; ++j) bv[j] = bp[j * 64];
                    });
                }
                if (l + 1 < NL) wload<OWN>(a0, (const float4*)a.w[l + 1], wo, 0u);
                else if (!head_now && t + 1 < a.T) wload<OWN>(a0, (const float4*)a.w[0] + w0_skip, wo, 0u);
            } else {
                const float* bin;
                if (l == 0) bin = (XMODE == 0) ? a.x_tileb + ((size_t)(rowtile * a.T + t) * kx) * 64 + lv : xs;
                else bin = (SAVE && dropping) ? ha0 + lv : hb + (size_t)(l - 1) * HPK * 64;
                if (!(SAVE && rec) && !(XMODE == 0 && !SAVE && l == 0 && (a.pre_t || a.pre_g)))
                    gemm_seg<OWN>(acc, w, lv, toff, NT * 64, bin, nq_in);
                if (!shared) gemm_seg<OWN>(acc, w + (size_t)nq_in * NT * 64, lv, toff, NT * 64, hl, NT);
            }
            if constexpr (HEAD) {
                if (l == NL - 1 && head_now) head_preload<NT>(a, wq, bq, opaque<OWN == 4>(lane), wave);
            }
            VSV_STAMP(2 + 3 * l);
            lstm_cell<SAVE, OWN>(acc, c[l], hn, a.gru);
            VSV_STAMP(3 + 3 * l);
            if (SAVE) {
                const size_t il = (size_t)(rowtile * a.T + t) * SL + L0 + l;
                float* sg = a.sv_gates + il * (16 * NT) * 64 + lane;
                float* sc = a.sv_c + il * HPK * 64 + lane;
                float* sh = a.sv_h + il * HPK * 64 + lane;
#pragma unroll
                for (int i = 0; i < OWN; ++i) {
                    if (tval[i]) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sg[((toff[i] >> 2) + r) * 64] = acc[i][r];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            sc[((toff[i] >> 4) + jj) * 64] = c[l][i][jj];
                            sh[((toff[i] >> 4) + jj) * 64] = hn[i][jj];
                        }
                    }
                }
            }
            float hd[OWN][4];
            if (SAVE && L0 + l + 1 < SL && dropping) {
#pragma unroll
                for (int i = 0; i < OWN; ++i) {
                    float fct[4];
                    dropout_factors(drop, row32, (uint32_t)t, toff[i] >> 6, hh, L0 + l, fct);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) hd[i][jj] = hn[i][jj] * fct[jj];
                    if (tval[i]) {
                        float* sd = a.sv_hd + (size_t)(rowtile * a.T + t) * HPK * 64 + lane;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) sd[((toff[i] >> 4) + jj) * 64] = hd[i][jj];
                    }
                }
            }
            // the accumulators are free: start the next stage's initial values
            if (l + 1 < NL) acc_init(l + 1, shared, opaque<OWN == 4>(lane), t);
            else if (!head_now && t + 1 < a.T) acc_init(0, false, opaque<OWN == 4>(lane), t + 1);
            // h of this layer is overwritten only after every wave has read the old one (a shared first
            // step has no recurrent product, so nobody reads it)
            if (!shared) __syncthreads();
            const int lw = opaque<OWN == 4>(lane);
            float* const hlw = hb0 + lw + (size_t)l * HPK * 64;
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                if (tval[i]) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) hlw[((toff[i] >> 4) + jj) * 64] = hn[i][jj];
                    if (SAVE && l + 1 < NL && dropping) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) (ha0 + lw)[((toff[i] >> 4) + jj) * 64] = hd[i][jj];
                    }
                }
            }
            if (XMODE == 1 && l == NL - 1 && t + 1 < a.T) {
                // x part of the NEXT step: xs was last read before this layer's first barrier
                for (int s = wave; s < kx; s += NW) {
                    const float v = decoder_input(a, di, row, row_ok, true, t + 1, 2 * s + (lw >> 5));
                    (xs0 + lw)[s * 64] = v;
                    if (SAVE) a.sv_x[((size_t)(rowtile * a.T + t + 1) * kx + s) * 64 + lane] = v;
                }
            }
            __syncthreads();
            VSV_STAMP(4 + 3 * l);
        }
        // ---- outputs
        if (a.hout && t >= a.t_out0) {
            const int lo = opaque<OWN == 4>(lane);
            const float* htop = hb0 + lo + (size_t)(NL - 1) * HPK * 64;
            float* ho = a.hout + ((size_t)(rowtile * t_outs + (t - a.t_out0)) * HPK) * 64 + lo;
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                if (tval[i]) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) ho[((toff[i] >> 4) + jj) * 64] = htop[((toff[i] >> 4) + jj) * 64];
                }
            }
        }
        if constexpr (HEAD) {
            if (head_now) {
                fused_head<NT>(a, wq, bq, hb0 + (size_t)(NL - 1) * HPK * 64, ha0, lane, wave, rowtile, t - a.t_out0);
                if (t + 1 < a.T) {
                    const int ln = opaque<OWN == 4>(lane);
                    acc_init(0, false, ln, t + 1);
                    if (OWN == 4) {
                        unsigned wn[OWN];
#pragma unroll
                        for (int i = 0; i < OWN; ++i) wn[i] = (unsigned)ln + (unsigned)toff[i];
                        wload<OWN>(a0, (const float4*)a.w[0] + w0_skip, wn, 0u);
                    }
                }
            }
        }
    }
    VSV_STAMP(13);
    // epilogue: per-lane values re-derived from an opaque copy of the lane index (kept alive across the
    // time loop they sat in scratch: the row, its validity, the LDS view)
    const int le = opaque<OWN == 4>(lane);
    const int hhe = le >> 5;
    const long rowe_raw = rowtile * 32 + (le & 31);
    const bool rowe_ok = rowe_raw < a.n_rows;
    const long rowe = rowe_ok ? rowe_raw : a.n_rows - 1;
    float* const hbe = hb0 + le;
    if (a.state_out && rowe_ok) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            float* st = a.state_out + ((size_t)rowe * SNL + a.st_l0 + l) * 2 * HP;
#pragma unroll
            for (int i = 0; i < OWN; ++i) {
                if (tval[i]) {
                    const int s0 = l * HPK + (toff[i] >> 4);
                    *(float4*)(st + (toff[i] >> 3) + 4 * hhe) =
                        make_float4(hbe[(s0 + 0) * 64], hbe[(s0 + 1) * 64], hbe[(s0 + 2) * 64], hbe[(s0 + 3) * 64]);
                    *(float4*)(st + HP + (toff[i] >> 3) + 4 * hhe) = make_float4(c[l][i][0], c[l][i][1], c[l][i][2], c[l][i][3]);
                }
            }
        }
    }
    if (a.din.gate_out) {
        // bias + W_hh . h_final of every layer: the part of the NEXT step's gates that all rows
        // continuing from this state share (consumed through gate_in)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x16 acc[OWN];
            const int nq_in = (l == 0) ? (kx >> 2) : NT;
            load_bias<OWN>(acc, a.b[l], toff, hhe);
            gemm_seg<OWN>(acc, (const float4*)a.w[l] + (size_t)nq_in * NT * 64, le, toff, NT * 64,
                           hbe + (size_t)l * HPK * 64, NT);
            if (rowe_ok) {
#pragma unroll
                for (int i = 0; i < OWN; ++i) {
                    if (tval[i]) {
                        float4* p = (float4*)(a.din.gate_out + (((size_t)rowe * SNL + a.st_l0 + l) * NT + (toff[i] >> 6)) * 32 + hhe * 16);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            p[v] = make_float4(acc[i][4 * v + 0], acc[i][4 * v + 1], acc[i][4 * v + 2], acc[i][4 * v + 3]);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// lstm_rec_kernel: ONE memory layer whose gate pre-activations b + W_ih . x are given for every step (per-calendar-
// day table pre_t, per-instance pre_g, or -- training, SAVE -- the layer's slots of sv_gates), for few row tiles
// (<= 256): one wave per gate tile, NT waves per 32-row workgroup.  Only W_hh . h is on the time loop, and the
// wave's W_hh fragments (NT quads = 4*NT registers) stay in REGISTERS for all T steps: a step streams nothing from
// L2 and is bound by the CU's matrix pipe (NT*NT quads of four products over four SIMDs) plus one barrier -- h is
// double-buffered in LDS, so a step's writers need not wait for the previous step's readers.
// Same products in the same order as lstm_kernel's per-layer launches: results are bit-equal.
// ---------------------------------------------------------------------------
template <int NT, bool SAVE>
__global__ void __launch_bounds__(64 * NT, (NT + 3) / 4) lstm_rec_kernel(const LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HPK = 4 * NT, HP = 8 * NT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long rowtile = (long)blockIdx.x;
    const int hh = lane >> 5, m = lane & 31;
    const long row_raw = rowtile * 32 + m;
    const bool row_ok = row_raw < a.n_rows;
    const long row = row_ok ? row_raw : a.n_rows - 1;
    const uint32_t row32 = (uint32_t)row;
    const int SL = (SAVE && a.sv_nl > 0) ? a.sv_nl : 1;           // layers per instance in the saved arrays
    const int L0 = (SAVE && a.sv_nl > 0) ? a.sv_l0 : 0;           // the model's layer this launch runs
    const int SNL = a.st_nl > 0 ? a.st_nl : 1;                    // layers per row in the state arrays
    const bool dropping = SAVE && L0 + 1 < SL && a.drop.p > 0.0f;
    const DropArgs drop = dropping ? resolve_drop(a.drop) : a.drop;
    // (offsets into smem, not an array of two pointers: selecting a pointer at run time loses the LDS address
    // space and every h read becomes a flat load with full waits)
    float* const hbase = smem + lane;
    const int nq_in = a.kx >> 2;

    // recurrent fragments of this wave's gate tile
    float4 wreg[NT];
    {
        const float4* w = (const float4*)a.w[0] + (size_t)nq_in * NT * 64 + wave * 64 + lane;
#pragma unroll
        for (int q = 0; q < NT; ++q) wreg[q] = w[(size_t)q * NT * 64];
    }
    auto load_pre = [&](f32x16& acc, int t) {
        if (!SAVE && a.pre_t) {
            // (a running pointer: sixteen 64-bit addresses at once do not fit beside the resident weights)
            const float* p = a.pre_t + ((size_t)hh + (size_t)(32 * wave)) * a.pre_stride + row32 + (size_t)t * a.pre_tstride;
            const size_t step = 2 * (size_t)a.pre_stride;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] = *p;
                p += step;
                asm volatile("" : "+v"(p));
            }
        } else {
            const float* p = SAVE ? a.sv_gates + ((size_t)(rowtile * a.T + t) * SL + L0) * (16 * NT) * 64 + lane
                                  : a.pre_g + (size_t)(rowtile * a.T + t) * (16 * NT) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = p[(16 * wave + r) * 64];
        }
    };
    f32x16 acc[1];
    load_pre(acc[0], 0);

    // state: (h | c), position 8*tile + 4*hh + jj holds unit 8*tile + 2*jj + hh
    float c[1][4];
    const bool has_state = a.din.state_in != nullptr;              // wave-uniform
    {
        float4 hv = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cv = hv;
        if (has_state) {
            const uint32_t srow = a.din.state_div == 1 ? row32 : row32 / (uint32_t)a.din.state_div;
            const float* ps = a.din.state_in + ((size_t)srow * SNL + a.st_l0) * 2 * HP + 8 * wave + 4 * hh;
            hv = *(const float4*)ps;
            cv = *(const float4*)(ps + HP);
        }
        const float hq[4] = {hv.x, hv.y, hv.z, hv.w};
        const float cq[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            hbase[(4 * wave + jj) * 64] = hq[jj];
            c[0][jj] = cq[jj];
        }
    }
    __syncthreads();
    const int t_outs = a.T - a.t_out0;
    for (int t = 0; t < a.T; ++t) {
        const float* hr = hbase + (t & 1) * (HPK * 64);
        float* hw = hbase + ((t + 1) & 1) * (HPK * 64);
        if (t > 0 || has_state) {                                  // (h = 0 at the first step of a fresh sequence)
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const float b0 = hr[(4 * q + 0) * 64], b1 = hr[(4 * q + 1) * 64], b2 = hr[(4 * q + 2) * 64],
                            b3 = hr[(4 * q + 3) * 64];
                acc[0] = mfma32(wreg[q].x, b0, acc[0]);
                acc[0] = mfma32(wreg[q].y, b1, acc[0]);
                acc[0] = mfma32(wreg[q].z, b2, acc[0]);
                acc[0] = mfma32(wreg[q].w, b3, acc[0]);
            }
        }
        float hn[1][4];
        lstm_cell<SAVE, 1>(acc, c, hn, a.gru);
        if (SAVE) {
            const size_t il = (size_t)(rowtile * a.T + t) * SL + L0;
            float* sg = a.sv_gates + il * (16 * NT) * 64 + lane;
            float* sc = a.sv_c + il * HPK * 64 + lane;
            float* sh = a.sv_h + il * HPK * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) sg[(16 * wave + r) * 64] = acc[0][r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                sc[(4 * wave + jj) * 64] = c[0][jj];
                sh[(4 * wave + jj) * 64] = hn[0][jj];
            }
            if (dropping) {
                float fct[4];
                dropout_factors(drop, row32, (uint32_t)t, wave, hh, L0, fct);
                float* sd = a.sv_hd + (size_t)(rowtile * a.T + t) * HPK * 64 + lane;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) sd[(4 * wave + jj) * 64] = hn[0][jj] * fct[jj];
            }
        }
        if (a.hout && t >= a.t_out0) {
            float* ho = a.hout + ((size_t)(rowtile * t_outs + (t - a.t_out0)) * HPK) * 64 + lane;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ho[(4 * wave + jj) * 64] = hn[0][jj];
        }
        if (t + 1 < a.T) load_pre(acc[0], t + 1);                  // lands while the waves meet
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) hw[(4 * wave + jj) * 64] = hn[0][jj];
        __syncthreads();                                           // h(t) published (the other buffer is free again)
    }
    const float* hf = hbase + (a.T & 1) * (HPK * 64);
    if (a.state_out && row_ok) {
        float* st = a.state_out + ((size_t)row_raw * SNL + a.st_l0) * 2 * HP + 8 * wave + 4 * hh;
        *(float4*)st = make_float4(hf[(4 * wave + 0) * 64], hf[(4 * wave + 1) * 64], hf[(4 * wave + 2) * 64], hf[(4 * wave + 3) * 64]);
        *(float4*)(st + HP) = make_float4(c[0][0], c[0][1], c[0][2], c[0][3]);
    }
    if (a.din.gate_out) {
        // bias + W_hh . h_final: the part of the next step's gates shared by all rows continuing from this state
        const int toff1[1] = {wave * 64};
        f32x16 g[1];
        load_bias<1>(g, a.b[0], toff1, hh);
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            g[0] = mfma32(wreg[q].x, hf[(4 * q + 0) * 64], g[0]);
            g[0] = mfma32(wreg[q].y, hf[(4 * q + 1) * 64], g[0]);
            g[0] = mfma32(wreg[q].z, hf[(4 * q + 2) * 64], g[0]);
            g[0] = mfma32(wreg[q].w, hf[(4 * q + 3) * 64], g[0]);
        }
        if (row_ok) {
            float4* p = (float4*)(a.din.gate_out + (((size_t)row_raw * SNL + a.st_l0) * NT + wave) * 32 + hh * 16);
#pragma unroll
            for (int v = 0; v < 4; ++v) p[v] = make_float4(g[0][4 * v + 0], g[0][4 * v + 1], g[0][4 * v + 2], g[0][4 * v + 3]);
        }
    }
}

// ---------------------------------------------------------------------------
// lstm_rec16_kernel: the recurrent-only launch of ONE nn.LSTM layer (pre-activations given: pre_t or pre_g) on
// 16-ROW tiles with v_mfma_f32_16x16x4_f32, for the date-range prefix: its launches have few row tiles (4000 days
// = 125 tiles of 32 rows = 125 of 256 CUs, each bound by its own matrix pipe for every one of the 2 x 60 sequential
// steps), so halving the tile doubles the CUs in use and halves the matrix time per step.
//   * a 16x16x4 D tile is (16 gate rows) x (16 days); gate tile q holds hidden units 4q .. 4q+3, D row 4*unit + gate,
//     so lane l (day l & 15, unit 4q + (l >> 4)) holds the four gates of ONE unit of ONE day in its four D registers:
//     the cell is lane-local;
//   * the B operand of k-step j (hidden units 4j .. 4j+3) wants unit 4j + (l >> 4) of day l & 15 in lane l -- exactly
//     where the cell of gate tile j left h': h goes through LDS as [k-step][lane], no transposition;
//   * 8 waves per workgroup, wave w owns gate tiles w, w+8, w+16, w+24; its W_hh fragments (one register per tile and
//     k-step: 4 x 32 = 128 at hidden 128) are gathered ONCE from the blob's 32x32x2 section and stay in registers for
//     all T steps; h is double-buffered in LDS; the next step's pre-activations are fetched while the products run;
//   * a step runs in two halves with a barrier each, so that cells and barrier waits have products beside them (see
//     the comment at `run` below and DESIGN.md, "Date-range prefix, 2.50 -> 2.20 ms").
// Reads and writes the 32-row formats of the other kernels (pre_t / pre_g, hout in tile-B, state and gate arrays).
// Every gate sum takes its k-steps in ascending order on one accumulator chain, which the matrix pipe adds one
// product at a time whatever the instruction's shape: bit for bit the sums of lstm_kernel and lstm_rec4_kernel
// (tests/test_gpu_parity.py::test_segmented_context_prefix_equals_single_launches and its neighbours hold it).
// ---------------------------------------------------------------------------
#ifndef VSV_REC16_IA13
#define VSV_REC16_IA13 1
#endif
// f(integral_constant<int, I>) for I = I0 .. I1-1, unrolled at compile time
template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int NT, bool SAVE>
__global__ void __launch_bounds__(512) lstm_rec16_kernel(const LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 2 * NT;                                     // k-steps of 4 hidden units = gate tiles of 16 rows
    constexpr int TPW = (KS + 7) / 8;                              // gate tiles per wave
    constexpr int HP = 8 * NT, HPK = 4 * NT;
    constexpr int HB = (KS + 1) * 64;                              // floats of one h buffer: KS rows + a spare one
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile16 = (long)blockIdx.x;
    const int d = lane & 15, ul = lane >> 4;
    const long row_raw = tile16 * 16 + d;
    const bool row_ok = row_raw < a.n_rows;
    const long row = row_ok ? row_raw : a.n_rows - 1;
    const long rowtile = tile16 >> 1;                              // the 32-row tile of the other kernels' layouts
    const int m = 16 * (int)(tile16 & 1) + d;                      // ... and this lane's row inside it
    const int SNL = a.st_nl > 0 ? a.st_nl : 1;
    const int SL = (SAVE && a.sv_nl > 0) ? a.sv_nl : 1;           // training: layers per instance in the saved arrays
    const int L0 = (SAVE && a.sv_nl > 0) ? a.sv_l0 : 0;           // ... and the model's layer this launch runs
    const bool dropping = SAVE && L0 + 1 < SL && a.drop.p > 0.0f;
    const DropArgs drop = dropping ? resolve_drop(a.drop) : a.drop;
    const int ntile = __builtin_amdgcn_readfirstlane((KS - wave + 7) / 8);     // tiles this wave owns (TPW or TPW - 1)
    float* const hbase = smem + lane;
    // (a Linear head on h rides as wave 2's spare tile, see LstmSeqArgs::hd_out)
    constexpr bool HD_BUILT = !SAVE && NT == 13;
    const bool hd = HD_BUILT && a.hd_out != nullptr && wave == 2;

    // per owned tile: this lane's hidden unit and where its values live in the 32-row layouts
    int gslot[TPW], l32[TPW], hpos[TPW], hslot[TPW], hwo[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int q = (i < ntile) ? wave + 8 * i : KS - 1;
        const int u = 4 * q + ul;
        // where this tile's h goes in an LDS buffer; tiles beyond the wave's own: the spare row behind both buffers
        hwo[i] = ((i < ntile) ? wave + 8 * i : KS) * 64 + lane;
        gslot[i] = 16 * (u >> 3) + 4 * ((u & 7) >> 1);             // + gate: slot of the gate value
        l32[i] = (u & 1) * 32 + m;                                 // lane of the 32-row layouts
        hslot[i] = 4 * (u >> 3) + ((u & 7) >> 1);                  // slot of h in tile-B
        hpos[i] = 8 * (u >> 3) + 4 * (u & 1) + ((u & 7) >> 1);     // position in the state arrays
    }

    // W_hh fragments: A[i][k] of tile q, k-step j = W_hh[gate row of (unit 4q + (i >> 2), gate i & 3)][unit 4j + k],
    // lane = i + 16 k, gathered from the blob's 32x32x2 fragment layout (vsv_layout.h)
    float wreg[TPW][KS];
    {
        const int ir = lane & 15;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int q = (i < ntile) ? wave + 8 * i : KS - 1;
            const int ua = 4 * q + (ir >> 2);
            const int n = 32 * (ua >> 3) + 4 * (ua & 7) + (ir & 3);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int kk = 2 * a.kx + 4 * j + ul;
                wreg[i][j] = a.w[0][((size_t)((kk >> 3) * NT + (n >> 5)) * 64 + ((kk & 1) * 32 + (n & 31))) * 4 + ((kk >> 1) & 3)];
            }
        }
        if (hd) {
            // the head's rows as the wave's spare tile: A[i][k] = W_head[output i][unit 4j + k] from the dense layer's
            // fragments ([quad][tile 0][lane half * 32 + output][k-step of the quad], k-step s = units 2s, 2s + 1)
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const int u = 4 * j + ul;
                wreg[TPW - 1][j] = a.hd_w[((size_t)(u >> 3) * 64 + ((u & 1) * 32 + ir)) * 4 + ((u >> 1) & 3)];
            }
        }
    }
    // Per-step traffic is buffer-addressed: descriptor base (this row tile's part of the array) + a lane offset that
    // never changes + a scalar step offset, so that a load or store costs no vector instruction and no address pair.
    const bool from_t = !SAVE && a.pre_t != nullptr;               // wave-uniform
    const float* pbase = SAVE ? a.sv_gates + ((size_t)rowtile * a.T * SL + L0) * (16 * NT) * 64
                       : from_t ? a.pre_t : a.pre_g + (size_t)rowtile * a.T * (16 * NT) * 64;
    const wrsrc_t prs = make_wrsrc(pbase);
    const unsigned pstep = SAVE ? (unsigned)SL * (16 * NT) * 256u : from_t ? (unsigned)a.pre_tstride * 4u : (16 * NT) * 256u;
    const unsigned pgate = from_t ? 8u * (unsigned)a.pre_stride : 256u;
    unsigned pvo[TPW], hvo[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        pvo[i] = from_t ? 4u * ((unsigned)(2 * gslot[i] + (l32[i] >> 5)) * (unsigned)a.pre_stride + (unsigned)row)
                        : 4u * (unsigned)(gslot[i] * 64 + l32[i]);
        hvo[i] = 4u * (unsigned)(hslot[i] * 64 + l32[i]);
    }
    const int t_outs = a.T - a.t_out0;
    // (h sequence: a descriptor of 2 GiB where the launch wants it and of nothing where not; a step before t_out0 sets
    // the lane offset's top bit: the hardware drops a store whose lane offset is out of range, so the stores need no
    // branch -- behind one, the compiler's load counters would wait for the stores at the top of every step)
    const wrsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(a.hout ? a.hout + (size_t)rowtile * t_outs * HPK * 64 : a.hout, 0,
                                                          a.hout ? 0x7fffffff : 0, 0x00020000);
    // (training: the arrays the backward kernels read, this row tile's and this layer's part)
    const wrsrc_t svc = make_wrsrc(SAVE ? a.sv_c + ((size_t)rowtile * a.T * SL + L0) * HPK * 64 : nullptr);
    const wrsrc_t svh = make_wrsrc(SAVE ? a.sv_h + ((size_t)rowtile * a.T * SL + L0) * HPK * 64 : nullptr);
    const wrsrc_t svd = make_wrsrc(dropping ? a.sv_hd + (size_t)rowtile * a.T * HPK * 64 : nullptr);
    // the head's bias (this lane: outputs 4 ul .. 4 ul + 3 of its day), where its outputs go and which of them exist
    f32x4 hbias = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned zvo[4] = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    if (hd) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 4 * ul + r;
            hbias[r] = a.hd_b[(ul & 1) * 16 + 4 * (ul >> 1) + r];
            if (j < 2 * a.hd_slots) zvo[r] = 4u * (unsigned)((j >> 1) * 64 + (j & 1) * 32 + m);
        }
    }
    const wrsrc_t hds = __builtin_amdgcn_make_buffer_rsrc(hd ? a.hd_out + (size_t)rowtile * a.T * a.hd_slots * 64 : a.hd_out, 0,
                                                          hd ? 0x7fffffff : 0, 0x00020000);
    auto head_store = [&](const f32x4& v, int ts) {
        const unsigned skip = ts >= 0 ? 0u : 0x80000000u;
        const unsigned so = (unsigned)ts * (unsigned)(a.hd_slots * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) bstore1(hds, (4 * ul + r < a.hd_n) ? v[r] : 0.0f, zvo[r] | skip, so);
    };
    auto load_pre = [&](f32x4 (&pre)[TPW], int t) {
        const unsigned so = (unsigned)t * pstep;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
            pre[i] = f32x4{bload1(prs, pvo[i], so), bload1(prs, pvo[i], so + pgate), bload1(prs, pvo[i], so + 2 * pgate),
                           bload1(prs, pvo[i], so + 3 * pgate)};
    };
    f32x4 pre[TPW];
    load_pre(pre, 0);

    float c[TPW], h[TPW];
    const bool has_state = a.din.state_in != nullptr;              // wave-uniform
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        c[i] = 0.0f;
        h[i] = 0.0f;
        if (has_state) {
            const long srow = a.din.state_div == 1 ? row : row / a.din.state_div;
            const float* ps = a.din.state_in + ((size_t)srow * SNL + a.st_l0) * 2 * HP + hpos[i];
            h[i] = ps[0];
            c[i] = ps[HP];
        }
        smem[hwo[i]] = h[i];
    }
    __syncthreads();
    // products of k-steps [J0, J1) into the sums of the wave's NTL tiles, k-major (one accumulator chain per tile,
    // ascending k)
    auto products = [&](f32x4 (&acc)[TPW], const float* hr, auto ntl, auto j0, auto j1) {
#pragma unroll
        for (int j = decltype(j0)::value; j < decltype(j1)::value; ++j) {
            const float b = hr[j * 64];
#pragma unroll
            for (int i = 0; i < decltype(ntl)::value; ++i) acc[i] = mfma16(wreg[i][j], b, acc[i]);
        }
    };
    // The step in two halves, so that the cells and the barrier waits have products beside them.  A wave publishes
    // the h of its tiles 0 .. IA-1 (k-steps < KA of the next step) before the step's FIRST barrier and runs the cells
    // of its remaining tiles together with the next step's k-steps < KA, publishing them before the SECOND barrier:
    //   step t:  [late cells of t-1 | k < KA of t]  barrier B(t-1)  [k >= KA of t, tile by tile | early cells of t]  barrier A(t)
    // Every tile's sum still takes its k-steps in ascending order on one accumulator chain: bit for bit the sums of
    // the other recurrent kernels.  The whole recurrence is compiled twice, for waves that own TPW tiles and for
    // those that own TPW - 1 (wave-uniform), so that a step has no branch between its products and its cells.
    // (IA: half of the step's k-steps at 16 gate tiles; at 13 -- 26 k-steps, waves of 3 or 4 tiles -- one early tile,
    // whose cell then has two tiles' products behind it: 2.214 -> 2.200 ms for the 4000-day prefix; -DVSV_REC16_IA13=2
    // builds the other split)
    constexpr int IA = NT == 13 ? VSV_REC16_IA13 : 2, KA = 8 * IA;
    static_assert(KA < KS && IA >= 1 && IA < TPW - 1, "lstm_rec16_kernel: the two-phase step needs more than IA + 1 tiles per wave");
    using std::integral_constant;
    auto run = [&](auto ntl_c, auto young_c, auto hd_c) {
        constexpr int NTL = decltype(ntl_c)::value;                // tiles this wave owns
        constexpr bool HD = decltype(hd_c)::value;                 // ... and the head as one more tile of products
        constexpr int NTP = HD ? NTL + 1 : NTL;
        const integral_constant<int, NTP> ntp_c{};
        // The two waves of a SIMD (w and w + 4) do not share the matrix pipe evenly: the older one's products go first
        // and the younger one's instruction stream waits behind its own first product.  So the older wave issues a
        // half's products first and its cells behind them (they then run beside the younger wave's products), the
        // younger wave its cells first where it has them (beside the older wave's products).
        constexpr bool YOUNG = decltype(young_c)::value;
        // the LSTM cell of tile i on its gate sums of step ts (lane-local: the four gates of one unit of one day)
        auto cell = [&](auto ic, const f32x4& sum, int ts) {
            constexpr int i = decltype(ic)::value;
            const float ig = fast_sigmoid(sum[0]);
            const float fg = fast_sigmoid(sum[1]);
            const float gg = fast_tanh(sum[2]);
            const float og = fast_sigmoid(sum[3]);
            c[i] = __builtin_fmaf(fg, c[i], ig * gg);
            h[i] = og * fast_tanh(c[i]);
            if (SAVE) {
                // activated gates, cell and hidden state of the step, where the backward kernels read them
                // (sv_gates: the slots the pre-activations were read from)
                const unsigned so = (unsigned)ts * pstep;
                bstore1(prs, ig, pvo[i], so);
                bstore1(prs, fg, pvo[i], so + 256u);
                bstore1(prs, gg, pvo[i], so + 512u);
                bstore1(prs, og, pvo[i], so + 768u);
                const unsigned sc = (unsigned)ts * (unsigned)(SL * HPK * 256);
                bstore1(svc, c[i], hvo[i], sc);
                bstore1(svh, h[i], hvo[i], sc);
                if (dropping) {
                    float fct[4];
                    const int u = 4 * (wave + 8 * i) + ul;
                    dropout_factors(drop, (uint32_t)row, (uint32_t)ts, u >> 3, u & 1, L0, fct);
                    bstore1(svd, h[i] * fct[(u & 7) >> 1], hvo[i], (unsigned)ts * (unsigned)(HPK * 256));
                }
            }
        };
        // h of the wave's tiles [I0, I1) of step ts: into the LDS buffer the next step multiplies from and, where the
        // launch wants the sequence, into HBM (padding rows too: the next stage reads whole tiles)
        auto publish = [&](auto i0, auto i1, float* hb, int ts) {
#pragma unroll
            for (int i = decltype(i0)::value; i < decltype(i1)::value; ++i) hb[hwo[i]] = h[i];
            const unsigned skip = ts >= a.t_out0 ? 0u : 0x80000000u;
#pragma unroll
            for (int i = decltype(i0)::value; i < decltype(i1)::value; ++i)
                bstore1(hrs, h[i], hvo[i] | skip, (unsigned)(ts - a.t_out0) * (unsigned)(HPK * 256));
        };
        f32x4 late[TPW - IA];
        auto step = [&](int t, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            float* hb_prev = smem + (t & 1) * HB;                  // h(t-1)
            float* hb_next = smem + ((t + 1) & 1) * HB;            // h(t)
            const float* hr = hb_prev + lane;
            f32x4 acc[TPW];
            REC_STAMP(0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = pre[i];
            if (HD) acc[NTL] = hbias;
            load_pre(pre, t + 1 < a.T ? t + 1 : t);                // lands while the products run
            auto late_cells = [&]() {
                static_for<IA, NTL>([&](auto ic) { cell(ic, late[decltype(ic)::value - IA], t - 1); });
            };
            // (sched_barrier: nothing moves across; the order of the two groups is the point, see YOUNG above)
            if (!FIRST && YOUNG) {
                late_cells();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!FIRST || has_state) products(acc, hr, ntp_c, integral_constant<int, 0>{}, integral_constant<int, KA>{});
            if (!FIRST && !YOUNG) {
                __builtin_amdgcn_sched_barrier(0);
                late_cells();
            }
            if (!FIRST) publish(integral_constant<int, IA>{}, ntl_c, hb_prev, t - 1);
            REC_STAMP(1);
            __syncthreads();                                       // B(t-1): all of h(t-1) published
            REC_STAMP(4);
            if (YOUNG || FIRST) {
                // (tile by tile: the early tiles finish first and their cells run beside the later tiles' products)
                static_for<0, IA>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if (!FIRST || has_state) {
#pragma unroll
                        for (int j = KA; j < KS; ++j) acc[i] = mfma16(wreg[i][j], hr[j * 64], acc[i]);
                    }
                    if (!FIRST && i == 0) __builtin_amdgcn_sched_barrier(0);
                    cell(ic, acc[i], t);
                });
                if (!FIRST || has_state) {
#pragma unroll
                    for (int i = IA; i < NTL; ++i) {
#pragma unroll
                        for (int j = KA; j < KS; ++j) acc[i] = mfma16(wreg[i][j], hr[j * 64], acc[i]);
                    }
                }
                if (!FIRST) {
                    // (the early tiles' cells one instruction per product of the tiles behind them)
#pragma unroll
                    for (int r = 0; r < 32 * IA; ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
                    }
                }
            } else {
#pragma unroll
                for (int j = KA; j < KS; ++j) {
                    const float b = hr[j * 64];
#pragma unroll
                    for (int i = 0; i < NTP; ++i) acc[i] = mfma16(wreg[i][j], b, acc[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, IA>([&](auto ic) { cell(ic, acc[decltype(ic)::value], t); });
            }
#pragma unroll
            for (int i = IA; i < NTL; ++i) late[i - IA] = acc[i];
            if (HD) head_store(acc[NTL], t - 1);                   // (the head of h(t-1): the step's B operands)
            REC_STAMP(5);
            publish(integral_constant<int, 0>{}, integral_constant<int, IA>{}, hb_next, t);
            REC_STAMP(2);
            __syncthreads();                                       // A(t): tiles < IA of h(t) published
            REC_STAMP(3);
        };
        step(0, integral_constant<bool, true>{});
        for (int t = 1; t < a.T; ++t) step(t, integral_constant<bool, false>{});
        // the late tiles of the last step
        static_for<IA, NTL>([&](auto ic) { cell(ic, late[decltype(ic)::value - IA], a.T - 1); });
        publish(integral_constant<int, IA>{}, ntl_c, smem + (a.T & 1) * HB, a.T - 1);
        __syncthreads();
        if (HD) {
            // the head of the last step's h
            const float* hf = hbase + (a.T & 1) * HB;
            f32x4 hv = hbias;
#pragma unroll
            for (int j = 0; j < KS; ++j) hv = mfma16(wreg[NTL][j], hf[j * 64], hv);
            head_store(hv, a.T - 1);
        }
    };
    const integral_constant<bool, false> no{};
    const integral_constant<bool, true> yes{};
    if (wave < 4) {
        if (ntile == TPW) run(integral_constant<int, TPW>{}, no, no);
        else if (HD_BUILT && hd) run(integral_constant<int, TPW - 1>{}, no, integral_constant<bool, HD_BUILT>{});
        else run(integral_constant<int, TPW - 1>{}, no, no);
    } else {
        if (ntile == TPW) run(integral_constant<int, TPW>{}, yes, no);
        else run(integral_constant<int, TPW - 1>{}, yes, no);
    }
    if (a.state_out && row_ok) {
        float* st = a.state_out + ((size_t)row_raw * SNL + a.st_l0) * 2 * HP;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
            if (i < ntile) {
                st[hpos[i]] = h[i];
                st[HP + hpos[i]] = c[i];
            }
    }
    if (a.din.gate_out) {
        // bias + W_hh . h_final: the part of the next step's gates shared by all rows continuing from this state
        const float* hf = hbase + (a.T & 1) * HB;
        f32x4 g[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            // bias of (unit, gate): tile32 * 32 + hh * 16 + 4 * jj + gate, with gslot = 16 * tile32 + 4 * jj
            const int bi = (gslot[i] >> 4) * 32 + (l32[i] >> 5) * 16 + (gslot[i] & 15);
            g[i] = f32x4{a.b[0][bi], a.b[0][bi + 1], a.b[0][bi + 2], a.b[0][bi + 3]};
        }
        if (ntile == TPW) products(g, hf, integral_constant<int, TPW>{}, integral_constant<int, 0>{}, integral_constant<int, KS>{});
        else products(g, hf, integral_constant<int, TPW - 1>{}, integral_constant<int, 0>{}, integral_constant<int, KS>{});
        if (row_ok) {
            float* go = a.din.gate_out + ((size_t)row_raw * SNL + a.st_l0) * NT * 32;
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                if (i < ntile) {
                    const int bi = (gslot[i] >> 4) * 32 + (l32[i] >> 5) * 16 + (gslot[i] & 15);
                    *(float4*)(go + bi) = make_float4(g[i][0], g[i][1], g[i][2], g[i][3]);
                }
        }
    }
}

// ---------------------------------------------------------------------------
// lstm_rec4_kernel: lstm_rec16_kernel's job on 4-ROW tiles with v_mfma_f32_4x4x1_16B_f32, for launches with very few
// rows (a training batch of 64, a date range of a few hundred days): a recurrent step is a chain of dependent work
// on the one CU a tile sits on, so with CUs to spare the tile should be as small as the matrix instruction allows.
//   * the instruction multiplies 16 independent blocks of (4 x 1) . (1 x 4): block b = 4 gate rows of ONE hidden unit
//     (A lane 4 b + gate), B = h[k] of the tile's 4 days replicated in every block (lane 4 b + day), so that lane
//     (b, day) receives the four gates of unit 16 G + b for its day in its four D registers: the cell is lane-local;
//   * wave w owns unit group G = w (16 units, 64 gate rows); its W_hh rows (one register per k: 104 - 128) stay in
//     registers; h goes through LDS as [day][unit] (double-buffered, one barrier per step) and comes back as one
//     broadcast float4 per four k.
// Per step a workgroup issues 8 NT x NG / ... = 728 - 1024 two-pass products (1.5 - 2 k cycles per SIMD) against
// lstm_rec16_kernel's 5.8 - 8.2 k for 16 rows.  Same formats in and out as lstm_rec16_kernel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// R tiles per workgroup (R = 1 or 2: 4 or 8 rows), their product chains interleaved, W_hh shared in registers: R = 2
// serves launches of 257 - 512 tiles in one round of workgroups (1000 days: 0.97 ms against 1.12 with R = 1).  A step's
// time grows with R (four tiles per workgroup measured no better than lstm_rec16_kernel), so larger launches stay on
// that kernel.  Every tile's chain is the same ascending-k chain: results do not depend on R.
template <int NT, bool SAVE, int R>
__device__ __forceinline__ void lstm_rec4_body(const LstmSeqArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HP = 8 * NT, HPK = 4 * NT;
    constexpr int NG = (HP + 15) / 16;                             // unit groups (waves with work)
    constexpr int HPAD = 144;                                      // LDS floats per day: >= HP, the 4 days 16 banks apart
    static_assert(NG <= 8 && HP <= HPAD && HP % 8 == 0, "one unit group per wave");
    static_assert(R == 1 || R == 2, "tiles of one workgroup stay inside a 32-row tile");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile4 = (long)blockIdx.x * R;                       // first of this workgroup's tiles
    const int j = lane & 3, b = lane >> 2;                         // day of the tile / unit of the group (B, D); A: gate / unit
    const long rowtile = tile4 >> 3;                               // the 32-row tile of the other kernels' layouts
    const int m0 = 4 * (int)(tile4 & 7) + j;                       // this lane's row inside it (tile r: + 4 r)
    const int SNL = a.st_nl > 0 ? a.st_nl : 1;
    const int SL = (SAVE && a.sv_nl > 0) ? a.sv_nl : 1;
    const int L0 = (SAVE && a.sv_nl > 0) ? a.sv_l0 : 0;
    const bool dropping = SAVE && L0 + 1 < SL && a.drop.p > 0.0f;
    const DropArgs drop = dropping ? resolve_drop(a.drop) : a.drop;
    const bool active = wave < NG;                                 // (wave-uniform)
    const int G = active ? wave : NG - 1;
    const int u0 = 16 * G + b;
    const bool uok = active && u0 < HP;
    const int u = u0 < HP ? u0 : HP - 1;                           // (clamped: addresses stay inside the layouts)
    const int gslot = 16 * (u >> 3) + 4 * ((u & 7) >> 1);          // + gate: slot of the gate value
    const int l32 = (u & 1) * 32 + m0;                             // lane of the 32-row layouts (tile r: + 4 r)
    const int hslot = 4 * (u >> 3) + ((u & 7) >> 1);               // slot of h in tile-B
    const int hpos = 8 * (u >> 3) + 4 * (u & 1) + ((u & 7) >> 1);  // position in the state arrays
    long row[R];
    bool row_ok[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long rr = (tile4 + r) * 4 + j;
        row_ok[r] = rr < a.n_rows;
        row[r] = row_ok[r] ? rr : a.n_rows - 1;
    }

    // W_hh rows of (unit 16 G + b, gate lane & 3), one register per hidden input k, gathered from the blob's 32x32x2
    // fragment layout (vsv_layout.h)
    // (element (row n, input kk) sits at quad kk >> 3, tile n >> 5, lane (kk & 1) * 32 + (n & 31), component
    // (kk >> 1) & 3: eight consecutive k of one row are two float4s, the even and the odd k; kx is a multiple of 4)
    float wreg[HP];
    {
        const int n = 32 * (u >> 3) + 4 * (u & 7) + (lane & 3);
        const float4* wq = (const float4*)a.w[0] + (size_t)((a.kx >> 2) * NT + (n >> 5)) * 64 + (n & 31);
#pragma unroll
        for (int q = 0; q < HP / 8; ++q) {
            const float4 we = wq[(size_t)q * NT * 64], wo = wq[(size_t)q * NT * 64 + 32];
            wreg[8 * q + 0] = uok ? we.x : 0.0f; wreg[8 * q + 2] = uok ? we.y : 0.0f;
            wreg[8 * q + 4] = uok ? we.z : 0.0f; wreg[8 * q + 6] = uok ? we.w : 0.0f;
            wreg[8 * q + 1] = uok ? wo.x : 0.0f; wreg[8 * q + 3] = uok ? wo.y : 0.0f;
            wreg[8 * q + 5] = uok ? wo.z : 0.0f; wreg[8 * q + 7] = uok ? wo.w : 0.0f;
        }
    }
    // Per-step traffic is buffer-addressed as in lstm_rec16_kernel: descriptor base (this row tile's part of the array)
    // + a constant lane offset + a scalar step offset; stores are unconditional -- a lane without a unit, a launch
    // without an h sequence and a step before t_out0 put the lane offset out of the descriptor's range and the hardware
    // drops the store -- so that the wait for the next step's pre-activations does not also wait for the stores.
    const bool from_t = !SAVE && a.pre_t != nullptr;               // wave-uniform
    const float* pbase = SAVE ? a.sv_gates + ((size_t)rowtile * a.T * SL + L0) * (16 * NT) * 64
                       : from_t ? a.pre_t : a.pre_g + (size_t)rowtile * a.T * (16 * NT) * 64;
    const wrsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pbase), 0, 0x7fffffff, 0x00020000);
    const unsigned pstep = SAVE ? (unsigned)SL * (16 * NT) * 256u : from_t ? (unsigned)a.pre_tstride * 4u : (16 * NT) * 256u;
    const unsigned pgate = from_t ? 8u * (unsigned)a.pre_stride : 256u;
    const unsigned nouse = uok ? 0u : 0x80000000u;                 // (stores of a lane without a unit)
    unsigned pvo[R], hvo[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        pvo[r] = from_t ? 4u * ((unsigned)(2 * gslot + (l32 >> 5)) * (unsigned)a.pre_stride + (unsigned)row[r])
                        : 4u * (unsigned)(gslot * 64 + l32 + 4 * r);
        hvo[r] = 4u * (unsigned)(hslot * 64 + l32 + 4 * r) | nouse;
    }
    const int t_outs = a.T - a.t_out0;
    const wrsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(a.hout ? a.hout + (size_t)rowtile * t_outs * HPK * 64 : a.hout, 0,
                                                          a.hout ? 0x7fffffff : 0, 0x00020000);
    const wrsrc_t svc = __builtin_amdgcn_make_buffer_rsrc(SAVE ? a.sv_c + ((size_t)rowtile * a.T * SL + L0) * HPK * 64 : nullptr, 0,
                                                          SAVE ? 0x7fffffff : 0, 0x00020000);
    const wrsrc_t svh = __builtin_amdgcn_make_buffer_rsrc(SAVE ? a.sv_h + ((size_t)rowtile * a.T * SL + L0) * HPK * 64 : nullptr, 0,
                                                          SAVE ? 0x7fffffff : 0, 0x00020000);
    const wrsrc_t svd = __builtin_amdgcn_make_buffer_rsrc(dropping ? a.sv_hd + (size_t)rowtile * a.T * HPK * 64 : nullptr, 0,
                                                          dropping ? 0x7fffffff : 0, 0x00020000);
    auto load_pre = [&](int r, int t) -> f32x4 {
        const unsigned so = (unsigned)t * pstep;
        return f32x4{bload1(prs, pvo[r], so), bload1(prs, pvo[r], so + pgate), bload1(prs, pvo[r], so + 2 * pgate),
                     bload1(prs, pvo[r], so + 3 * pgate)};
    };
    f32x4 pre[R];
    float c[R], h[R];
    const bool has_state = a.din.state_in != nullptr;              // wave-uniform
    float* const hcol = smem + j * HPAD;                           // this lane's day: tile r, buffer p at (p * R + r) * 4 HPAD
    const int uw = uok ? u0 : HPAD - 1;                            // where this lane's h goes (no unit: a pad float nothing reads)
    static_assert(HP < HPAD, "a pad float per day for the lanes without a unit");
#pragma unroll
    for (int r = 0; r < R; ++r) {
        pre[r] = load_pre(r, 0);
        c[r] = 0.0f;
        h[r] = 0.0f;
        if (has_state) {
            const long srow = a.din.state_div == 1 ? row[r] : row[r] / a.din.state_div;
            const float* ps = a.din.state_in + ((size_t)srow * SNL + a.st_l0) * 2 * HP + hpos;
            h[r] = ps[0];
            c[r] = ps[HP];
        }
        hcol[(size_t)r * (4 * HPAD) + uw] = h[r];
    }
    __syncthreads();
    // per tile ONE chain in ascending k: the matrix pipe accumulates a product at a time in k order whatever the
    // tile shape, so this is bit for bit what lstm_kernel / lstm_rec16_kernel compute; the R chains (and the two
    // waves of a SIMD) fill each other's issue gaps
    auto products = [&](f32x4 (&acc)[R], const float* hr) {
        if (SAVE) {
            // training: FOUR chains per tile (k mod 4), summed at the end -- the single chain is HP dependent
            // products at the instruction's 28-cycle dependent-issue latency (3.6 k of a step's 4.5 k cycles,
            // profiles/r3_rec_step_stamps.txt), four are issue-bound at 8.  The sum's order differs from the
            // generation kernels' (which stay on one ascending-k chain, bit-equal to each other): a training forward
            // agrees with them to rounding, as its backward (lstm_bwd4_kernel, the same split) always has.
            f32x4 p1[R], p2[R], p3[R];
#pragma unroll
            for (int r = 0; r < R; ++r) p1[r] = p2[r] = p3[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < HP; k += 4) {
                float4 hv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) hv[r] = *(const float4*)(hr + (size_t)r * (4 * HPAD) + k);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r] = mfma4(wreg[k + 0], hv[r].x, acc[r]);
                    p1[r] = mfma4(wreg[k + 1], hv[r].y, p1[r]);
                    p2[r] = mfma4(wreg[k + 2], hv[r].z, p2[r]);
                    p3[r] = mfma4(wreg[k + 3], hv[r].w, p3[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[r][v] += (p1[r][v] + p2[r][v]) + p3[r][v];
            return;
        }
#pragma unroll
        for (int k = 0; k < HP; k += 4) {
            float4 hv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) hv[r] = *(const float4*)(hr + (size_t)r * (4 * HPAD) + k);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = mfma4(wreg[k + 0], hv[r].x, acc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = mfma4(wreg[k + 1], hv[r].y, acc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = mfma4(wreg[k + 2], hv[r].z, acc[r]);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = mfma4(wreg[k + 3], hv[r].w, acc[r]);
        }
    };
    for (int t = 0; t < a.T; ++t) {
        const float* hr = hcol + (size_t)((t & 1) * R) * (4 * HPAD);
        float* hw = hcol + (size_t)(((t + 1) & 1) * R) * (4 * HPAD);
        f32x4 acc[R];
        REC_STAMP(0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = pre[r];
#pragma unroll
        for (int r = 0; r < R; ++r) pre[r] = load_pre(r, t + 1 < a.T ? t + 1 : t);   // lands while the products run
        if ((t > 0 || has_state) && active) products(acc, hr);
        REC_STAMP(1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float ig = fast_sigmoid(acc[r][0]);
            const float fg = fast_sigmoid(acc[r][1]);
            const float gg = fast_tanh(acc[r][2]);
            const float og = fast_sigmoid(acc[r][3]);
            c[r] = __builtin_fmaf(fg, c[r], ig * gg);
            h[r] = og * fast_tanh(c[r]);
            hw[(size_t)r * (4 * HPAD) + uw] = h[r];
            if (SAVE) {
                // activated gates, cell and hidden state of the step, where the backward kernels read them (sv_gates:
                // the slots the pre-activations were read from)
                const unsigned so = (unsigned)t * pstep, po = pvo[r] | nouse;
                bstore1(prs, ig, po, so);
                bstore1(prs, fg, po, so + 256u);
                bstore1(prs, gg, po, so + 512u);
                bstore1(prs, og, po, so + 768u);
                const unsigned sc = (unsigned)t * (unsigned)(SL * HPK * 256);
                bstore1(svc, c[r], hvo[r], sc);
                bstore1(svh, h[r], hvo[r], sc);
                if (dropping) {
                    float fct[4];
                    dropout_factors(drop, (uint32_t)row[r], (uint32_t)t, u >> 3, u & 1, L0, fct);
                    bstore1(svd, h[r] * fct[(u & 7) >> 1], hvo[r], (unsigned)t * (unsigned)(HPK * 256));
                }
            }
            // (padding rows too: the next stage reads whole tiles)
            bstore1(hrs, h[r], hvo[r] | (t >= a.t_out0 ? 0u : 0x80000000u), (unsigned)(t - a.t_out0) * (unsigned)(HPK * 256));
        }
        REC_STAMP(2);
        __syncthreads();                                           // h(t) published (the other buffer is free again)
        REC_STAMP(3);
    }
    if (a.state_out && uok) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row_ok[r]) {
                float* st = a.state_out + ((size_t)row[r] * SNL + a.st_l0) * 2 * HP;
                st[hpos] = h[r];
                st[HP + hpos] = c[r];
            }
    }
    if (a.din.gate_out) {
        // bias + W_hh . h_final: the part of the next step's gates shared by all rows continuing from this state
        const float* hf = hcol + (size_t)((a.T & 1) * R) * (4 * HPAD);
        const int bi = (gslot >> 4) * 32 + (l32 >> 5) * 16 + (gslot & 15);
        f32x4 g[R];
#pragma unroll
        for (int r = 0; r < R; ++r) g[r] = f32x4{a.b[0][bi], a.b[0][bi + 1], a.b[0][bi + 2], a.b[0][bi + 3]};
        if (active) products(g, hf);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row_ok[r] && uok)
                *(float4*)(a.din.gate_out + ((size_t)row[r] * SNL + a.st_l0) * NT * 32 + bi) = make_float4(g[r][0], g[r][1], g[r][2], g[r][3]);
    }
}
template <int NT, bool SAVE, int R>
__global__ void __launch_bounds__(512) lstm_rec4_kernel(const LstmSeqArgs a) {
    lstm_rec4_body<NT, SAVE, R>(a);
}
// Two independent launches of the above as ONE (blockIdx.y picks the argument set): the two encoders of a training
// step are independent stacks of sixteen 4-row tiles each at batch 64 -- a launch of its own per stack leaves the
// chip to one of them at a time, and running them on two streams costs a cross-queue fork and join (~25 us, and the
// side queue picks its first packet up 60 - 90 us late: profiles/r3_f_train_timeline_graph.txt).
template <int NT, bool SAVE, int R>
__global__ void __launch_bounds__(512) lstm_rec4_pair_kernel(const LstmSeqArgs a0, const LstmSeqArgs a1) {
    if (blockIdx.y == 0) {
        if ((long)blockIdx.x * R < 8 * a0.n_rowtiles) lstm_rec4_body<NT, SAVE, R>(a0);
    } else {
        if ((long)blockIdx.x * R < 8 * a1.n_rowtiles) lstm_rec4_body<NT, SAVE, R>(a1);
    }
}

// The decoder memory's input [ctx_pad | z] of every (row, step) in the tile-B layout the training kernels save
// (sv_x: [inst = rowtile*T + t][kx slots][64], slot s of lane hh*32 + r = feature 2s + hh of row 32*rowtile + r),
// with the reparameterised z written to z_out: what lstm_kernel<.., XMODE 1> assembles on its time loop, as one
// launch of its own, so that the decoder's first layer can run as a pre-activation GEMM (lstm_pre_kernel) and the
// register-resident recurrent launch (lstm_rec_kernel) like every other layer of a training step.
__global__ void __launch_bounds__(256) decoder_input_kernel(const LstmSeqArgs a) {
    const long total = a.n_rowtiles * a.T * a.kx * 64;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const int s = (int)((i >> 6) % a.kx);
    const long inst = (i >> 6) / a.kx;
    const int t = (int)(inst % a.T);
    const long rowtile = inst / a.T;
    const long row_raw = rowtile * 32 + (lane & 31);
    const bool row_ok = row_raw < a.n_rows;
    const long row = row_ok ? row_raw : a.n_rows - 1;
    const vsv_decode_in& d = a.din;
    DecIn di = {};
    const long day_rel = row / d.S;
    if (d.ctx) di.ctx = d.ctx + (row / d.ctx_div) * d.ctx_rs;
    if (d.zb) di.z_idx = (row / d.z_div) * d.z_rs;
    di.eps = d.eps_in ? d.eps_in + (size_t)row * d.eps_T * a.L : nullptr;
    di.sample = (uint32_t)(row - day_rel * d.S);
    di.day = (uint32_t)d.day0 + (uint32_t)day_rel;
    a.sv_x[i] = decoder_input(a, di, row, row_ok, true, t, 2 * s + (lane >> 5));
}

// layer[li] of the decoder head (LstmSeqArgs::head, part of the by-value kernel argument) for a run-time li,
// read from the kernarg segment by scalar loads: indexing the by-value copy itself would put it in scratch.
// The struct is the kernel's only explicit argument, so it sits at offset 0 of the segment.
__device__ __forceinline__ DenseLayerArg chain_layer_of(int li) {
    typedef const __attribute__((address_space(4))) LstmSeqArgs* KP;
    const KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    DenseLayerArg r;
    r.w = kp->head.layer[li].w;
    r.b = kp->head.layer[li].b;
    r.nq = kp->head.layer[li].nq;
    r.nt = kp->head.layer[li].nt;
    r.n_relu = kp->head.layer[li].n_relu;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        r.q0[t] = kp->head.layer[li].q0[t];
        r.nqb[t] = kp->head.layer[li].nqb[t];
    }
    return r;
}

// ---------------------------------------------------------------------------
// mc_decode1_kernel: the Monte-Carlo one-day decode (prior draw from a day's shared prefix state, T = 1):
// the whole decoder step of 32 rows -- Philox latents, both LSTM layers, the four head layers, the
// sample-minor store -- runs in ONE WAVE, with no workgroup barrier and no other wave to wait for.
// lstm_kernel<.., HEAD> splits a row tile's gate tiles over four waves that meet at a barrier twice per
// layer and once per head layer; its waves spend 3/4 of their lifetime waiting (prologue latency, cell
// math, barriers, the widest head band), which two co-resident workgroups cannot cover.  Here a wave walks
// the 16 gate tiles of a layer in four passes of four tiles (64 accumulator registers), keeps the layer's
// h in registers until the last pass has read the layer's input, and owns one 64-slot LDS column that
// every layer updates in place.  Same products in the same k order as lstm_kernel: results are bit-equal.
// Eight such waves per CU (two 4-wave workgroups, 64 KB LDS each) overlap freely.
// ---------------------------------------------------------------------------
#ifndef VSV_MC1_STREAM
#define VSV_MC1_STREAM gemm_stream_buf2
#endif
#ifndef VSV_MC1_HEAD_REG
#define VSV_MC1_HEAD_REG 1
#endif
// (the register-resident head, defined with dec_seq_kernel below)
template <int NT>
__device__ __forceinline__ void dec_seq_head(const LstmSeqArgs& a, float (&act)[16 * kHeadTiles], const int lane,
                                             const long rowtile, const int to, const uint32_t day_rel, const uint32_t sample,
                                             float4 (&fa)[4 * kHeadTiles], float (&bf)[4], const int dsq_t, const int wave);
template <int NT>
__global__ void __launch_bounds__(256, 2) mc_decode1_kernel(const LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(NT == 16, "built for 16 gate tiles (decoder hidden 121..128)");
    constexpr int HP = 8 * NT;
    constexpr int NP = NT / 4;                                    // passes of four gate tiles
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long rowtile = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (rowtile >= a.n_rowtiles) return;                          // (no barriers below: a wave may leave alone)
    float* const col0 = smem + (size_t)wave * (64 * 64);          // this wave's activation column, without the lane
    const vsv_decode_in& d = a.din;
    const int kx = a.kx;
    const DenseChainArgs& hd = a.head;
    MC1_STAMP(0);

    // ---- per-row constants, latents
    uint32_t state_row, day_rel, sample;
    {
        const int hh = lane >> 5, m = lane & 31;
        const long row_raw = rowtile * 32 + m;
        const uint32_t row32 = (uint32_t)(row_raw < a.n_rows ? row_raw : a.n_rows - 1);
        const uint32_t S32 = (uint32_t)(d.S > 0x7fffffffL ? 0x7fffffffL : d.S);
        day_rel = row32 / S32;
        sample = row32 - day_rel * S32;
        const uint32_t day = (uint32_t)d.day0 + day_rel;
        state_row = (d.state_div == d.S) ? day_rel : row32 / (uint32_t)(d.state_div > 0x7fffffffL ? 0x7fffffffL : d.state_div);
        float* const col = col0 + lane;
        // x part of layer 0: [ctx (zero padding) | z | alignment padding]; k-step s, half hh <-> feature 2s + hh
        for (int s = 0; s < kx; ++s) col[s * 64] = (2 * s + hh >= a.Lc && 2 * s + hh < a.Lc + a.L) ? d.mu0 : 0.0f;
        if (d.std0 != 0.0f) {
            // one Philox block gives the latents 4b .. 4b+3 (philox_normal_elem: word pair f & 2, cos / sin by f & 1);
            // this lane holds the two of them with its parity
            const int par = (hh - a.Lc) & 1;
            const uint32_t k0 = (uint32_t)d.seed, k1 = (uint32_t)(d.seed >> 32);
            for (int b = 0; 4 * b < a.L; ++b) {
                uint32_t c0 = sample, c1 = day, c2 = (uint32_t)(d.step0 - d.eps_t0), c3 = (uint32_t)b;
                philox4x32_10(k0, k1, c0, c1, c2, c3);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int f = 4 * b + 2 * p + par;
                    const uint32_t xa = p ? c2 : c0, xb = p ? c3 : c1;
                    const float u1 = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-07f;
                    const float u2 = ((float)(xb >> 9) + 0.5f) * 1.1920928955078125e-07f;
                    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
                    const float e = r * (par ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
                    if (f < a.L) col[((f + a.Lc - hh) >> 1) * 64] = d.mu0 + d.std0 * e;
                }
            }
        }
    }
    const float* const gate_in = d.gate_in + (size_t)state_row * 2 * NT * 32;
    const float* const st = d.state_in + (size_t)state_row * 2 * 2 * HP;
    // of layer 0's input quads only those that hold z are multiplied (the context half is the zero padding)
    const int qx0 = a.Lc >> 3, qxn = ((a.Lc + a.L + 7) >> 3) - qx0;
    MC1_STAMP(1);

    // ---- the two memory layers: eight passes of four gate tiles.  A pass's accumulators start from the day's
    // bias + W_hh . h_prefix (gate_in) and its cell state from the day's state: both are requested one pass AHEAD,
    // tile by tile, as the cell math of the pass before frees the tile's registers -- a pass that began with these
    // loads waited an L2 round trip for them with nothing to multiply (layer 0's passes are 24 products long).
    // Layer 0 multiplies only the k-pairs that hold z (kmask): the context half of its input is the zero padding.
    unsigned kmask = 0;
    for (int j = 0; j < 4 * qxn && j < 32; ++j) {
        const int f = 8 * qx0 + 2 * j;
        if (f + 1 >= a.Lc && f < a.Lc + a.L) kmask |= 1u << j;
    }
    f32x16 acc[4];
    float c[4][4];
    float4 a0[4];
    auto fetch_tile = [&](int l, int tile, f32x16& ac, float (&cc)[4], int hh) {
        const float4* gp = (const float4*)(gate_in + (size_t)l * NT * 32 + tile * 32 + hh * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 x = gp[v];
            ac[4 * v + 0] = x.x; ac[4 * v + 1] = x.y; ac[4 * v + 2] = x.z; ac[4 * v + 3] = x.w;
        }
        const float4 cv = *(const float4*)(st + (size_t)l * 2 * HP + HP + 8 * tile + 4 * hh);
        cc[0] = cv.x; cc[1] = cv.y; cc[2] = cv.z; cc[3] = cv.w;
    };
    auto w_of = [&](int l) { return (const float4*)a.w[l] + (l == 0 ? (size_t)qx0 * NT * 64 : 0); };
    {
        const int lv = opaque(lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) fetch_tile(0, i, acc[i], c[i], lv >> 5);
        wload_buf<4>(a0, make_wrsrc(w_of(0)), (unsigned)lv * 16u, 0u);
    }
    float hold[NT][4];                                            // a layer's h, unit 8*tile + 2*jj + hh
#if VSV_MC1_HEAD_REG
    float4 fa[4 * kHeadTiles];                                    // the head's fragment buffer (dec_seq_head)
    float hbf[4];
#endif
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int lv = opaque(lane);
            const int hh = lv >> 5;
            const wrsrc_t wr = make_wrsrc(w_of(l));
            const unsigned voff = (unsigned)lv * 16u, sb = (unsigned)p * 4096u;      // tiles 4p .. 4p + 3
            if (l == 0) {
                const float* const bcol = col0 + lv + (size_t)(4 * qx0) * 64;
#pragma unroll 1
                for (int q = 0; q < qxn; ++q) {
                    float4 a1[4];
                    const int qn = (q + 1 < qxn) ? q + 1 : q;
                    wload_buf<4>(a1, wr, voff, sb + (unsigned)qn * (NT * 1024u));
                    float bv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = bcol[(4 * q + j) * 64];
                    const unsigned km = kmask >> (4 * q);
                    if (km & 1u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = mfma32(a0[i].x, bv[0], acc[i]);
                    }
                    if (km & 2u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = mfma32(a0[i].y, bv[1], acc[i]);
                    }
                    if (km & 4u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = mfma32(a0[i].z, bv[2], acc[i]);
                    }
                    if (km & 8u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = mfma32(a0[i].w, bv[3], acc[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) a0[i] = a1[i];
                }
            } else {
                const float* const bcol = col0 + lv;
                VSV_MC1_STREAM<4>(acc, a0, wr, voff, sb, NT * 1024u, NT, [&](int q, float (&bv)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = bcol[(4 * q + j) * 64];
                });
            }
            MC1_STAMP(2 * (l * NP + p) + 2);
#if VSV_MC1_HEAD_REG
            if (l == 1 && p == NP - 1) {
                // the head's first fragments and bias, in front of the memory's last cell
                const DenseLayerArg& ly = hd.layer[0];
                band_load_all<4 * kHeadTiles>(fa, make_wrsrc(ly.w), voff, 0u, (unsigned)ly.nt * 1024u, 0, 4 * kHeadTiles);
#pragma unroll
                for (int i = 0; i < 4; ++i) hbf[i] = bias_frag(ly.b, i, lv);
            }
#endif
            // cell math tile by tile; a tile's registers then take the NEXT pass's starting values
            const bool more = !(l == 1 && p == NP - 1);
            const int nl = (p == NP - 1) ? l + 1 : l, np = (p + 1) % NP;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float hn[1][4];
                lstm_cell_pk<1>(reinterpret_cast<f32x16(&)[1]>(acc[i]), reinterpret_cast<float(&)[1][4]>(c[i]), hn);
                // (pinned: left to itself the compiler sinks the cell math of all four passes below the last
                // pass's GEMM and parks the earlier passes' accumulators in scratch meanwhile)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    asm volatile("" : "+v"(hn[0][jj]));
                    hold[4 * p + i][jj] = hn[0][jj];
                }
                if (more) fetch_tile(nl, 4 * np + i, acc[i], c[i], hh);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) wload_buf<4>(a0, make_wrsrc(w_of(nl)), voff, (unsigned)np * 4096u);
            __builtin_amdgcn_sched_barrier(0);
            MC1_STAMP(2 * (l * NP + p) + 3);
        }
        // every pass has read the layer's input: the column now takes the layer's h (natural k-step order)
        if (l == 0 || !VSV_MC1_HEAD_REG) {
            float* const colw = col0 + opaque(lane);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) colw[(4 * t + jj) * 64] = hold[t][jj];
        }
    }

#if VSV_MC1_HEAD_REG
    // ---- decoder head, from registers: its input is the top layer's h as it stands (k-step 4 t + jj <-> hold[t][jj]),
    // a layer's accumulators are the next layer's B operands (dec_seq_head)
    {
        float (&act)[16 * kHeadTiles] = reinterpret_cast<float(&)[16 * kHeadTiles]>(hold);
        dec_seq_head<NT>(a, act, lane, rowtile, 0, day_rel, sample, fa, hbf, 1, wave);       // (1: the head's phase stamps, diagnostic builds)
    }
#else
    // ---- decoder head: every layer reads the column and overwrites it with its (ReLU'd) outputs
    const int n_layers = hd.n_layers;                             // 4: [Linear . ConvT] ConvT ConvT Conv
    auto bias_init = [&](f32x16 (&acc)[4], const DenseLayerArg& ly, int hh) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ly.nt) {
                const float4* bp = (const float4*)(ly.b + t * 32 + hh * 16);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 x = bp[v];
                    acc[t][4 * v + 0] = x.x; acc[t][4 * v + 1] = x.y; acc[t][4 * v + 2] = x.z; acc[t][4 * v + 3] = x.w;
                }
            }
        }
    };
    // n_relu, out_f and the aux range are scalars: a tile that lies wholly inside (or outside) them takes a
    // path without per-value compares
    auto epilogue = [&](f32x16 (&acc)[4], const DenseLayerArg& ly, int li, int lv) {
        const bool last = (li == n_layers - 1);
        const int hh = lv >> 5;
        const bool do_aux = (li == hd.aux_layer) && hd.aux != nullptr;
        const long rr = rowtile * 32 + (lv & 31);
        const bool row_ok = rr < hd.n_rows;
        float* const colw = col0 + lv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ly.nt) {
                const int f_lo = 32 * t;
                if (ly.n_relu >= f_lo + 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = relu1(acc[t][r]);
                } else if (ly.n_relu > f_lo) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (feat_of(t, r, hh) < ly.n_relu) acc[t][r] = relu1(acc[t][r]);
                }
                if (!last) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) colw[(16 * t + r) * 64] = acc[t][r];
                }
                if (do_aux && row_ok && f_lo + 32 > hd.aux_f0 && f_lo < hd.aux_f0 + hd.aux_n) {
                    float* const ax = hd.aux + (size_t)rr * hd.aux_n - hd.aux_f0;       // (T = 1)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = feat_of(t, r, hh);
                        if (f >= hd.aux_f0 && f < hd.aux_f0 + hd.aux_n) ax[f] = acc[t][r];
                    }
                }
                if (last && hd.out && row_ok && f_lo < hd.out_f) {
                    // feature f of this row lives at po0 + f * fs: sample-minor (day, f, sample) or row-major (row, f)
                    const bool smin = hd.out_mode == 2;
                    const long fs = smin ? hd.S : 1;
                    float* po = smin ? hd.out + (size_t)day_rel * hd.out_f * hd.S + sample : hd.out + (size_t)rr * hd.out_f;
                    po += (size_t)(f_lo + 4 * hh) * fs;
                    if (f_lo + 32 <= hd.out_f) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) po[(size_t)(8 * (r >> 2) + (r & 3)) * fs] = acc[t][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (feat_of(t, r, hh) < hd.out_f) po[(size_t)(8 * (r >> 2) + (r & 3)) * fs] = acc[t][r];
                    }
                }
            }
        }
    };
    {
        // first layer ([Linear . ConvT] folded): full K, four tiles, one pinned stream
        const DenseLayerArg& ly = hd.layer[0];
        const int lv = opaque(lane);
        const float* const bcol = col0 + lv;
        f32x16 acc[4];
        bias_init(acc, ly, lv >> 5);
        float4 a0[4];
        const wrsrc_t wr = make_wrsrc(ly.w);
        const unsigned voff = (unsigned)lv * 16u;
        wload_buf<4>(a0, wr, voff, 0u);
        gemm_stream_buf<4>(acc, a0, wr, voff, 0u, 4 * 1024u, ly.nq, [&](int q, float (&bv)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = bcol[(4 * q + j) * 64];
        });
        MC1_STAMP(18);
        epilogue(acc, ly, 0, lv);
        MC1_STAMP(19);
    }
    // conv layers in pixel-major order: tile t reads the K quads [q0[t], q0[t] + nqb[t]) only.  One copy of
    // the code for all of them (the kernel's instruction footprint matters: eight waves per CU walk it out of
    // phase with each other)
#pragma unroll 1
    for (int li = 1; li < n_layers; ++li) {
        const DenseLayerArg ly = chain_layer_of(li);
        const int lv = opaque(lane);
        const float* const bcol = col0 + lv;
        f32x16 acc[4];
        bias_init(acc, ly, lv >> 5);
        const wrsrc_t wr = make_wrsrc(ly.w);
        const unsigned voff = (unsigned)lv * 16u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ly.nt) {
                const int q0 = pick4(ly.q0, t), nqb = pick4(ly.nqb, t);
                tile_band_all_buf<4 * kHeadTiles>(acc[t], wr, voff, t * 1024u, (unsigned)ly.nt * 1024u, bcol, q0, nqb);
            }
        }
        MC1_STAMP(18 + 2 * li);
        epilogue(acc, ly, li, lv);
        MC1_STAMP(19 + 2 * li);
    }
#endif
}

// ---------------------------------------------------------------------------
// dec_seq_kernel (round 3): the decoder over T steps for MANY rows -- horizon > 1 prior draws, the oracle
// (posterior) decode of all C + H positions, the C + 1 steps of an autoregressive rollout step -- on
// mc_decode1_kernel's schedule: ONE WAVE per 32-row tile runs both memory layers (16 gate tiles in four passes
// of four), the step's input assembly (context embedding, z = base + scale * Philox / injected eps) and the
// head, with no workgroup barrier.  What a sequence adds to the one-day draw is state: the cell state of both
// layers (2 x 16 x 4 = 128 registers per lane) stays in REGISTERS for the whole time loop, so the kernel runs
// one wave per SIMD on the 512-register budget (the fp32 matrix and vector instructions of a SIMD's two waves
// serialise anyway, DESIGN.md "The bound that is left"); h of both layers lives in the wave's own LDS columns
// (x: kx slots, h0 and h1: 64 slots each; 35 KB per wave at kx = 12), a layer's new h is held in registers
// until the last pass has read the old one, and the head runs from those registers without touching LDS (a
// layer's accumulators are the next layer's B operands as they stand).  Same products in the same k order as
// lstm_kernel: results are bit-equal.
// ---------------------------------------------------------------------------
// One banded head tile of dec_seq_kernel: the chain of its K quads on one accumulator, the B operands taken straight
// from REGISTERS (the D layout of a layer is the B layout of the next: k-step 16 t + r of a layer's input is register r
// of the previous layer's tile t), indexed by the ABSOLUTE quad, so that the index is static; the band is a bit mask
// (bit q set: quad q is multiplied) tested by scalar guards.  The NEXT tile's fragments are requested quad by quad from
// inside the chain, each into the slot whose quad has just been multiplied (a fetch issues in the shadow of the product
// before it -- in front of the chain sixteen fetches cost ~1000 cycles of a lone wave's time -- and has the rest of
// this tile's band to land), unconditionally, all sixteen quads of its K range.
template <int NQ>
__device__ __forceinline__ void band_tile_reg(f32x16& acc, float4 (&f)[NQ], wrsrc_t rn, unsigned voff, unsigned sbn, unsigned qsbn,
                                              const float (&b)[4 * NQ], unsigned qmask) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if ((qmask >> q) & 1u) {
            acc = mfma32(f[q].x, b[4 * q + 0], acc);
            acc = mfma32(f[q].y, b[4 * q + 1], acc);
            acc = mfma32(f[q].z, b[4 * q + 2], acc);
            acc = mfma32(f[q].w, b[4 * q + 3], acc);
        }
        f[q] = bload(rn, voff, sbn + (unsigned)q * qsbn);      // the next tile's quad q takes the slot
    }
}

// The decoder head of dec_seq_kernel, entirely in registers: act holds the head's input on entry (the top layer's new
// h, k-step order) and every layer's activations in turn; nothing goes through LDS, so the h columns stay as the
// memory left them.  All layers, the full-K first one included, run as banded tiles through ONE rolled loop (the
// kernel's instruction footprint is what eight out-of-phase waves share an instruction cache for).  fa: the fragment
// buffer; on entry it holds layer 0's tile 0 and bf layer 0's bias fragments (dec_seq_kernel requests them behind the
// memory's last cell, in front of the column write).
template <int NT>
__device__ __forceinline__ void dec_seq_head(const LstmSeqArgs& a, float (&act)[16 * kHeadTiles], const int lane,
                                             const long rowtile, const int to, const uint32_t day_rel, const uint32_t sample,
                                             float4 (&fa)[4 * kHeadTiles], float (&bf)[4], const int dsq_t, const int wave) {
    constexpr int NQB = 4 * kHeadTiles;                           // K quads of a head layer (16)
    const DenseChainArgs& hd = a.head;
    const int n_layers = hd.n_layers;                             // 4: [Linear . ConvT] ConvT ConvT Conv
#pragma unroll 1
    for (int li = 0; li < n_layers; ++li) {
        const DenseLayerArg ly = chain_layer_of(li);
        const DenseLayerArg lyn = chain_layer_of(li + 1 < n_layers ? li + 1 : li);
        const int lv = opaque(lane);
        const int hh = lv >> 5;
        const wrsrc_t wr = make_wrsrc(ly.w), wrn = make_wrsrc(lyn.w);
        const unsigned voff = (unsigned)lv * 16u;
        float bfn[4];                                             // the next layer's bias fragments
#pragma unroll
        for (int i = 0; i < 4; ++i) bfn[i] = bias_frag(lyn.b, i < lyn.nt ? i : lyn.nt - 1, lv);
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ly.nt) {
                // what follows this tile: the layer's next tile, or the next layer's first
                const bool wrap = t + 1 >= ly.nt;
                const unsigned sbn = wrap ? 0u : (unsigned)(t + 1) * 1024u, qsbn = (unsigned)(wrap ? lyn.nt : ly.nt) * 1024u;
                const unsigned qmask = ((1u << pick4(ly.nqb, t)) - 1u) << pick4(ly.q0, t);
                acc[t] = bias_acc(bf[t], lv < 32 ? 1.0f : 0.0f);
                band_tile_reg<NQB>(acc[t], fa, wrap ? wrn : wr, voff, sbn, qsbn, act, qmask);
            }
        }
        DSQ_STAMP(18 + 2 * li);
        // ---- epilogue: ReLU; the activations take the layer's outputs; ex-features / surfaces leave from the layers
        // that produce them
        const bool last = (li == n_layers - 1);
        const bool do_aux = (li == hd.aux_layer) && hd.aux != nullptr;
        const long rr = rowtile * 32 + (lv & 31);
        const bool row_ok = rr < hd.n_rows;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < ly.nt) {
                const int f_lo = 32 * t;
                float v[16];
                if (ly.n_relu >= f_lo + 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = relu1(acc[t][r]);
                } else if (ly.n_relu > f_lo) {
                    // register r holds feature f_lo + 4 hh + 8 (r >> 2) + (r & 3): max(x, 0) below n_relu, max(x, -inf)
                    // from there on (a select and a max per value, no branches)
                    const int thr = ly.n_relu - f_lo - 4 * hh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float lo = (8 * (r >> 2) + (r & 3) < thr) ? 0.0f : -__builtin_inff();
                        float y;
                        asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(lo), "v"(acc[t][r]));
                        v[r] = y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[t][r];
                }
                if (do_aux && row_ok && f_lo + 32 > hd.aux_f0 && f_lo < hd.aux_f0 + hd.aux_n) {
                    float* const ax = hd.aux + ((size_t)rr * hd.T + to) * hd.aux_n - hd.aux_f0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int f = feat_of(t, r, hh);
                        if (f >= hd.aux_f0 && f < hd.aux_f0 + hd.aux_n) ax[f] = v[r];
                    }
                }
                if (last && hd.out && row_ok && f_lo < hd.out_f) {
                    // feature f of this (row, step) lives at po0 + f * fs: sample-minor (day, step, f, sample) or
                    // row-major (row, step, f)
                    const bool smin = hd.out_mode == 2;
                    const long fs = smin ? hd.S : 1;
                    float* po = smin ? hd.out + ((size_t)day_rel * hd.T + to) * hd.out_f * hd.S + sample
                                     : hd.out + ((size_t)rr * hd.T + to) * hd.out_f;
                    po += (size_t)(f_lo + 4 * hh) * fs;
                    if (f_lo + 32 <= hd.out_f) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) po[(size_t)(8 * (r >> 2) + (r & 3)) * fs] = v[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (feat_of(t, r, hh) < hd.out_f) po[(size_t)(8 * (r >> 2) + (r & 3)) * fs] = v[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) act[16 * t + r] = v[r];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = bfn[i];
        DSQ_STAMP(19 + 2 * li);
    }
}

#ifndef VSV_DSEQ_STREAM
#define VSV_DSEQ_STREAM gemm_stream_buf2
#endif
template <int NT>
__global__ void __launch_bounds__(256, 1) dec_seq_kernel(const LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(NT == 16, "built for 16 gate tiles (decoder hidden 121..128)");
    constexpr int HP = 8 * NT, HPK = 4 * NT, NP = NT / 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long rowtile = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (rowtile >= a.n_rowtiles) return;                          // (no barriers below: a wave may leave alone)
    const vsv_decode_in& d = a.din;
    const int kx = a.kx, L = a.L, Lc = a.Lc, T = a.T;
    // this wave's columns, without the lane: [x | h of layer 0 | h of layer 1] -- the B operands of a layer's stream,
    // [input part | recurrent part], are then ONE run of k-steps (layer 0: x, h0; layer 1: h0, h1)
    float* const xs0 = smem + (size_t)wave * (size_t)(kx + 2 * HPK) * 64;
    float* const hc0 = xs0 + (size_t)kx * 64;
    float* const hc1 = hc0 + (size_t)HPK * 64;
    auto hcol = [&](int l) { return l == 0 ? hc0 : hc1; };

    // ---- per-row constants
    const int hh0 = lane >> 5;
    const long row_raw = rowtile * 32 + (lane & 31);
    const uint32_t row32 = (uint32_t)(row_raw < a.n_rows ? row_raw : a.n_rows - 1);
    auto div32 = [](long dv) { return (uint32_t)(dv > 0x7fffffffL ? 0x7fffffffL : dv); };
    const uint32_t day_rel = row32 / div32(d.S);
    const uint32_t sample = row32 - day_rel * div32(d.S);
    const uint32_t day = (uint32_t)d.day0 + day_rel;
    const float* const ctxp = d.ctx ? d.ctx + (long)(d.ctx_div == 1 ? row32 : row32 / div32(d.ctx_div)) * d.ctx_rs : nullptr;
    const long z_idx = d.zb ? (long)(d.z_div == 1 ? row32 : row32 / div32(d.z_div)) * d.z_rs : 0;
    const float* const epsp = d.eps_in ? d.eps_in + (size_t)row32 * d.eps_T * L : nullptr;
    const bool has_state = d.state_in != nullptr;                 // wave-uniform
    const bool shared0 = d.gate_in != nullptr;                    // first step from a shared state with its gates given
    const uint32_t state_row = has_state ? ((d.state_div == d.S) ? day_rel : row32 / div32(d.state_div)) : 0u;
    const float* const gate_in = shared0 ? d.gate_in + (size_t)state_row * 2 * NT * 32 : nullptr;
    // a shared first step without context: of layer 0's input quads only those that hold z are multiplied
    const int qx0 = Lc >> 3, qxn = ((Lc + L + 7) >> 3) - qx0;
    const bool zfast = shared0 && d.ctx_T <= 0;

    // x part of layer 0 at step t: [ctx | z | alignment padding]; k-step s, half hh <-> feature 2s + hh.  A lone wave
    // pays every load's round trip it waits for, so all of a step's loads go out together, from clamped addresses
    // (kx <= 16 slots, L <= 16 latents: checked by the launcher), and the values are selected afterwards.
    auto assemble_x = [&](int t, int lv) {
        constexpr int KXM = 16, NB = 4;
        const int hh = lv >> 5;
        float* const xs = xs0 + lv;
        const bool given = t < d.z_T;                             // z_t = zb (+ exp(lv / 2) eps); else mu0 + std0 eps
        const bool noisy = given ? (d.lv != nullptr) : (d.std0 != 0.0f);
        const bool has_ctx = t < d.ctx_T;
        const int par = (hh - Lc) & 1;                            // parity of this lane's latent indices
        const int te = t - d.eps_t0;
        // ---- loads
        float cv[KXM], zv[KXM], zn[2 * NB], lvv[2 * NB], ev[2 * NB];
        if (has_ctx) {
            const float* const cp = ctxp + (long)t * Lc;
#pragma unroll
            for (int s = 0; s < KXM; ++s) {
                const int f = 2 * s + hh;
                cv[s] = cp[f < Lc ? f : Lc - 1];
            }
        }
        if (given) {
            const float* const zp = d.zb + z_idx + (long)t * L;
#pragma unroll
            for (int s = 0; s < KXM; ++s) {
                const int j = 2 * s + hh - Lc;
                zv[s] = zp[j < 0 ? 0 : (j < L ? j : L - 1)];
            }
            if (noisy) {
                const float* const lp = d.lv + z_idx + (long)t * L;
#pragma unroll
                for (int i = 0; i < 2 * NB; ++i) {
                    const int j = 2 * i + par;
                    zn[i] = zp[j < L ? j : L - 1];
                    lvv[i] = lp[j < L ? j : L - 1];
                }
            }
        }
        if (noisy && epsp) {
            const float* const ep = epsp + (size_t)te * L;
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) {
                const int j = 2 * i + par;
                ev[i] = ep[j < L ? j : L - 1];
            }
        }
        // ---- the noiseless vector
#pragma unroll
        for (int s = 0; s < KXM; ++s) {
            if (s < kx) {
                const int f = 2 * s + hh, j = f - Lc;
                float v = 0.0f;
                if (f < Lc) v = has_ctx ? cv[s] : 0.0f;
                else if (j < L) v = given ? zv[s] : d.mu0;
                xs[s * 64] = v;
            }
        }
        // ---- noise: one Philox block gives the latents 4b .. 4b+3 (philox_normal_elem: word pair j & 2, cos / sin by
        // j & 1); this lane holds the two of them with its parity
        if (noisy) {
            const uint32_t k0 = (uint32_t)d.seed, k1 = (uint32_t)(d.seed >> 32);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (4 * b < L) {
                    uint32_t c0 = (uint32_t)opaque((int)sample), c1 = day, c2 = (uint32_t)(d.step0 + te), c3 = (uint32_t)b;
                    if (!epsp) philox4x32_10(k0, k1, c0, c1, c2, c3);
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int j = 4 * b + 2 * p + par;
                        float e;
                        if (epsp) {
                            e = ev[2 * b + p];
                        } else {
                            const uint32_t xa = p ? c2 : c0, xb = p ? c3 : c1;
                            const float u1 = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-07f;
                            const float u2 = ((float)(xb >> 9) + 0.5f) * 1.1920928955078125e-07f;
                            const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
                            e = r * (par ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
                        }
                        float base, scale;
                        if (given) {
                            base = zn[2 * b + p];
                            scale = expf(0.5f * lvv[2 * b + p]);
                        } else {
                            base = d.mu0;
                            scale = d.std0;
                        }
                        // (this lane's latent j sits in slot (j + Lc - hh) / 2)
                        if (j < L) xs[((j + Lc - hh) >> 1) * 64] = base + scale * e;
                    }
                }
            }
        }
    };

    // ---- state: c in (accumulator) registers, h in the LDS columns (a shared first step never reads the old h)
    float c[2][NT][4];                                            // (values live in AGPRs: agpr_put / agpr_get)
    {
        const float* const st = has_state ? d.state_in + (size_t)state_row * 2 * 2 * HP : nullptr;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            float* const hb = hcol(l) + lane;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float4 hv = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cv = hv;
                if (has_state) {
                    const float* ps = st + (size_t)l * 2 * HP + 8 * i + 4 * hh0;
                    if (!shared0) hv = *(const float4*)ps;
                    cv = *(const float4*)(ps + HP);
                }
                c[l][i][0] = agpr_put(cv.x); c[l][i][1] = agpr_put(cv.y); c[l][i][2] = agpr_put(cv.z); c[l][i][3] = agpr_put(cv.w);
                if (!shared0) {
                    hb[(4 * i + 0) * 64] = hv.x; hb[(4 * i + 1) * 64] = hv.y;
                    hb[(4 * i + 2) * 64] = hv.z; hb[(4 * i + 3) * 64] = hv.w;
                }
            }
        }
    }
    assemble_x(0, lane);

    // A pass's first fetches -- the gate accumulators' initial values (bias, or a shared first step's given gates)
    // and quad 0 of its weight stream -- are requested one pass early, in front of the previous pass's cell math
    // (and the head's first layer in front of the memory's last cell): with one wave per SIMD nothing else covers
    // an L2 round trip.
    auto pass_w = [&](int l, bool shared) {                       // layer l's stream: [input quads from q0 | recurrent quads]
        const bool zq = (l == 0) && shared && zfast;
        return (const float4*)a.w[l] + (size_t)(zq ? qx0 : 0) * NT * 64;
    };
    // (the bias as ONE register per tile, see bias_frag; a shared first step's gates are per row: sixteen registers
    // per tile, loaded where the pass starts)
    auto pass_fetch = [&](int l, int p, bool shared, float (&bf)[4], float4 (&a0)[4]) {
        const int lv = opaque(lane);
        wload_buf<4>(a0, make_wrsrc(pass_w(l, shared)), (unsigned)lv * 16u, (unsigned)p * 4096u);     // tiles 4p .. 4p + 3
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = bias_frag(a.b[l], 4 * p + i, lv);
    };
    float4 fa[4 * kHeadTiles];                                    // the head's fragment buffer (dec_seq_head)
    auto head_fetch = [&](float (&bf)[4]) {
        const DenseLayerArg& ly = a.head.layer[0];
        const int lv = opaque(lane);
        band_load_all<4 * kHeadTiles>(fa, make_wrsrc(ly.w), (unsigned)lv * 16u, 0u, (unsigned)ly.nt * 1024u, 0, 4 * kHeadTiles);
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = bias_frag(ly.b, i, lv);
    };

    float bf[4];
    float4 a0[4];
    pass_fetch(0, 0, shared0, bf, a0);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const bool shared = shared0 && t == 0;                    // wave-uniform
        const bool head_now = t >= a.t_out0;
        float hold[NT][4];                                        // the layer's new h, unit 8*tile + 2*jj + hh
        const int dsq_t = t;                                      // (stamps)
        (void)dsq_t;
        DSQ_STAMP(0);
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const int nq_in = (l == 0) ? (kx >> 2) : NT;
            const bool zq = (l == 0) && shared && zfast;
            const int q0 = zq ? qx0 : 0;
            const int nqi = nq_in - q0;                           // input quads from q0 on
            const int nqs = shared ? (zq ? qxn : nqi) : nqi + NT; // quads streamed: [input part | recurrent part]
            const float4* const w = pass_w(l, shared);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int lv = opaque(lane);
                float hn[4][4];
                const wrsrc_t wr = make_wrsrc(w);
                const unsigned voff = (unsigned)lv * 16u, sb = (unsigned)p * 4096u;      // tiles 4p .. 4p + 3
                const float* const bin = ((l == 0) ? xs0 : hc0) + lv + (size_t)(4 * q0) * 64;   // layer 1 reads h0
                f32x16 acc[4];
                if (shared) {
                    const int hh = lv >> 5;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4* gp = (const float4*)(gate_in + (size_t)l * NT * 32 + (4 * p + i) * 32 + hh * 16);
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float4 x = gp[v];
                            acc[i][4 * v + 0] = x.x; acc[i][4 * v + 1] = x.y; acc[i][4 * v + 2] = x.z; acc[i][4 * v + 3] = x.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = bias_acc(bf[i], lv < 32 ? 1.0f : 0.0f);
                }
                VSV_DSEQ_STREAM<4>(acc, a0, wr, voff, sb, NT * 1024u, nqs, [&](int q, float (&bv)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bv[j] = bin[(size_t)(4 * q + j) * 64];
                });
                DSQ_STAMP(1 + 2 * (NP * l + p));
                // the next stage's first fetches, in front of this pass's cell
                float bfn[4];
                float4 a0n[4];
                if (p + 1 < NP) pass_fetch(l, p + 1, shared, bfn, a0n);
                else if (l == 0) pass_fetch(1, 0, shared, bfn, a0n);
                else if (!head_now) pass_fetch(0, 0, false, bfn, a0n);       // (t + 1 == T: fetched and dropped)
                else {                                            // (the head fetches for itself, behind this cell)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        bfn[i] = bf[i];
                        a0n[i] = a0[i];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                float cc[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) cc[i][jj] = agpr_get(c[l][4 * p + i][jj]);
                lstm_cell_pk(acc, cc, hn);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) c[l][4 * p + i][jj] = agpr_put(cc[i][jj]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        asm volatile("" : "+v"(hn[i][jj]));
                        hold[4 * p + i][jj] = hn[i][jj];
                    }
                __builtin_amdgcn_sched_barrier(0);
                DSQ_STAMP(2 + 2 * (NP * l + p));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf[i] = bfn[i];
                    a0[i] = a0n[i];
                }
            }
            // (the head's first fetches: in flight while the column is written)
            if (l == 1 && head_now) head_fetch(bf);
            // every pass has read the layer's old h: the column takes the new one (natural k-step order)
            float* const colw = hcol(l) + opaque(lane);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) colw[(4 * i + jj) * 64] = hold[i][jj];
            // x of the next step: layer 0 has read this step's
            if (l == 0 && t + 1 < T) assemble_x(t + 1, opaque(lane));
            if (l == 1) DSQ_STAMP(17);
        }
        if (head_now) {
            // the head's input is the top layer's new h, still in registers (k-step 4 i + jj <-> hold[i][jj])
            float act[16 * kHeadTiles];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) act[4 * i + jj] = hold[i][jj];
            dec_seq_head<NT>(a, act, lane, rowtile, t - a.t_out0, day_rel, sample, fa, bf, t, wave);
            if (t + 1 < T) pass_fetch(0, 0, false, bf, a0);
        }
        DSQ_STAMP(26);
    }
}

// ---------------------------------------------------------------------------
// enc_seq_kernel (round 3): an encoder's two memory layers and its latent head over T steps for MANY rows whose
// first-layer input products come from a per-day table (pre_t: the per-row encoder steps of an autoregressive rollout,
// vsv_encode_from_pre), on dec_seq_kernel's schedule: ONE WAVE per 32-row tile and SIMD, no workgroup barrier, the
// cell state of both layers (2 x 13 x 4 registers) parked in accumulator registers, h of both layers in the wave's
// own LDS columns [h0 | h1] (26 KB per wave), a layer's new h held in registers until the last pass has read the old
// one.  The 13 gate tiles of a layer run in passes of 4 + 3 + 3 + 3; layer 0's accumulators start from the table
// (requested one pass ahead), layer 1's from the bias fragment.  The head (one tile: z_mean | z_log_var, or the
// context compression) multiplies the top layer's new h straight from registers with fragments that stay resident
// for the whole launch, and stores (row, step, feature) as the dense chain behind lstm_kernel did.  Same products in
// the same k order as lstm_kernel + dense_chain_kernel: results are bit-equal.
// ---------------------------------------------------------------------------
template <int NC, typename BL>
__device__ __forceinline__ void enc_stream(f32x16 (&acc)[4], float4 (&a0)[4], wrsrc_t r, unsigned voff, unsigned sbase,
                                           unsigned qsb, int nq, BL bl) {
    float4 a1[4];
    float b0[4], b1[4];
    auto quad = [&](const float4 (&aa)[4], const float (&bb)[4]) {
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = mfma32(aa[i].x, bb[0], acc[i]);
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = mfma32(aa[i].y, bb[1], acc[i]);
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = mfma32(aa[i].z, bb[2], acc[i]);
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = mfma32(aa[i].w, bb[3], acc[i]);
    };
    auto pattern = [] {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one product
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one fragment load
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one LDS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * NC - 4, 0);
    };
    bl(0, b0);
    int q = 0;
#pragma unroll 1
    for (; q + 1 < nq; q += 2) {
#pragma unroll
        for (int i = 0; i < NC; ++i) a1[i] = bload(r, voff, sbase + (unsigned)(q + 1) * qsb + (unsigned)i * 1024u);
        bl(q + 1, b1);
        quad(a0, b0);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
        const int q2 = (q + 2 < nq) ? q + 2 : q + 1;
#pragma unroll
        for (int i = 0; i < NC; ++i) a0[i] = bload(r, voff, sbase + (unsigned)q2 * qsb + (unsigned)i * 1024u);
        bl(q2, b0);
        quad(a1, b1);
        pattern();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (q < nq) quad(a0, b0);
}

template <int NT>
__global__ void __launch_bounds__(256, 1) enc_seq_kernel(const LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(NT == 13, "built for 13 gate tiles (memory hidden 97..104)");
    constexpr int HP = 8 * NT, HPK = 4 * NT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long rowtile = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    if (rowtile >= a.n_rowtiles) return;                          // (no barriers below: a wave may leave alone)
    const int T = a.T;
    float* const hc0 = smem + (size_t)wave * (size_t)(2 * HPK) * 64;   // this wave's columns, without the lane: [h0 | h1]
    float* const hc1 = hc0 + (size_t)HPK * 64;
    auto hcol = [&](int l) { return l == 0 ? hc0 : hc1; };
    const long row_raw = rowtile * 32 + (lane & 31);
    const bool row_ok = row_raw < a.n_rows;
    const uint32_t row32 = (uint32_t)(row_ok ? row_raw : a.n_rows - 1);
    const bool has_state = a.din.state_in != nullptr;             // wave-uniform
    const uint32_t state_row = has_state ? (a.din.state_div == 1 ? row32 : row32 / (uint32_t)(a.din.state_div > 0x7fffffffL ? 0x7fffffffL : a.din.state_div)) : 0u;
    // layer 0 streams its recurrent quads only: its input product comes from the table
    const float4* const w0 = (const float4*)a.w[0] + (size_t)(a.kx >> 2) * NT * 64;
    const float4* const w1 = (const float4*)a.w[1];
    const DenseLayerArg& hd = a.head.layer[0];

    // ---- state: c in accumulator registers, h in the LDS columns
    float c[2][NT][4];
    {
        const float* const st = has_state ? a.din.state_in + (size_t)state_row * 2 * 2 * HP : nullptr;
        const int hh0 = lane >> 5;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            float* const hb = hcol(l) + lane;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float4 hv = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cv = hv;
                if (has_state) {
                    const float* ps = st + (size_t)l * 2 * HP + 8 * i + 4 * hh0;
                    hv = *(const float4*)ps;
                    cv = *(const float4*)(ps + HP);
                }
                c[l][i][0] = agpr_put(cv.x); c[l][i][1] = agpr_put(cv.y); c[l][i][2] = agpr_put(cv.z); c[l][i][3] = agpr_put(cv.w);
                hb[(4 * i + 0) * 64] = hv.x; hb[(4 * i + 1) * 64] = hv.y; hb[(4 * i + 2) * 64] = hv.z; hb[(4 * i + 3) * 64] = hv.w;
            }
        }
    }
    // the head's fragments (one output tile, NT quads) and bias fragment stay in registers
    float4 hw[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) hw[q] = ((const float4*)hd.w)[q * 64 + lane];
    const float hbf = bias_frag(hd.b, 0, lane);

    // first fetches of a pass, requested one pass ahead: quad 0 of its weight stream and its accumulators' initial
    // values -- layer 0: the table's gate values of step t (value j = 2 (16 tile + r) + lane half, column row + t *
    // pre_tstride), layer 1: the bias fragments
    auto fetch_w = [&](const float4* w, int tile0, int nc, float4 (&a0)[4]) {
        const wrsrc_t wr = make_wrsrc(w);
        const unsigned voff = (unsigned)opaque(lane) * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nc) a0[i] = bload(wr, voff, (unsigned)(tile0 + i) * 1024u);
    };
    auto fetch_tab = [&](int t, int tile0, int nc, f32x16 (&acc)[4]) {
        const int lv = opaque(lane);
        const float* const p = a.pre_t + (size_t)(lv >> 5) * a.pre_stride + row32 + (size_t)t * a.pre_tstride;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = p[(size_t)(2 * (16 * (tile0 + i) + r)) * a.pre_stride];
            }
    };
    auto fetch_bias = [&](int tile0, int nc, float (&bf)[4]) {
        const int lv = opaque(lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nc) bf[i] = bias_frag(a.b[1], tile0 + i, lv);
    };

    f32x16 acc[4];
    float4 a0[4];
    float bf[4];
    fetch_w(w0, 0, 4, a0);
    fetch_tab(0, 0, 4, acc);
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        float hold[NT][4];                                        // the layer's new h, unit 8*tile + 2*jj + hh
        // one pass: tiles tile0 .. tile0 + NC - 1 of layer l; then the next pass's first fetches, the cell, the hold
        auto pass = [&](auto l_, auto tile0_, auto nc_, auto nl_, auto ntile0_, auto nnc_) {
            constexpr int l = decltype(l_)::value, tile0 = decltype(tile0_)::value, NC = decltype(nc_)::value;
            constexpr int nl = decltype(nl_)::value, ntile0 = decltype(ntile0_)::value, nnc = decltype(nnc_)::value;
            const int lv = opaque(lane);
            const wrsrc_t wr = make_wrsrc(l == 0 ? w0 : w1);
            const unsigned voff = (unsigned)lv * 16u, sb = (unsigned)tile0 * 1024u;
            if (l == 1) {
#pragma unroll
                for (int i = 0; i < NC; ++i) acc[i] = bias_acc(bf[i], lv < 32 ? 1.0f : 0.0f);
            }
            // layer 0: the recurrent quads (h0); layer 1: [h0 | h1], one run of k-steps
            const float* const bin = hc0 + lv;
            enc_stream<NC>(acc, a0, wr, voff, sb, NT * 1024u, l == 0 ? NT : 2 * NT, [&](int q, float (&bv)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = bin[(size_t)(4 * q + j) * 64];
            });
            f32x16 accn[4];
            float4 a0n[4];
            float bfn[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0n[i] = a0[i];
                bfn[i] = bf[i];
            }
            if (nl == 0) {
                if (l == 0 || t + 1 < T) {                        // (l == 1: the next step's first pass)
                    fetch_w(w0, ntile0, nnc, a0n);
                    fetch_tab(l == 0 ? t : t + 1, ntile0, nnc, accn);
                }
            } else {
                fetch_w(w1, ntile0, nnc, a0n);
                fetch_bias(ntile0, nnc, bfn);
            }
            __builtin_amdgcn_sched_barrier(0);
            float cc[NC][4], hn[NC][4];
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) cc[i][jj] = agpr_get(c[l][tile0 + i][jj]);
            {
                f32x16 (&accv)[NC] = reinterpret_cast<f32x16 (&)[NC]>(acc);
                lstm_cell_pk<NC>(accv, cc, hn);
            }
#pragma unroll
            for (int i = 0; i < NC; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    c[l][tile0 + i][jj] = agpr_put(cc[i][jj]);
                    asm volatile("" : "+v"(hn[i][jj]));
                    hold[tile0 + i][jj] = hn[i][jj];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0[i] = a0n[i];
                bf[i] = bfn[i];
                if (nl == 0) acc[i] = accn[i];
            }
        };
        auto write_col = [&](int l) {
            // every pass has read the layer's old h: the column takes the new one (natural k-step order)
            float* const colw = hcol(l) + opaque(lane);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) colw[(4 * i + jj) * 64] = hold[i][jj];
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>;
        using I7 = std::integral_constant<int, 7>; using I10 = std::integral_constant<int, 10>;
        pass(I0{}, I0{}, I4{}, I0{}, I4{}, I3{});
        pass(I0{}, I4{}, I3{}, I0{}, I7{}, I3{});
        pass(I0{}, I7{}, I3{}, I0{}, I10{}, I3{});
        pass(I0{}, I10{}, I3{}, I1{}, I0{}, I4{});
        write_col(0);
        pass(I1{}, I0{}, I4{}, I1{}, I4{}, I3{});
        pass(I1{}, I4{}, I3{}, I1{}, I7{}, I3{});
        pass(I1{}, I7{}, I3{}, I1{}, I10{}, I3{});
        pass(I1{}, I10{}, I3{}, I0{}, I0{}, I4{});
        write_col(1);
        // ---- head: one output tile on the top layer's new h, straight from registers (k-step 4 q + j <-> hold[q][j])
        {
            const int lv = opaque(lane);
            const int hh = lv >> 5;
            f32x16 hacc = bias_acc(hbf, lv < 32 ? 1.0f : 0.0f);
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                hacc = mfma32(hw[q].x, hold[q][0], hacc);
                hacc = mfma32(hw[q].y, hold[q][1], hacc);
                hacc = mfma32(hw[q].z, hold[q][2], hacc);
                hacc = mfma32(hw[q].w, hold[q][3], hacc);
            }
            const long rr = rowtile * 32 + (lv & 31);
            if (rr < a.n_rows) {
                const int nh = a.head.out_f;
                float* const po = a.head.out + ((size_t)rr * T + t) * nh + 4 * hh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fo = 8 * (r >> 2) + (r & 3);              // feature fo + 4 hh
                    if (fo + 4 * hh < nh) po[fo] = hacc[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// dense_bwd_kernel: backward of a dense chain for independent (row tile, t)
// instances, one wave per instance.  col holds the gradient w.r.t. a forward
// layer's OUTPUT in that layer's slot order (slot 16t + reg <-> feature
// 32t + 8(reg>>2) + 4hh + (reg&3)); per layer, last forward layer first:
//   g = (col + add) * [forward output > 0 where the layer has a ReLU]   -> g_out (for dW = g^T . input)
//   col = W^T . g   (transposed fragments whose output order is the previous layer's slot order)
// ---------------------------------------------------------------------------
typedef vsv_dense_bwd_layer DenseBwdLayer;
typedef vsv_dense_bwd DenseBwdArgs;

template <int NT>
__device__ __forceinline__ void dense_bwd_layer(const DenseBwdLayer& ly, float* col, int lane) {
    f32x16 acc[NT];
    int toff[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        toff[i] = i * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    }
    gemm_seg_deep<NT>(acc, (const float4*)ly.wT, lane, toff, NT * 64, col, ly.nqT);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) col[(16 * tt + r) * 64] = acc[tt][r];
    for (int s = 16 * NT; s < kDenseSlots; ++s) col[s * 64] = 0.0f;
}

// Element-wise phase of a dense-chain backward layer for sixteen slots s0 .. s0 + 15 of tile `tile`:
//   g = (col + add) * [forward output > 0 where the layer has a ReLU]  -> col, g_out
// Every global load of the group is issued before the first store and none sits behind a per-value branch
// (addresses are clamped instead, values selected afterwards): written value by value -- conditional load, use,
// store -- the compiler emitted a load and an s_waitcnt vmcnt(0) per value, thirty-two dependent L2 round trips
// per layer and wave.
__device__ __forceinline__ void dense_bwd_elementwise(const DenseBwdLayer& ly, float* col, const float* act, const float* add,
                                                      bool row_ok, float* go, int s0, int tile, int hh) {
    float av[16], ad[16];
    const bool has_add = ly.add != nullptr;                       // (uniform; `add` is valid for every lane: rows are clamped)
    if (act) {
#pragma unroll
        for (int k = 0; k < 16; ++k) av[k] = act[(s0 + k) * 64];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) av[k] = 1.0f;
    }
    if (has_add) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int f = feat_of(tile, k, hh);
            const int fc = f < ly.add_f0 ? ly.add_f0 : (f >= ly.add_f0 + ly.add_n ? ly.add_f0 + ly.add_n - 1 : f);
            ad[k] = add[fc];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) ad[k] = 0.0f;
    }
    float g[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int f = feat_of(tile, k, hh);
        g[k] = col[(s0 + k) * 64];
        if (has_add && row_ok && f >= ly.add_f0 && f < ly.add_f0 + ly.add_n) g[k] += ad[k];
        if (act && f < ly.n_relu && !(av[k] > 0.0f)) g[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) col[(s0 + k) * 64] = g[k];
    if (go) {
#pragma unroll
        for (int k = 0; k < 16; ++k) go[(s0 + k) * 64] = g[k];
    }
}

__global__ void __launch_bounds__(256, 2) dense_bwd_kernel(const DenseBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long inst = (long)blockIdx.x * 4 + wave;
    if (inst >= a.n_rowtiles * a.T) return;
    const long rowtile = inst / a.T;
    const int t = (int)(inst % a.T);
    const int hh = lane >> 5, m = lane & 31;
    long row = rowtile * 32 + m;
    const bool row_ok = row < a.n_rows;
    if (!row_ok) row = a.n_rows - 1;
    float* col = smem + (size_t)wave * (kDenseSlots * 64) + lane;
    if (a.in_mode == 0) {
        // (row and feature clamped, value selected: no load behind a per-lane branch)
        const float* src = a.in + ((size_t)row * a.T + t) * a.in_f;
#pragma unroll 1
        for (int s0 = 0; s0 < kDenseSlots; s0 += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int f = feat_of(s0 >> 4, k, hh);
                v[k] = src[f < a.in_f ? f : a.in_f - 1];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) col[(s0 + k) * 64] = (row_ok && feat_of(s0 >> 4, k, hh) < a.in_f) ? v[k] : 0.0f;
        }
    } else {
        const float* src = a.in + (size_t)inst * kDenseSlots * 64 + lane;
#pragma unroll 1
        for (int s0 = 0; s0 < kDenseSlots; s0 += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[(s0 + k) * 64];
#pragma unroll
            for (int k = 0; k < 16; ++k) col[(s0 + k) * 64] = row_ok ? v[k] : 0.0f;
        }
    }
#pragma unroll 1
    for (int li = 0; li < a.n_layers; ++li) {
        const DenseBwdLayer& ly = a.layer[li];
        const float* act = ly.act_out ? ly.act_out + (size_t)inst * kDenseSlots * 64 + lane : nullptr;
        const float* add = ly.add ? ly.add + ((size_t)row * a.T + t) * ly.add_n - ly.add_f0 : nullptr;
        float* go = ly.g_out ? ly.g_out + (size_t)inst * kDenseSlots * 64 + lane : nullptr;
#pragma unroll 1
        for (int s0 = 0; s0 < kDenseSlots; s0 += 16) dense_bwd_elementwise(ly, col, act, add, row_ok, go, s0, s0 >> 4, hh);
        switch (ly.ntT) {
            case 0: break;
            case 1: dense_bwd_layer<1>(ly, col, lane); break;
            case 2: dense_bwd_layer<2>(ly, col, lane); break;
            case 3: dense_bwd_layer<3>(ly, col, lane); break;
            default: dense_bwd_layer<4>(ly, col, lane); break;
        }
    }
    if (a.dx_out) {
        float* o = a.dx_out + (size_t)inst * kDenseSlots * 64 + lane;
#pragma unroll 16
        for (int s = 0; s < kDenseSlots; ++s) o[s * 64] = col[s * 64];
    }
}

// dense_bwd_wg_kernel: the same chain with one WORKGROUP per instance, for few instances (a training batch of 64
// at horizon 1 is two instances of the decoder head: one wave each walked five layers alone, 128 us).  The column
// is shared in LDS; wave w handles slots 16w .. 16w+15 in the element-wise phase and output tile w of W^T . g.
// The per-tile product chains are those of dense_bwd_kernel: results are bit-equal.
__device__ __forceinline__ void dense_bwd_wg_body(const DenseBwdArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long inst = (long)blockIdx.x;
    const long rowtile = inst / a.T;
    const int t = (int)(inst % a.T);
    const int hh = lane >> 5, m = lane & 31;
    long row = rowtile * 32 + m;
    const bool row_ok = row < a.n_rows;
    if (!row_ok) row = a.n_rows - 1;
    float* col = smem + lane;
    const int s_lo = 16 * wave;
    {
        float v[16];
        if (a.in_mode == 0) {
            // (row and feature clamped, value selected: no load behind a per-lane branch)
            const float* src = a.in + ((size_t)row * a.T + t) * a.in_f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int f = feat_of(wave, k, hh);
                v[k] = src[f < a.in_f ? f : a.in_f - 1];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) col[(s_lo + k) * 64] = (row_ok && feat_of(wave, k, hh) < a.in_f) ? v[k] : 0.0f;
        } else {
            const float* src = a.in + (size_t)inst * kDenseSlots * 64 + lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = src[(s_lo + k) * 64];
#pragma unroll
            for (int k = 0; k < 16; ++k) col[(s_lo + k) * 64] = row_ok ? v[k] : 0.0f;
        }
    }
#pragma unroll 1
    for (int li = 0; li < a.n_layers; ++li) {
        const DenseBwdLayer& ly = a.layer[li];
        const float* act = ly.act_out ? ly.act_out + (size_t)inst * kDenseSlots * 64 + lane : nullptr;
        const float* add = ly.add ? ly.add + ((size_t)row * a.T + t) * ly.add_n - ly.add_f0 : nullptr;
        float* go = ly.g_out ? ly.g_out + (size_t)inst * kDenseSlots * 64 + lane : nullptr;
        // this wave's sixteen slots (it wrote them itself: no barrier needed before)
        dense_bwd_elementwise(ly, col, act, add, row_ok, go, s_lo, wave, hh);
        if (ly.ntT > 0) {                                          // (block-uniform)
            __syncthreads();                                       // g complete
            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
            if (wave < ly.ntT) {
                const int toff1[1] = {wave * 64};
                gemm_seg_deep<1>(acc, (const float4*)ly.wT, lane, toff1, ly.ntT * 64, col, ly.nqT);
            }
            __syncthreads();                                       // every wave has read g
#pragma unroll
            for (int r = 0; r < 16; ++r) col[(s_lo + r) * 64] = acc[0][r];        // (tiles >= ntT: zeros)
        }
    }
    if (a.dx_out) {
        float* o = a.dx_out + (size_t)inst * kDenseSlots * 64 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[(s_lo + k) * 64] = col[(s_lo + k) * 64];
    }
}
__global__ void __launch_bounds__(256) dense_bwd_wg_kernel(const DenseBwdArgs a) { dense_bwd_wg_body(a); }
// two launches as one (blockIdx.y picks the argument set; see lstm_rec4_pair_kernel)
__global__ void __launch_bounds__(256) dense_bwd_wg_pair_kernel(const DenseBwdArgs a0, const DenseBwdArgs a1) {
    if (blockIdx.y == 0) {
        if ((long)blockIdx.x < a0.n_rowtiles * a0.T) dense_bwd_wg_body(a0);
    } else {
        if ((long)blockIdx.x < a1.n_rowtiles * a1.T) dense_bwd_wg_body(a1);
    }
}

// ---------------------------------------------------------------------------
// lstm_bwd_kernel: BPTT of a 1- or 2-layer LSTM for a tile of 32 rows per
// workgroup (4 waves).  Wave w owns the hidden units of gate tiles w, w+4, ...,
// i.e. "unit tile" w: accumulator register 4*jj' + r' of that tile holds unit
// 8*(4*jj' + w) + 2*r' + hh -- the unit whose gates the same lane produced in the
// forward kernel, so saved gates / cell states are read back lane-for-lane.
// Per step (t = T-1 .. 0) and layer (top first):
//   dh  = [dh_top(t) | W_in(l+1)^T . dgates(l+1, t)] + W_hh(l)^T . dgates(l, t+1)
//   dgates(l, t) from dh, dc, the saved gates and cell states  -> LDS (B operand) + global (for dW)
//   dx(t) = W_in(0)^T . dgates(0, t)
// ---------------------------------------------------------------------------
typedef vsv_lstm_bwd LstmBwdArgs;

template <int NT, int NL>
__global__ void __launch_bounds__(256, 1) lstm_bwd_kernel(const LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int GK = 16 * NT;                                   // k-steps of one layer's gate rows
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long rowtile = (long)blockIdx.x;
    const int hh = lane >> 5, m = lane & 31;
    const long row_raw = rowtile * 32 + m;
    const bool row_ok = row_raw < a.n_rows;
    float* dg = smem + lane;                                      // dgates of layer l: dg + l*GK*64
    for (int s = wave; s < NL * GK; s += 4) dg[s * 64] = 0.0f;
    float dc[NL][16];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) dc[l][r] = 0.0f;
    const int toff1[1] = {wave * 64};
    const int t_outs = a.T - a.t_out0;
    // layers of the saved arrays / first of them handled here (one launch per layer: NL = 1, sv_nl = 2)
    const int SL = a.sv_nl > 0 ? a.sv_nl : NL, L0 = a.sv_nl > 0 ? a.sv_l0 : 0;
    // One layer per launch (the training schedule): this wave's W_hh^T fragments -- its 32 hidden units against
    // all 4H gate rows, 4*NT quads -- stay in REGISTERS for the whole time loop (the kernel runs one wave per
    // SIMD, 512 VGPRs each), so a step streams nothing from L2 and starts its products as soon as dgates(t+1)
    // is in LDS; the dependent chain of 16*NT products runs as two interleaved chains.
    constexpr bool WREG = (NL == 1);
    float4 wreg[WREG ? 4 * NT : 1];
    if constexpr (WREG) {
        const float4* wT = (const float4*)a.wT_hh[0] + lane + wave * 64;
#pragma unroll
        for (int q = 0; q < 4 * NT; ++q) wreg[q] = wT[(size_t)q * 4 * 64];
    }
    __syncthreads();

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t inst = (size_t)(rowtile * a.T + t);
#pragma unroll
        for (int l = NL - 1; l >= 0; --l) {
            f32x16 acc[1];
            // saved gates and cell states of this step: issued before the matrix products so that they
            // land while those run (nothing here depends on the products)
            float gsv[16][4], cnv[16], cpv[16];
            {
                const float* sg = a.sv_gates + (inst * SL + L0 + l) * GK * 64 + lane;
                const float* sc = a.sv_c + (inst * SL + L0 + l) * (4 * NT) * 64 + lane;
                const float* scp = (t > 0) ? a.sv_c + ((inst - 1) * SL + L0 + l) * (4 * NT) * 64 + lane : nullptr;
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    const int tg = 4 * jq + wave;
                    const int tc = tg < NT ? tg : NT - 1;              // clamped: out-of-range tiles are never used
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int reg = 4 * jq + r, gs = 16 * tc + 4 * r, cs = 4 * tc + r;
#pragma unroll
                        for (int g = 0; g < 4; ++g) gsv[reg][g] = sg[(gs + g) * 64];
                        cnv[reg] = sc[cs * 64];
                        cpv[reg] = scp ? scp[cs * 64] : 0.0f;
                    }
                }
            }
            if (l == NL - 1) {
                if (t >= a.t_out0 && a.dh_top) {
                    const float* p = a.dh_top + ((size_t)(rowtile * t_outs + (t - a.t_out0)) * 64 + 16 * wave) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][r] = p[r * 64];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
                gemm_seg<1>(acc, (const float4*)a.wT_in[l + 1], lane, toff1, 4 * 64, dg + (size_t)(l + 1) * GK * 64, 4 * NT);
                if (a.drop_p > 0.0f) {
                    // layer l+1 read the dropped copy of this layer's h: same Philox factors as the forward
                    const DropArgs da = resolve_drop(DropArgs{a.drop_p, 1.0f / (1.0f - a.drop_p), (uint32_t)a.drop_seed,
                                                              (uint32_t)(a.drop_seed >> 32), (uint32_t)a.drop_id,
                                                              (const uint32_t*)a.drop_seed_dev});
                    const uint32_t rowc = (uint32_t)(row_ok ? row_raw : a.n_rows - 1);
#pragma unroll
                    for (int jq = 0; jq < 4; ++jq) {
                        float fct[4];
                        dropout_factors(da, rowc, (uint32_t)t, 4 * jq + wave, hh, l, fct);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[0][4 * jq + r] *= fct[r];
                    }
                }
            }
            if (t + 1 < a.T) {
                if constexpr (WREG) {
                    f32x16 acc2;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
                    const float* b = dg;
#pragma unroll
                    for (int q = 0; q < 4 * NT; q += 2) {
                        const float b0 = b[(4 * q + 0) * 64], b1 = b[(4 * q + 1) * 64], b2 = b[(4 * q + 2) * 64],
                                    b3 = b[(4 * q + 3) * 64], b4 = b[(4 * q + 4) * 64], b5 = b[(4 * q + 5) * 64],
                                    b6 = b[(4 * q + 6) * 64], b7 = b[(4 * q + 7) * 64];
                        acc[0] = mfma32(wreg[q].x, b0, acc[0]);
                        acc2 = mfma32(wreg[q + 1].x, b4, acc2);
                        acc[0] = mfma32(wreg[q].y, b1, acc[0]);
                        acc2 = mfma32(wreg[q + 1].y, b5, acc2);
                        acc[0] = mfma32(wreg[q].z, b2, acc[0]);
                        acc2 = mfma32(wreg[q + 1].z, b6, acc2);
                        acc[0] = mfma32(wreg[q].w, b3, acc[0]);
                        acc2 = mfma32(wreg[q + 1].w, b7, acc2);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][r] += acc2[r];
                } else {
                    gemm_seg<1>(acc, (const float4*)a.wT_hh[l], lane, toff1, 4 * 64, dg + (size_t)l * GK * 64, 4 * NT);
                }
            }
            // cell backward for the 16 units of this lane
            float dgv[16][4];
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                const int tg = 4 * jq + wave;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int reg = 4 * jq + r;
                    float di = 0.0f, df = 0.0f, dgg = 0.0f, dog = 0.0f;
                    if (tg < NT && a.gru == 2) {
                        // nn.RNN: saved slot 0 is h' = tanh(pre); d pre = dh * (1 - h'^2)
                        const float hv = gsv[reg][0];
                        di = acc[0][reg] * (1.0f - hv * hv);
                    } else if (tg < NT && a.gru) {
                        // nn.GRU: saved slots (r, z, n, n_h); sv_c holds h.  dc carries dh' * z, the direct path
                        const float rg = gsv[reg][0], zg = gsv[reg][1], ng = gsv[reg][2], nh = gsv[reg][3];
                        const float hp = cpv[reg];
                        const float dh = acc[0][reg] + dc[l][reg];
                        const float dnp = dh * (1.0f - zg) * (1.0f - ng * ng);
                        di = dnp * nh * rg * (1.0f - rg);                       // d r_pre
                        df = dh * (hp - ng) * zg * (1.0f - zg);                 // d z_pre
                        dgg = dnp;                                              // d n_x
                        dog = dnp * rg;                                         // d n_h
                        dc[l][reg] = dh * zg;
                    } else if (tg < NT) {
                        const float ig = gsv[reg][0], fg = gsv[reg][1], gg = gsv[reg][2], og = gsv[reg][3];
                        const float cn = cnv[reg];
                        const float cp = cpv[reg];
                        const float th = fast_tanh(cn);
                        const float dh = acc[0][reg];
                        const float dct = dc[l][reg] + dh * og * (1.0f - th * th);
                        dog = dh * th * og * (1.0f - og);
                        di = dct * gg * ig * (1.0f - ig);
                        df = dct * cp * fg * (1.0f - fg);
                        dgg = dct * ig * (1.0f - gg * gg);
                        dc[l][reg] = dct * fg;
                    }
                    dgv[reg][0] = di; dgv[reg][1] = df; dgv[reg][2] = dgg; dgv[reg][3] = dog;
                }
            }
            __syncthreads();                                       // every wave has read dgates(l, t+1)
            float* dgl = dg + (size_t)l * GK * 64;
            float* dgo = a.dgates + (inst * SL + L0 + l) * GK * 64 + lane;
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                const int tg = 4 * jq + wave;
                if (tg < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float v = row_ok ? dgv[4 * jq + r][g] : 0.0f;
                            dgl[(16 * tg + 4 * r + g) * 64] = v;
                            dgo[(16 * tg + 4 * r + g) * 64] = v;
                        }
                }
            }
            __syncthreads();                                       // dgates(l, t) complete
        }
        if (a.nt_x > 0 && wave < a.nt_x) {
            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
            gemm_seg<1>(acc, (const float4*)a.wT_in[0], lane, toff1, a.nt_x * 64, dg, 4 * NT);
            if (a.dx_tile) {
                float* o = a.dx_tile + (inst * 64 + 16 * wave) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r * 64] = acc[0][r];
            }
            if (a.dx_rows && row_ok) {
                float* o = a.dx_rows + ((size_t)row_raw * a.T + t) * a.dx_f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = feat_of(wave, r, hh);
                    if (f < a.dx_f) o[f] = acc[0][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// lstm_bwd16_kernel: the BPTT recurrence of ONE nn.LSTM layer on 16-ROW tiles (v_mfma_f32_16x16x4_f32), for
// training batches of few row tiles (batch 64 = two 32-row tiles = two CUs in lstm_bwd_kernel, each bound by its
// matrix pipe for every one of the T sequential steps): twice the CUs, half the matrix time per step.
//   * D tile = (16 hidden units) x (16 rows): lane (n = l & 15, g4 = l >> 4) holds dh of units 16 To + 4 g4 + v
//     (v = its four D registers) of row n, runs their cell backward and so owns their 16 gate gradients;
//   * k-step (T, v, g) multiplies gate g of units 16 T + 4 k' + v (k' = 0..3): its B operand wants, in lane
//     (n, k'), the gate gradient that very lane produced as owner of output tile T -- the gate gradients go
//     through LDS as [k-step][lane], written and read without transposition or bank conflicts;
//   * wave w owns output tile w; its W_hh^T fragments (one register per k-step, 112 - 128) are gathered once from
//     the transposed blob's 32x32x2 'unit' section and stay in registers; dgates double-buffered in LDS, one
//     barrier per step; saved gates / cell states of the step are requested before the products.
// Reads and writes lstm_bwd_kernel's formats (sv_gates, sv_c, dh_top in unit-tile order, dgates).  Products are
// summed four k at a time in another order: results agree with lstm_bwd_kernel's to rounding.
// ---------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(512) lstm_bwd16_kernel(const LstmBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HP = 8 * NT;
    constexpr int NTT = (HP + 15) / 16;                            // 16-unit tiles (waves with work)
    constexpr int KS = 16 * NTT;                                   // k-steps: (unit tile T, v, gate g)
    constexpr int GK = 16 * NT;
    static_assert(NTT <= 8, "one output tile per wave");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile16 = (long)blockIdx.x;
    const int n = lane & 15, g4 = lane >> 4;
    const long row_raw = tile16 * 16 + n;
    const bool row_ok = row_raw < a.n_rows;
    const long rowtile = tile16 >> 1;
    const int m = 16 * (int)(tile16 & 1) + n;
    const int SL = a.sv_nl > 0 ? a.sv_nl : 1, L0 = a.sv_nl > 0 ? a.sv_l0 : 0;
    const bool active = wave < NTT;                                // (wave-uniform)
    const int To = active ? wave : NTT - 1;
    const int t_outs = a.T - a.t_out0;

    // this lane's four units and where their values live in the 32-row layouts
    bool uok[4];
    int gsl[4], l32[4], hsl[4], dsl[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int u0 = 16 * To + 4 * g4 + v;
        uok[v] = active && u0 < HP;
        const int u = u0 < HP ? u0 : HP - 1;                       // (clamped: addresses stay inside the instance)
        gsl[v] = 16 * (u >> 3) + 4 * ((u & 7) >> 1);               // + gate: slot of a gate value
        l32[v] = (u & 1) * 32 + m;
        hsl[v] = 4 * (u >> 3) + ((u & 7) >> 1);                    // slot of c / h
        dsl[v] = 16 * ((u >> 3) & 3) + 4 * (u >> 5) + ((u & 7) >> 1);   // slot of dh_top (unit-tile order)
    }
    // W_hh^T fragments: A[i][k'] of k-step (T, v, g) = W_hh[gate g of unit 16 T + 4 k' + v][unit 16 To + i],
    // lane = i + 16 k'; in the transposed blob (train.py transposed_map, kind 'unit', 4 output tiles) that element
    // sits at quad s >> 2, tile (uk >> 3) & 3, lane (ua & 1) * 32 + i32, component s & 3 with s the gate's slot
    float wreg[KS];
    {
        const int uk = 16 * To + n;
        const int t8 = uk >> 3;
        const int i32 = 8 * (t8 >> 2) + 4 * (uk & 1) + ((uk & 7) >> 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int T = ks >> 4, v = (ks >> 2) & 3, g = ks & 3;
            const int ua = 16 * T + 4 * g4 + v;
            const bool ok = active && ua < HP && uk < HP;
            const int uac = ua < HP ? ua : HP - 1;
            const int sslot = 16 * (uac >> 3) + 4 * ((uac & 7) >> 1) + g;
            const size_t pos = ((size_t)((sslot >> 2) * 4 + (t8 & 3)) * 64 + (uac & 1) * 32 + (i32 & 31)) * 4 + (sslot & 3);
            const float w = a.wT_hh[0][uk < HP ? pos : 0];
            wreg[ks] = ok ? w : 0.0f;
        }
    }
    float dc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float* const lbase = smem + lane;

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t inst = (size_t)(rowtile * a.T + t);
        // saved gates and cell states of the step: requested before the products, which do not depend on them
        float gsv[4][4], cnv[4], cpv[4];
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            const float* sg = a.sv_gates + (inst * SL + L0) * GK * 64;
            const float* sc = a.sv_c + (inst * SL + L0) * (4 * NT) * 64;
            const float* scp = (t > 0) ? a.sv_c + ((inst - 1) * SL + L0) * (4 * NT) * 64 : nullptr;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gsv[v][g] = sg[(size_t)(gsl[v] + g) * 64 + l32[v]];
                cnv[v] = sc[(size_t)hsl[v] * 64 + l32[v]];
                cpv[v] = scp ? scp[(size_t)hsl[v] * 64 + l32[v]] : 0.0f;
            }
            if (t >= a.t_out0 && a.dh_top) {
                const float* p = a.dh_top + (size_t)(rowtile * t_outs + (t - a.t_out0)) * 64 * 64;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = p[(size_t)dsl[v] * 64 + l32[v]];
            }
        }
        if (t + 1 < a.T && active) {
            const float* b = lbase + ((t + 1) & 1) * (KS * 64);
            f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < KS; ks += 2) {
                acc = mfma16(wreg[ks], b[ks * 64], acc);
                acc2 = mfma16(wreg[ks + 1], b[(ks + 1) * 64], acc2);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] += acc2[v];
        }
        if (active) {
            float* const lw = lbase + (t & 1) * (KS * 64) + (size_t)(To * 16) * 64;
            float* const dgo = a.dgates + (inst * SL + L0) * GK * 64;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float ig = gsv[v][0], fg = gsv[v][1], gg = gsv[v][2], og = gsv[v][3];
                const float th = fast_tanh(cnv[v]);
                const float dh = acc[v];
                const float dct = dc[v] + dh * og * (1.0f - th * th);
                const bool ok = uok[v] && row_ok;
                const float dog = ok ? dh * th * og * (1.0f - og) : 0.0f;
                const float di = ok ? dct * gg * ig * (1.0f - ig) : 0.0f;
                const float df = ok ? dct * cpv[v] * fg * (1.0f - fg) : 0.0f;
                const float dgg = ok ? dct * ig * (1.0f - gg * gg) : 0.0f;
                dc[v] = dct * fg;
                lw[(4 * v + 0) * 64] = di; lw[(4 * v + 1) * 64] = df; lw[(4 * v + 2) * 64] = dgg; lw[(4 * v + 3) * 64] = dog;
                if (uok[v]) {
                    float* o = dgo + (size_t)gsl[v] * 64 + l32[v];
                    o[0] = di; o[64] = df; o[128] = dgg; o[192] = dog;
                }
            }
        }
        __syncthreads();                                           // dgates(t) published (the other buffer is free again)
    }
}

// ---------------------------------------------------------------------------
// lstm_bwd4_kernel: lstm_bwd16_kernel's job on 4-ROW tiles with v_mfma_f32_4x4x1_16B_f32 (see lstm_rec4_kernel), for
// the smallest training batches: batch 64 becomes 16 tiles on 16 CUs instead of 4.
//   * one instruction gives dh of 64 hidden units (block b = units 4 b .. 4 b + 3, D register i) for the tile's 4 rows
//     from ONE gate row k: A lane (b, i) = W_hh[k][unit 64 Go + 4 b + i], B lane (b, j) = dgates[k][row j] in every block;
//   * the 8 waves are (output group Go = w >> 2) x (K slice s = w & 3), slice s = gate s of every unit (HP products,
//     W_hh^T rows in registers); the four partial tiles of a group meet in LDS, and thread (w, lane 4 b + j) finishes
//     ONE (unit 64 Go + 4 b + s, row j): sum of the four partials' register s, cell backward, four gate gradients,
//     which go to LDS as [gate][row][unit] -- the next step's B operands, read as broadcast float4s -- and to global
//     memory in lstm_bwd_kernel's format.  Two barriers per step.
// ---------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void lstm_bwd4_body(const LstmBwdArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HP = 8 * NT;
    constexpr int GK = 16 * NT;
    constexpr int HPAD = 144;                                      // LDS floats per (gate, row): >= HP, rows 16 banks apart
    static_assert(HP <= 128 && HP <= HPAD && HP % 4 == 0, "two output groups of 64 units");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile4 = (long)blockIdx.x;
    const int j = lane & 3, b = lane >> 2;
    const int Go = wave >> 2, s = wave & 3;
    const long row_raw = tile4 * 4 + j;
    const bool row_ok = row_raw < a.n_rows;
    const long rowtile = tile4 >> 3;
    const int m = 4 * (int)(tile4 & 7) + j;
    const int SL = a.sv_nl > 0 ? a.sv_nl : 1, L0 = a.sv_nl > 0 ? a.sv_l0 : 0;
    const int t_outs = a.T - a.t_out0;
#ifdef VSV_STAMPS
    if (NT == 16 && lane == 0 && wave < 4 && blockIdx.x < kStampWgs)
        g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 6] = __builtin_amdgcn_s_memtime();
#endif
    float* const dgl = smem;                                       // [gate 4][row 4][HPAD]
    float* const part = smem + 4 * 4 * HPAD;                       // [wave 8][register 4][lane 64]

    // the (unit, row) this thread finishes, and where its values live in the 32-row layouts
    const int u0 = 64 * Go + 4 * b + s;
    const bool uok = u0 < HP;
    const int u = uok ? u0 : HP - 1;                               // (clamped: addresses stay inside the instance)
    const int gsl = 16 * (u >> 3) + 4 * ((u & 7) >> 1);            // + gate: slot of a gate value
    const int l32 = (u & 1) * 32 + m;
    const int hsl = 4 * (u >> 3) + ((u & 7) >> 1);                 // slot of c
    const int dsl = 16 * ((u >> 3) & 3) + 4 * (u >> 5) + ((u & 7) >> 1);   // slot of dh_top (unit-tile order)

    // W_hh[gate s of unit ua][unit uk = 64 Go + 4 b + (lane & 3)] for every ua, from the transposed blob
    // (train.py transposed_map, kind 'unit': see lstm_bwd16_kernel)
    float wreg[HP];
    {
        const int uk = 64 * Go + 4 * b + (lane & 3);
        const bool kok = uk < HP;
        const int ukc = kok ? uk : HP - 1;
        const int t8 = ukc >> 3;
        const int i32 = 8 * (t8 >> 2) + 4 * (ukc & 1) + ((ukc & 7) >> 1);
#pragma unroll
        for (int ua = 0; ua < HP; ++ua) {
            const int sslot = 16 * (ua >> 3) + 4 * ((ua & 7) >> 1) + s;
            const size_t pos = ((size_t)((sslot >> 2) * 4 + (t8 & 3)) * 64 + (ua & 1) * 32 + i32) * 4 + (sslot & 3);
            const float w = a.wT_hh[0][pos];
            wreg[ua] = kok ? w : 0.0f;
        }
    }
    float dc = 0.0f;
    const float* const hr = dgl + (size_t)(s * 4 + j) * HPAD;      // this wave's K slice (gate s), this lane's row

    // saved gates and cell states of a step, and the top gradient: fetched ONE STEP AHEAD (a step is ~3 us of dependent
    // work; requested at its own top, the L2 round trip of these seven loads was most of a microsecond of it)
    float gsv_n[4], cn_n, cp_n, dht_n;
    auto fetch = [&](int t) {
        const size_t inst = (size_t)(rowtile * a.T + t);
        const float* sg = a.sv_gates + ((inst * SL + L0) * GK + gsl) * 64 + l32;
#pragma unroll
        for (int g = 0; g < 4; ++g) gsv_n[g] = sg[(size_t)g * 64];
        cn_n = a.sv_c[((inst * SL + L0) * (4 * NT) + hsl) * 64 + l32];
        cp_n = (t > 0) ? a.sv_c[(((inst - 1) * SL + L0) * (4 * NT) + hsl) * 64 + l32] : 0.0f;
        dht_n = 0.0f;
        if (t >= a.t_out0 && a.dh_top)
            dht_n = a.dh_top[((size_t)(rowtile * t_outs + (t - a.t_out0)) * 64 + dsl) * 64 + l32];
    };
    fetch(a.T - 1);
#ifdef VSV_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    if (NT == 16 && lane == 0 && wave < 4 && blockIdx.x < kStampWgs)
        g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 7] = __builtin_amdgcn_s_memtime() + (unsigned long long)(wreg[0] == 12345.0f);
    __builtin_amdgcn_sched_barrier(0);
#endif
    for (int t = a.T - 1; t >= 0; --t) {
        const size_t inst = (size_t)(rowtile * a.T + t);
        float gsv[4], cn = cn_n, cp = cp_n, dht = dht_n;
#pragma unroll
        for (int g = 0; g < 4; ++g) gsv[g] = gsv_n[g];
        BWD_STAMP(0);
        if (t > 0) fetch(t - 1);
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (t + 1 < a.T) {
            f32x4 a1 = acc, a2 = acc, a3 = acc;
#pragma unroll
            for (int k = 0; k < HP; k += 4) {
                const float4 gv = *(const float4*)(hr + k);
                acc = mfma4(wreg[k + 0], gv.x, acc);
                a1 = mfma4(wreg[k + 1], gv.y, a1);
                a2 = mfma4(wreg[k + 2], gv.z, a2);
                a3 = mfma4(wreg[k + 3], gv.w, a3);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] += (a1[v] + a2[v]) + a3[v];
        }
        BWD_STAMP(1);
        {
            float* o = part + (size_t)(wave * 4) * 64 + lane;
            o[0] = acc[0]; o[64] = acc[1]; o[128] = acc[2]; o[192] = acc[3];
        }
        BWD_STAMP(2);
        __syncthreads();                                           // partial tiles published; dgates(t+1) read by all
        BWD_STAMP(3);
        float dh = dht;
        {
            const float* pp = part + (size_t)(4 * Go * 4 + s) * 64 + lane;      // register s of the group's four waves
            dh += (pp[0] + pp[4 * 64]) + (pp[8 * 64] + pp[12 * 64]);
        }
        {
            const float ig = gsv[0], fg = gsv[1], gg = gsv[2], og = gsv[3];
            const float th = fast_tanh(cn);
            const float dct = dc + dh * og * (1.0f - th * th);
            const bool ok = uok && row_ok;
            const float dog = ok ? dh * th * og * (1.0f - og) : 0.0f;
            const float di = ok ? dct * gg * ig * (1.0f - ig) : 0.0f;
            const float df = ok ? dct * cp * fg * (1.0f - fg) : 0.0f;
            const float dgg = ok ? dct * ig * (1.0f - gg * gg) : 0.0f;
            dc = dct * fg;
            if (uok) {
                float* lw = dgl + (size_t)j * HPAD + u0;
                lw[0] = di; lw[4 * HPAD] = df; lw[8 * HPAD] = dgg; lw[12 * HPAD] = dog;
                float* o = a.dgates + ((inst * SL + L0) * GK + gsl) * 64 + l32;
                o[0] = di; o[64] = df; o[128] = dgg; o[192] = dog;
            }
        }
        BWD_STAMP(4);
        __syncthreads();                                           // dgates(t) published
        BWD_STAMP(5);
    }
#ifdef VSV_STAMPS
    if (NT == 16 && lane == 0 && wave < 4 && blockIdx.x < kStampWgs)
        g_stamps[((size_t)blockIdx.x * 4 + wave) * 32 + 8] = __builtin_amdgcn_s_memtime();
#endif
}
template <int NT>
__global__ void __launch_bounds__(512) lstm_bwd4_kernel(const LstmBwdArgs a) { lstm_bwd4_body<NT>(a); }
template <int NT>
__global__ void __launch_bounds__(512) lstm_bwd4_pair_kernel(const LstmBwdArgs a0, const LstmBwdArgs a1) {
    if (blockIdx.y == 0) {
        if ((long)blockIdx.x < 8 * a0.n_rowtiles) lstm_bwd4_body<NT>(a0);
    } else {
        if ((long)blockIdx.x < 8 * a1.n_rowtiles) lstm_bwd4_body<NT>(a1);
    }
}

// ---------------------------------------------------------------------------
// tileb_gemm_kernel: Y(inst) = W^T-fragments . G(inst) for every (row tile, t) instance at once -- the
// parts of BPTT that do not sit on the recurrence (d h_{l-1}(t) = W_in(l)^T . dgates(l, t) and
// dx(t) = W_in(0)^T . dgates(0, t)), taken out of lstm_bwd_kernel's time loop so that they run on as
// many CUs as there are instances.  One workgroup per instance, wave w = output tile w.
// ---------------------------------------------------------------------------
typedef vsv_tileb_gemm TilebGemmArgs;

// SPLIT (few instances: a training batch of 64 at T = 11 is 22): one workgroup per (instance, output tile), its four
// waves take a quarter of K each and wave 0 adds the partial tiles in a fixed order -- the dependent chain of 16 NT
// products per tile, 7 us of one CU's matrix pipe, becomes four chains of 4 NT on four SIMDs.
template <bool SPLIT>
__device__ __forceinline__ void tileb_gemm_body(const TilebGemmArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t inst = SPLIT ? blockIdx.x / (unsigned)a.ntT : blockIdx.x;
    const int tile = SPLIT ? (int)(blockIdx.x - inst * (unsigned)a.ntT) : wave;
    // the instance's G (4 nqT k-steps x 64 lanes, contiguous) goes through LDS: every float4 of it is requested at
    // once by the whole workgroup, one L2 round trip instead of one per K quad in front of the quad's products
    {
        const float4* src = (const float4*)(a.g + inst * (size_t)a.g_stride);
        const int n4 = a.nqT * 64;                                 // <= 64 quads (checked by the launcher): 16 per thread
        // (straight-line code on purpose: inside a loop over e0 the sixteen registers stayed an array in scratch
        // memory -- load, wait, scratch store sixteen times over, then scratch loads in front of the LDS writes)
        float4 tmp[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = u * 256 + (int)threadIdx.x;
            tmp[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (e < n4) tmp[u] = src[e];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = u * 256 + (int)threadIdx.x;
            if (e < n4) ((float4*)smem)[e] = tmp[u];
        }
    }
    __syncthreads();
    if (!SPLIT && wave >= a.ntT) return;
    const long rowtile = (long)(inst / (size_t)a.T);
    const int t = (int)(inst - (size_t)rowtile * a.T);
    const int hh = lane >> 5, m = lane & 31;
    const long row_raw = rowtile * 32 + m;
    const bool row_ok = row_raw < a.n_rows;
    const int toff1[1] = {tile * 64};
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.0f;
    if (SPLIT) {
        const int q0 = (wave * a.nqT) >> 2, q1 = ((wave + 1) * a.nqT) >> 2;
        if (q1 > q0)
            gemm_seg<1>(acc, (const float4*)a.wT + (size_t)q0 * a.ntT * 64, lane, toff1, a.ntT * 64,
                        smem + (size_t)q0 * 4 * 64 + lane, q1 - q0);
        __syncthreads();                                           // every wave has read its part of G
        if (wave > 0) {
            float* o = smem + (size_t)(wave - 1) * 16 * 64 + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r * 64] = acc[0][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += smem[(size_t)(k * 16 + r) * 64 + lane];
    } else {
        gemm_seg<1>(acc, (const float4*)a.wT, lane, toff1, a.ntT * 64, smem + lane, a.nqT);
    }
    if (a.drop_p > 0.0f) {
        // the consumer read the dropped copy of these units: same Philox factors as the forward
        const DropArgs da = resolve_drop(DropArgs{a.drop_p, 1.0f / (1.0f - a.drop_p), (uint32_t)a.drop_seed,
                                                  (uint32_t)(a.drop_seed >> 32), (uint32_t)a.drop_id,
                                                  (const uint32_t*)a.drop_seed_dev});
        const uint32_t rowc = (uint32_t)(row_ok ? row_raw : a.n_rows - 1);
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            float fct[4];
            dropout_factors(da, rowc, (uint32_t)t, 4 * jq + tile, hh, a.drop_layer, fct);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][4 * jq + r] *= fct[r];
        }
    }
    if (a.out_tile) {
        float* o = a.out_tile + (inst * 64 + 16 * tile) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r * 64] = acc[0][r];
    }
    if (a.out_rows && row_ok) {
        float* o = a.out_rows + ((size_t)row_raw * a.T + t) * a.out_f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = feat_of(tile, r, hh);
            if (f < a.out_f) o[f] = acc[0][r];
        }
    }
}
template <bool SPLIT>
__global__ void __launch_bounds__(256) tileb_gemm_kernel(const TilebGemmArgs a) { tileb_gemm_body<SPLIT>(a); }
// two SPLIT launches as one (blockIdx.y picks the argument set; see lstm_rec4_pair_kernel)
__global__ void __launch_bounds__(256) tileb_gemm_pair_kernel(const TilebGemmArgs a0, const TilebGemmArgs a1) {
    if (blockIdx.y == 0) {
        if ((long)blockIdx.x < a0.n_rowtiles * a0.T * a0.ntT) tileb_gemm_body<true>(a0);
    } else {
        if ((long)blockIdx.x < a1.n_rowtiles * a1.T * a1.ntT) tileb_gemm_body<true>(a1);
    }
}

// ---------------------------------------------------------------------------
// loss_grad_kernel: the loss terms of train_step (pinball over the quantile
// channels, ex-feature MSE/L1, KL over all T positions) and their gradients.
// sums[0..2] (double, zeroed by the caller) receive the three MEANS.
// ---------------------------------------------------------------------------
typedef vsv_loss LossArgs;

__global__ void __launch_bounds__(256) loss_grad_kernel(const LossArgs a) {
    __shared__ double red[3][256];
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // pred / d_pred hold Hs >= H steps per row; the loss is over the last H of them (the gradient of the
    // others is written as zero)
    const int Hs = a.Hs > 0 ? a.Hs : a.H, h_skip = Hs - a.H;
    const double gs = a.grad_scale != 0.0f ? (double)a.grad_scale : 1.0;
    // the two loss weights from device memory when given (a captured graph then follows a KL-annealing schedule
    // without being captured again), else from the argument struct
    const float ex_weight = a.weights_dev ? a.weights_dev[0] : a.ex_weight;
    const float kl_weight = a.weights_dev ? a.weights_dev[1] : a.kl_weight;
    const long n_s = (long)a.B * Hs * a.Q * 25;
    const double inv_s = 1.0 / ((double)a.B * a.H * 25);
    for (long i = (long)blockIdx.x * 256 + tid; i < n_s; i += (long)gridDim.x * 256) {
        const int p = (int)(i % 25);
        const int qi = (int)((i / 25) % a.Q);
        const long bh = i / (25L * a.Q);
        const int h = (int)(bh % Hs) - h_skip;
        const long b = bh / Hs;
        if (h < 0) { a.d_pred[i] = 0.0f; continue; }
        const float tg = a.target[((size_t)b * a.T + (a.T - a.H + h)) * 25 + p];
        const float e = tg - a.pred[i];
        if (a.surf_l2) {
            // F.mse_loss of the one-channel legacy decoder (cvae_with_mem.py:681): mean over B*H*25 (Q = 1)
            s0 += (double)e * e * inv_s;
            a.d_pred[i] = (float)(-2.0 * e * inv_s * gs);
            continue;
        }
        const float qq = a.q[qi];
        const float lo = (qq - 1.0f) * e, hi = qq * e;
        s0 += (double)a.qw[qi] * (double)fmaxf(lo, hi) * inv_s;
        const float d = (e > 0.0f) ? -qq : ((e < 0.0f) ? (1.0f - qq) : (0.5f - qq));
        a.d_pred[i] = (float)((double)a.qw[qi] * d * inv_s * gs);
    }
    if (a.ex_pred) {
        const long n_e = (long)a.B * Hs * a.X;
        const double inv_e = 1.0 / ((double)a.B * a.H * a.ex_n);
        for (long i = (long)blockIdx.x * 256 + tid; i < n_e; i += (long)gridDim.x * 256) {
            const int f = (int)(i % a.X);
            const long bh = i / a.X;
            const int h = (int)(bh % Hs) - h_skip;
            const long b = bh / Hs;
            float g = 0.0f;
            if (h >= 0 && f < a.ex_n) {
                const float d = a.ex_pred[i] - a.ex_target[((size_t)b * a.T + (a.T - a.H + h)) * a.X + f];
                if (a.ex_l2) {
                    s1 += (double)d * d * inv_e;
                    g = (float)(2.0 * d * inv_e * ex_weight * gs);
                } else {
                    s1 += (double)fabsf(d) * inv_e;
                    g = (float)(((d > 0.0f) - (d < 0.0f)) * inv_e * ex_weight * gs);
                }
            }
            a.d_ex[i] = g;
        }
    }
    const long n_z = (long)a.B * a.T * a.L;
    const double inv_z = 1.0 / ((double)a.B * a.T);
    for (long i = (long)blockIdx.x * 256 + tid; i < n_z; i += (long)gridDim.x * 256) {
        const float mu = a.zm[i], lv = a.zl[i];
        const float ev = expf(lv);
        s2 += -0.5 * (1.0 + (double)lv - (double)ev - (double)mu * mu) * inv_z;
        a.d_zm[i] = (float)((double)mu * inv_z * kl_weight * gs);
        a.d_zl[i] = (float)(-0.5 * (1.0 - (double)ev) * inv_z * kl_weight * gs);
    }
    red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { red[0][tid] += red[0][tid + k]; red[1][tid] += red[1][tid + k]; red[2][tid] += red[2][tid + k]; }
        __syncthreads();
    }
    if (tid < 3) atomicAdd(&a.sums[tid], red[tid][0]);
    if (a.loss_vec) {
        // the last workgroup to finish mixes the three sums into the five numbers a step reports and leaves sums and
        // the ticket at zero for the next call (vsv_loss::mix)
        __shared__ unsigned last;
        __threadfence();
        __syncthreads();
        if (tid == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (last && tid == 0) {
            __threadfence();
            const double s0 = atomicAdd(&a.sums[0], 0.0), s1 = atomicAdd(&a.sums[1], 0.0), s2 = atomicAdd(&a.sums[2], 0.0);
            for (int r = 0; r < 5; ++r) a.loss_vec[r] = a.mix[3 * r + 0] * s0 + a.mix[3 * r + 1] * s1 + a.mix[3 * r + 2] * s2;
            a.sums[0] = 0.0; a.sums[1] = 0.0; a.sums[2] = 0.0;
            __threadfence();
            atomicExch(a.ticket, 0u);
        }
    }
}

// ---------------------------------------------------------------------------
// percentile: one workgroup per vector of S MC samples; bitonic sort in LDS,
// then numpy's 'linear' interpolation.
// ---------------------------------------------------------------------------
struct PercArgs {
    int nq;
    int lo[8];
    float fr[8];
};

__global__ void __launch_bounds__(256) percentile_kernel(const float* __restrict__ in, long n_vec, int S, int P2,
                                                         PercArgs pa, int inner, long group_stride,
                                                         float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int tid = threadIdx.x;
    for (long v = blockIdx.x; v < n_vec; v += gridDim.x) {
        const float* src = in + ((size_t)(v / inner) * group_stride + (size_t)(v % inner)) * S;
        for (int i = tid; i < P2; i += 256) buf[i] = (i < S) ? src[i] : INFINITY;
        for (int k = 2; k <= P2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                __syncthreads();
                for (int i = tid; i < P2; i += 256) {
                    const int p = i ^ j;
                    if (p > i) {
                        const float x = buf[i], y = buf[p];
                        const bool asc = (i & k) == 0;
                        if ((x > y) == asc) { buf[i] = y; buf[p] = x; }
                    }
                }
            }
        }
        __syncthreads();
        if (tid < pa.nq) {
            const int lo = pa.lo[tid];
            const int hi = (lo + 1 < S) ? lo + 1 : S - 1;
            const float x = buf[lo], y = buf[hi], d = y - x, fr = pa.fr[tid];
            const float r = (fr >= 0.5f) ? y - d * (1.0f - fr) : x + d * fr;
            out[(size_t)(v / inner) * pa.nq * inner + (size_t)tid * inner + (v % inner)] = r;
        }
        __syncthreads();
    }
}

// Wavefront-level selection for S <= 1024: one wave sorts one vector entirely in registers (lane holds
// the E consecutive elements lane*E .. lane*E+E-1 of the padded vector) with a bitonic network --
// compare-exchanges at distance < E are register-to-register, the others one cross-lane exchange per
// element -- no LDS traffic and no barriers; then the numpy 'linear' interpolation.
template <int E>
__global__ void __launch_bounds__(256) percentile_wave_kernel(const float* __restrict__ in, long n_vec, int S, PercArgs pa,
                                                              int inner, long group_stride, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long v = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_vec) return;
    const float* src = in + ((size_t)(v / inner) * group_stride + (size_t)(v % inner)) * S;
    float x[E];
    if (E >= 4 && (S & 3) == 0) {
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const int i = lane * E + e;
            float4 q = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
            if (i < S) q = *(const float4*)(src + i);
            x[e] = q.x; x[e + 1] = q.y; x[e + 2] = q.z; x[e + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = lane * E + e;
            x[e] = (i < S) ? src[i] : INFINITY;
        }
    }
#pragma unroll
    for (int k = 2; k <= 64 * E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            // min(a, b) = med3(a, b, -inf), max(a, b) = med3(a, b, +inf): where the direction depends on the lane, the
            // third operand carries it (one register per stage) and a compare-exchange half is ONE instruction instead of
            // min + max + select
            if (j < E && k < E) {
                // both elements in this lane; ascending block iff bit k of the element index is clear
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) == 0) {
                        const float lo = fminf(x[e], x[e | j]), hi = fmaxf(x[e], x[e | j]);
                        const bool asc = (e & k) == 0;
                        x[e] = asc ? lo : hi;
                        x[e | j] = asc ? hi : lo;
                    }
                }
            } else if (j < E) {
                const bool asc = (lane & (k / E)) == 0;
                const float la = asc ? -INFINITY : INFINITY, lb = -la;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) == 0) {
                        const float p = x[e], q = x[e | j];
                        x[e] = __builtin_amdgcn_fmed3f(p, q, la);
                        x[e | j] = __builtin_amdgcn_fmed3f(p, q, lb);
                    }
                }
            } else {
                const int lm = j / E;                            // partner lane = lane ^ lm, same register
                const bool asc = (k >= 64 * E) ? true : ((lane & (k / E)) == 0);
                const bool keep_min = asc == ((lane & lm) == 0);
                const float lk = keep_min ? -INFINITY : INFINITY;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float o;
                    // neighbours inside a quad of lanes come through the DPP crossbar (no LDS round trip)
                    if (lm == 1) o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[e]), 0xB1, 0xF, 0xF, false));
                    else if (lm == 2) o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[e]), 0x4E, 0xF, 0xF, false));
                    else o = __shfl_xor(x[e], lm, 64);
                    x[e] = __builtin_amdgcn_fmed3f(x[e], o, lk);
                }
            }
        }
    }
    // element i of the sorted vector is register i % E of lane i / E
    float res = 0.0f;
    for (int k = 0; k < pa.nq; ++k) {
        const int lo = pa.lo[k];
        const int hi = (lo + 1 < S) ? lo + 1 : S - 1;
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if ((lo % E) == e) a = x[e];
            if ((hi % E) == e) b = x[e];
        }
        a = __shfl(a, lo / E, 64);
        b = __shfl(b, hi / E, 64);
        const float d = b - a, fr = pa.fr[k];
        const float r = (fr >= 0.5f) ? b - d * (1.0f - fr) : a + d * fr;
        if (lane == k) res = r;
    }
    if (lane < pa.nq) out[(size_t)(v / inner) * pa.nq * inner + (size_t)lane * inner + (v % inner)] = res;
}

// Workgroup-level selection for 1024 < S <= 16384 (the autoregressive stress case reduces 10 000 rollouts): the
// same bitonic network as percentile_wave_kernel over P2 = 2048 W elements (W = 1, 2, 4 or 8 waves of one
// workgroup, 32 registers per lane; element index = (wave * 64 + lane) * 32 + register).  Compare-exchanges at
// distance < 32 are register-to-register, below 2048 one cross-lane exchange (DPP / ds_bpermute), and only the
// 1 + 2 + 3 stages at distance >= 2048 go through LDS (partner = same lane and register of wave w ^ d): 6 of the 105
// stages instead of all of them (percentile_kernel: 105 stages of LDS traffic with a barrier each).  A
// compare-exchange half is one v_med3_f32 whose third operand (-inf / +inf) carries the direction.
constexpr int kPwgE = 32;
__global__ void __launch_bounds__(512) percentile_wg_kernel(const float* __restrict__ in, long n_vec, int S, int W, PercArgs pa,
                                                            int inner, long group_stride, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float buf[];   // [wave][register][lane]: W * 2048 floats
    constexpr int E = kPwgE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long v = (long)blockIdx.x;
    const float* src = in + ((size_t)(v / inner) * group_stride + (size_t)(v % inner)) * S;
    const int i0 = (wave * 64 + lane) * E;
    float x[E];
    if ((S & 3) == 0) {
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            float4 q = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
            if (i0 + e < S) q = *(const float4*)(src + i0 + e);
            x[e] = q.x; x[e + 1] = q.y; x[e + 2] = q.z; x[e + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = (i0 + e < S) ? src[i0 + e] : INFINITY;
    }
    const int P2 = 64 * E * W;
    float* const mine = buf + (size_t)wave * (E * 64) + lane;
#pragma unroll
    for (int k = 2; k <= 64 * E * 8; k <<= 1) {
        if (k <= P2) {                                             // (uniform)
            // ascending block iff bit k of the element index is clear; the last merge (k == P2) is ascending everywhere
            const bool asc = (k == P2) ? true
                           : (k < E) ? true /* per register, below */
                           : (k < 64 * E) ? ((lane & (k / E)) == 0) : ((wave & (k / (64 * E))) == 0);
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64 * E) {
                    // partner: the same lane and register of wave ^ (j / 2048), through LDS
                    const int wm = j / (64 * E);
                    const bool keep_min = asc == ((wave & wm) == 0);
                    const float lk = keep_min ? -INFINITY : INFINITY;
                    __syncthreads();                               // (the previous exchange's readers are done)
#pragma unroll
                    for (int e = 0; e < E; ++e) mine[e * 64] = x[e];
                    __syncthreads();
                    const float* theirs = buf + (size_t)(wave ^ wm) * (E * 64) + lane;
#pragma unroll
                    for (int e = 0; e < E; ++e) x[e] = __builtin_amdgcn_fmed3f(x[e], theirs[e * 64], lk);
                } else if (j >= E) {
                    const int lm = j / E;                          // partner lane = lane ^ lm, same register
                    const bool keep_min = asc == ((lane & lm) == 0);
                    const float lk = keep_min ? -INFINITY : INFINITY;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        float o;
                        if (lm == 1) o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[e]), 0xB1, 0xF, 0xF, false));
                        else if (lm == 2) o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[e]), 0x4E, 0xF, 0xF, false));
                        else o = __shfl_xor(x[e], lm, 64);
                        x[e] = __builtin_amdgcn_fmed3f(x[e], o, lk);
                    }
                } else if (k < E) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & j) == 0) {
                            const float lo = fminf(x[e], x[e | j]), hi = fmaxf(x[e], x[e | j]);
                            const bool up = (k == P2) || (e & k) == 0;
                            x[e] = up ? lo : hi;
                            x[e | j] = up ? hi : lo;
                        }
                    }
                } else {
                    const float la = asc ? -INFINITY : INFINITY, lb = -la;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & j) == 0) {
                            const float p = x[e], q = x[e | j];
                            x[e] = __builtin_amdgcn_fmed3f(p, q, la);
                            x[e | j] = __builtin_amdgcn_fmed3f(p, q, lb);
                        }
                    }
                }
            }
        }
    }
    // sorted element i lives in wave i / 2048, lane (i / 32) % 64, register i % 32: everything goes to LDS once and
    // thread k picks the two order statistics of percentile k
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) mine[e * 64] = x[e];
    __syncthreads();
    if ((int)threadIdx.x < pa.nq) {
        const int k = threadIdx.x;
        const int lo = pa.lo[k];
        const int hi = (lo + 1 < S) ? lo + 1 : S - 1;
        auto at = [&](int i) { return buf[(size_t)(i / (64 * E)) * (E * 64) + (size_t)(i % E) * 64 + (i / E) % 64]; };
        const float a = at(lo), b = at(hi), d = b - a, fr = pa.fr[k];
        const float r = (fr >= 0.5f) ? b - d * (1.0f - fr) : a + d * fr;
        out[(size_t)(v / inner) * pa.nq * inner + (size_t)k * inner + (v % inner)] = r;
    }
}

__global__ void __launch_bounds__(256) philox_normal_kernel(uint64_t seed, long day0, long S, int step0, int n_steps,
                                                            int L, long n_total, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int f = (int)(i % L);
    const long rs = i / L;
    const int step = (int)(rs % n_steps);
    const long row = rs / n_steps;
    out[i] = philox_normal_elem((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)(row % S),
                                (uint32_t)(day0 + row / S), (uint32_t)(step0 + step), f);
}

// ---------------------------------------------------------------------------
// gmm_sample: latents from a Gaussian mixture fitted to the aggregated posterior
// (CVAEMemRand._sample_from_fitted_prior, cvae_with_mem_randomized.py:605-640): component ~
// multinomial(weights), z = mean[c] + chol[c] . eps.  One thread per sample.  comp / eps are injected
// (parity tests) or drawn from Philox4x32-10 keyed by (seed; sample index, block): block 0 gives the
// uniform that picks the component through the cumulative weights, blocks 1.. give four normals each.
// ---------------------------------------------------------------------------
constexpr int kGmmMaxL = 16;

__global__ void __launch_bounds__(256) gmm_sample_kernel(const float* __restrict__ means, const float* __restrict__ chol,
                                                         const float* __restrict__ cdf, int K, int L, long n,
                                                         const int* __restrict__ comp_in, const float* __restrict__ eps_in,
                                                         uint64_t seed, float* __restrict__ z, int* __restrict__ comp_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    int c;
    if (comp_in) {
        c = comp_in[i];
        c = c < 0 ? 0 : (c >= K ? K - 1 : c);
    } else {
        uint32_t c0 = (uint32_t)i, c1 = (uint32_t)((uint64_t)i >> 32), c2 = 0u, c3 = 0x474D4D00u;
        philox4x32_10(k0, k1, c0, c1, c2, c3);
        const float u = ((float)(c0 >> 8) + 0.5f) * 5.9604644775390625e-08f;          // (0, 1)
        c = 0;
        while (c < K - 1 && u >= cdf[c]) ++c;
    }
    float e[kGmmMaxL];
#pragma unroll
    for (int j = 0; j < kGmmMaxL; ++j) e[j] = 0.0f;
    if (eps_in) {
#pragma unroll
        for (int j = 0; j < kGmmMaxL; ++j)
            if (j < L) e[j] = eps_in[i * L + j];
    } else {
#pragma unroll
        for (int b = 0; b < kGmmMaxL / 4; ++b) {
            if (4 * b < L) {
                uint32_t c0 = (uint32_t)i, c1 = (uint32_t)((uint64_t)i >> 32), c2 = (uint32_t)(b + 1), c3 = 0x474D4D00u;
                philox4x32_10(k0, k1, c0, c1, c2, c3);
                const uint32_t x[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float u1 = ((float)(x[2 * h] >> 9) + 0.5f) * 1.1920928955078125e-07f;
                    const float u2 = ((float)(x[2 * h + 1] >> 9) + 0.5f) * 1.1920928955078125e-07f;
                    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
                    e[4 * b + 2 * h] = r * __builtin_amdgcn_cosf(u2);
                    e[4 * b + 2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
                }
            }
        }
    }
    const float* m = means + (size_t)c * L;
    const float* ch = chol + (size_t)c * L * L;
#pragma unroll
    for (int r = 0; r < kGmmMaxL; ++r) {
        if (r < L) {
            float acc = m[r];
#pragma unroll
            for (int j = 0; j < kGmmMaxL; ++j)
                if (j <= r) acc = fmaf(ch[r * L + j], e[j], acc);
            z[i * L + r] = acc;
        }
    }
    if (comp_out) comp_out[i] = c;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int launch_check(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) {
        set_error("kernel needs %zu bytes of LDS (> 160 KiB)", bytes);
        return -3;
    }
    if (bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) {
            set_error("hipFuncSetAttribute(%zu): %s", bytes, hipGetErrorString(e));
            return (int)e;
        }
    }
    return 0;
}

static int launch_dense_chain(const DenseChainArgs& a, hipStream_t st) {
    const long insts = a.n_rowtiles * a.T;
    if (insts <= 0) return 0;
    for (int i = 0; i < a.n_layers; ++i) {
        if (a.layer[i].nt < 1 || a.layer[i].nt > kHeadTiles || a.layer[i].nq > 4 * kHeadTiles) {
            set_error("dense_chain: layer %d has an unsupported shape (nt %d, nq %d)", i, a.layer[i].nt, a.layer[i].nq);
            return -4;
        }
    }
    if (a.in_ks > kDenseSlots) { set_error("dense_chain: input wider than %d k-steps", kDenseSlots); return -4; }
    // (the kernels clamp indices into these ranges instead of branching around loads)
    if (a.in_ks < 1 || (a.in_mode == 0 && a.in_f < 1) || (a.inj != nullptr && a.inj_n < 1)) {
        set_error("dense_chain: empty input (in_ks %d, in_f %d) or injection range (%d)", a.in_ks, a.in_f, a.inj_n); return -1;
    }
    const size_t lds = (size_t)4 * kDenseSlots * 64 * sizeof(float);
    const long blocks = (insts + 3) / 4;
    if (blocks > 0x7fffffffL) { set_error("dense_chain: too many tiles"); return -2; }
    static const bool wg_off = [] { const char* e = getenv("VSV_DENSE_WG"); return e && e[0] == '0'; }();
    if (insts <= 512 && !wg_off) {
        // few instances: a workgroup per instance, its four waves splitting the output tiles
        const size_t lds1 = (size_t)kDenseSlots * 64 * sizeof(float);
        hipLaunchKernelGGL(dense_chain_wg_kernel, dim3((unsigned)insts), dim3(256), lds1, st, a);
        return launch_check("dense_chain_wg_kernel");
    }
    hipLaunchKernelGGL(dense_chain_kernel, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return launch_check("dense_chain_kernel");
}

// row tiles at or below this count run one wave per gate tile (NT waves per workgroup)
constexpr long kWideMaxRowtiles = 256;

template <int NT, int NL, int XMODE, bool HEAD, bool SAVE, int OWN>
static int launch_lstm_o(const LstmSeqArgs& a, hipStream_t st) {
    const int slots = (XMODE == 0 ? 0 : a.kx) + NL * 4 * NT + (HEAD ? 2 * kDenseSlots : 0) + (SAVE ? 4 * NT : 0);
    const size_t lds = (size_t)slots * 64 * sizeof(float);
    int rc = set_lds(lstm_kernel<NT, NL, XMODE, HEAD, SAVE, OWN>, lds);
    if (rc) return rc;
    if (a.n_rowtiles > 0x7fffffffL) { set_error("lstm_kernel: too many rows"); return -2; }
    hipLaunchKernelGGL((lstm_kernel<NT, NL, XMODE, HEAD, SAVE, OWN>), dim3((unsigned)a.n_rowtiles),
                       dim3(OWN == 4 ? 256 : 64 * NT), lds, st, a);
    return launch_check("lstm_kernel");
}

template <int NT, int NL, int XMODE, bool HEAD, bool SAVE = false>
static int launch_lstm_t(const LstmSeqArgs& a, hipStream_t st) {
    if (!HEAD && a.n_rowtiles <= kWideMaxRowtiles) return launch_lstm_o<NT, NL, XMODE, false, SAVE, 1>(a, st);
    return launch_lstm_o<NT, NL, XMODE, HEAD, SAVE, 4>(a, st);
}

template <int NT, int XMODE>
static int launch_lstm_nt(const LstmSeqArgs& a, int NL, bool head, hipStream_t st) {
    if (a.sv_gates)
        return NL == 2 ? launch_lstm_t<NT, 2, XMODE, false, true>(a, st) : launch_lstm_t<NT, 1, XMODE, false, true>(a, st);
    if (XMODE == 1 && head)
        return NL == 2 ? launch_lstm_t<NT, 2, XMODE, (XMODE == 1)>(a, st) : launch_lstm_t<NT, 1, XMODE, (XMODE == 1)>(a, st);
    return NL == 2 ? launch_lstm_t<NT, 2, XMODE, false>(a, st) : launch_lstm_t<NT, 1, XMODE, false>(a, st);
}

template <int NT, bool SAVE>
static int launch_lstm_rec(const LstmSeqArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 4 * NT * 64 * sizeof(float);
    int rc = set_lds(lstm_rec_kernel<NT, SAVE>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((lstm_rec_kernel<NT, SAVE>), dim3((unsigned)a.n_rowtiles), dim3(64 * NT), lds, st, a);
    return launch_check("lstm_rec_kernel");
}

// the recurrent-only single-layer launches (pre-activations given) of few row tiles: weights in registers
static bool lstm_rec_applies(const LstmSeqArgs& a, int NL) {
    static const bool off = [] { const char* e = getenv("VSV_LSTM_REC"); return e && e[0] == '0'; }();
    if (off || NL != 1 || a.n_rowtiles > kWideMaxRowtiles || a.din.gate_in) return false;
    if (a.sv_gates) return a.rec != 0 && (a.st_nl == 0);
    // inference (pre_t / pre_g): measured slower than lstm_kernel's per-layer launch on the date-range prefix
    // (2.8 vs 2.1 ms at 500 days), so it is opt-in there (VSV_LSTM_REC=2)
    static const bool infer = [] { const char* e = getenv("VSV_LSTM_REC"); return e && e[0] == '2'; }();
    return infer && (a.pre_t != nullptr || a.pre_g != nullptr);
}

// the 16-row-tile recurrent launch: one nn.LSTM layer, pre-activations given (inference: pre_t / pre_g; training: the
// layer's slots of sv_gates), few row tiles
static bool lstm_rec16_applies(const LstmSeqArgs& a, int NL) {
    static const bool off = [] { const char* e = getenv("VSV_REC16"); return e && e[0] == '0'; }();
    if (off || NL != 1 || a.gru != 0 || a.din.gate_in || a.n_rowtiles > kWideMaxRowtiles) return false;
    if (a.sv_gates) return a.rec != 0 && a.st_nl == 0;
    return a.pre_t != nullptr || a.pre_g != nullptr;
}

// lstm_rec16_kernel / lstm_rec4_kernel address a row tile's part of every array with 32-bit byte offsets
static int rec_spans_fit(const LstmSeqArgs& a, int NT) {
    const size_t sl = a.sv_gates && a.sv_nl > 0 ? (size_t)a.sv_nl : 1;
    const bool fits = a.pre_t && !a.sv_gates
        ? ((size_t)(32 * NT + 8) * (size_t)a.pre_stride + (size_t)a.n_rows + (size_t)a.T * (size_t)a.pre_tstride) * 4 < (1ull << 31)
        : (size_t)a.T * sl * (16 * NT) * 256 < (1ull << 31);
    if (!fits || (size_t)a.T * (4 * NT) * 256 >= (1ull << 31)) {
        set_error("recurrent launch: a row tile's part of an array spans more than 31-bit offsets reach");
        return -4;
    }
    return 0;
}

template <int NT>
static int launch_lstm_rec16(const LstmSeqArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * (2 * NT + 1) * 64 * sizeof(float);
    if (int rc = rec_spans_fit(a, NT)) return rc;
    // (two 16-row tiles per 32-row tile: the odd one of a ragged tail runs on clamped rows, so that the tile-B
    // outputs the next stage reads are written whole)
    if (a.sv_gates) hipLaunchKernelGGL((lstm_rec16_kernel<NT, true>), dim3((unsigned)(2 * a.n_rowtiles)), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((lstm_rec16_kernel<NT, false>), dim3((unsigned)(2 * a.n_rowtiles)), dim3(512), lds, st, a);
    return launch_check("lstm_rec16_kernel");
}

// 4-row tiles, one or two per workgroup, while the launch stays one round of workgroups; beyond 512 tiles (2048 rows)
// lstm_rec16_kernel is the faster one again (4000 days: 2.63 ms against 2.9).  VSV_REC4=0: always lstm_rec16_kernel;
// VSV_REC4_R=1|2 forces R
template <int NT, bool SAVE, int R>
static int launch_lstm_rec4_r(const LstmSeqArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * R * 4 * 144 * sizeof(float);
    if (int rc = rec_spans_fit(a, NT)) return rc;
    hipLaunchKernelGGL((lstm_rec4_kernel<NT, SAVE, R>), dim3((unsigned)(8 * a.n_rowtiles / R)), dim3(512), lds, st, a);
    return launch_check("lstm_rec4_kernel");
}
template <int NT>
static int launch_lstm_rec4(const LstmSeqArgs& a, hipStream_t st) {
    static const int forced = [] { const char* e = getenv("VSV_REC4_R"); return e ? atoi(e) : 0; }();
    int R = 8 * a.n_rowtiles <= 128 ? 1 : 2;                      // (two stacks usually run side by side: half the CUs each)
    if (forced == 1 || forced == 2) R = forced;
    if (a.sv_gates) return R == 1 ? launch_lstm_rec4_r<NT, true, 1>(a, st) : launch_lstm_rec4_r<NT, true, 2>(a, st);
    return R == 1 ? launch_lstm_rec4_r<NT, false, 1>(a, st) : launch_lstm_rec4_r<NT, false, 2>(a, st);
}
static bool lstm_rec4_applies(const LstmSeqArgs& a) {
    static const bool off = [] { const char* e = getenv("VSV_REC4"); return e && e[0] == '0'; }();
    return !off && 8 * a.n_rowtiles <= 512;
}

template <int XMODE>
static int launch_lstm(const LstmSeqArgs& a, int NT, int NL, bool head, hipStream_t st) {
    if (a.n_rowtiles <= 0 || a.T <= 0) return 0;
    if (XMODE == 0 && !head && lstm_rec16_applies(a, NL) && lstm_rec4_applies(a)) {
        if (NT == 13) return launch_lstm_rec4<13>(a, st);
        if (NT == 16) return launch_lstm_rec4<16>(a, st);
    }
    if (XMODE == 0 && !head && lstm_rec16_applies(a, NL)) {
        if (NT == 13) return launch_lstm_rec16<13>(a, st);
        if (NT == 16) return launch_lstm_rec16<16>(a, st);
    }
    if (XMODE == 0 && !head && lstm_rec_applies(a, NL)) {
        if (NT == 13) return a.sv_gates ? launch_lstm_rec<13, true>(a, st) : launch_lstm_rec<13, false>(a, st);
        if (NT == 16) return a.sv_gates ? launch_lstm_rec<16, true>(a, st) : launch_lstm_rec<16, false>(a, st);
    }
    switch (NT) {
        case 13: return launch_lstm_nt<13, XMODE>(a, NL, head, st);
        case 16: return launch_lstm_nt<16, XMODE>(a, NL, head, st);
        default:
            set_error("unsupported LSTM hidden size: %d tiles of 8 units (built: 13 = hidden 97..104, 16 = hidden 121..128)", NT);
            return -4;
    }
}

constexpr int kPreMaxQ = 16;     // input quads (8 features each) of a layer: <= 128 features
// Gate pre-activations b + W_ih . x of one layer for every (row tile, t) instance at once (training):
// the half of the gate products that does not depend on the recurrence, taken off the time loop and
// spread over as many CUs as there are instances.  One workgroup per instance, one wave per gate tile;
// the result lands in the layer's slots of sv_gates, where the recurrent launch (rec) picks it up and
// later overwrites it with the activated gates.
template <int NT>
__device__ __forceinline__ void lstm_pre_body(const float* w, const float* b, int nq_in, const float* x,
                                              long x_stride, float* gates, long g_stride, long n_inst, int ipw,
                                              const float* x2 = nullptr, long x2_stride = 0, int n4a = 0) {
    // an instance's x (4 nq_in k-steps x 64 lanes, contiguous) is shared by the NT waves: the workgroup copies it
    // into LDS once -- the NEXT instance's while this one's products run (double buffer, one barrier per instance) --
    // and every wave reads its B operands from there.  (Read straight from global memory, quad by quad in front of
    // the four products that use it, every quad exposed a full L2 round trip: 2.5 - 5 x the matrix time.)
    __shared__ __attribute__((aligned(16))) float xs[2][4 * kPreMaxQ * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = threadIdx.x;
    const int n4 = nq_in * 64;                                    // float4s of one instance's x
    // this wave's gate tile of W_ih stays in registers for the workgroup's ipw consecutive instances: streamed per
    // instance the launch is bound by L2 (NT x 13-16 KB of fragments for 0.4 us of products per wave)
    float4 wr[kPreMaxQ];
    {
        const float4* wp = (const float4*)w + wave * 64 + lane;
#pragma unroll
        for (int q = 0; q < kPreMaxQ; ++q) wr[q] = wp[(size_t)(q < nq_in ? q : nq_in - 1) * NT * 64];
    }
    const int toff1[1] = {wave * 64};
    f32x16 bias[1];
    load_bias<1>(bias, b, toff1, lane >> 5);
    const long i0 = (long)blockIdx.x * ipw;
    const int n_here = (int)((n_inst - i0) < (long)ipw ? (n_inst - i0) : (long)ipw);
    // (kPreMaxQ * 64 float4s over 64 NT >= 832 threads: at most two per thread.  Two named registers, not an array:
    // as `float4 pf[2]` indexed in an unrolled loop the pair stayed an alloca, which the compiler then placed in LDS --
    // every fetch became load, s_waitcnt vmcnt(0), ds_write at the TOP of the iteration, exposing the load's round
    // trip and the acknowledgement of the previous instance's sixteen stores before the products could start)
    float4 pf0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pf1 = pf0;
    const int e0 = tid, e1 = tid + 64 * NT;
    // (x2: an instance's x in two pieces -- its first n4a float4s at x, the rest at x2; the decoder's [ctx | z] read
    // where the two encoders left them)
    if (!x2) n4a = n4;
    auto fetch = [&](int k) {
        const float4* src = (const float4*)(x + (size_t)(i0 + k) * (size_t)x_stride);
        const float4* src2 = x2 ? (const float4*)(x2 + (size_t)(i0 + k) * (size_t)x2_stride) - n4a : src;
        if (e0 < n4) pf0 = e0 < n4a ? src[e0] : src2[e0];
        if (e1 < n4) pf1 = e1 < n4a ? src[e1] : src2[e1];
    };
    auto stash = [&](int buf, int off) {
        float4* dst = (float4*)xs[buf] + off;
        if (e0 < n4) dst[e0] = pf0;
        if (e1 < n4) dst[e1] = pf1;
    };
    if (n_here > 0) {
        fetch(0);
        stash(0, 0);
    }
    // every load so far (W_ih tile, bias, first x) has landed before the loop: with loads pending at its entry the
    // waits for them are repeated inside every iteration, where they also cover the iteration's own prefetch
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0)
    __syncthreads();
    for (int k = 0; k < n_here; ++k) {
        const size_t inst = (size_t)(i0 + k);
        if (k + 1 < n_here) fetch(k + 1);                          // lands while the products run
        const float* xb = xs[k & 1] + lane;
        f32x16 acc = bias[0];
#pragma unroll
        for (int q = 0; q < kPreMaxQ; ++q) {
            if (q < nq_in) {
                const float b0 = xb[(4 * q + 0) * 64], b1 = xb[(4 * q + 1) * 64], b2 = xb[(4 * q + 2) * 64], b3 = xb[(4 * q + 3) * 64];
                acc = mfma32(wr[q].x, b0, acc);
                acc = mfma32(wr[q].y, b1, acc);
                acc = mfma32(wr[q].z, b2, acc);
                acc = mfma32(wr[q].w, b3, acc);
            }
        }
        float* o = gates + inst * (size_t)g_stride + (size_t)(16 * wave) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r * 64] = acc[r];
        // (the LDS copy's address is tied to the last product so that the copy stays behind the products)
        int after = 0;                                             // a zero the compiler sees as made from acc[15]
        asm volatile("" : "+v"(after) : "v"(acc[15]));
        if (k + 1 < n_here) stash((k + 1) & 1, after);             // (that buffer's readers passed the last barrier)
        __syncthreads();
    }
}
template <int NT>
__global__ void __launch_bounds__(64 * NT) lstm_pre_kernel(const float* w, const float* b, int nq_in, const float* x,
                                                           long x_stride, float* gates, long g_stride, long n_inst, int ipw,
                                                           const float* x2, long x2_stride, int n4a) {
    lstm_pre_body<NT>(w, b, nq_in, x, x_stride, gates, g_stride, n_inst, ipw, x2, x2_stride, n4a);
}
// two launches as one (see lstm_rec4_pair_kernel): blockIdx.y picks the argument set, all of it scalars
struct PreArgs {
    const float *w, *b, *x;
    float* gates;
    long x_stride, g_stride, n_inst;
    int nq_in, ipw;
};
template <int NT>
__global__ void __launch_bounds__(64 * NT) lstm_pre_pair_kernel(const PreArgs p0, const PreArgs p1) {
    const bool y = blockIdx.y != 0;
    lstm_pre_body<NT>(y ? p1.w : p0.w, y ? p1.b : p0.b, y ? p1.nq_in : p0.nq_in, y ? p1.x : p0.x,
                      y ? p1.x_stride : p0.x_stride, y ? p1.gates : p0.gates, y ? p1.g_stride : p0.g_stride,
                      y ? p1.n_inst : p0.n_inst, y ? p1.ipw : p0.ipw);
}

static int launch_lstm_pre(int NT, const float* w, const float* b, int nq_in, const float* x, long x_stride,
                           float* gates, long g_stride, long insts, hipStream_t st,
                           const float* x2 = nullptr, long x2_stride = 0, int n4a = 0) {
    if (insts > 0x7fffffffL) { set_error("lstm_pre_kernel: too many instances"); return -2; }
    if (nq_in < 1 || nq_in > kPreMaxQ) { set_error("lstm_pre_kernel: %d input quads (1..%d built)", nq_in, kPreMaxQ); return -4; }
    if (insts <= 0) return 0;
    // instances per workgroup: one workgroup of NT waves fills a CU (its registers), so at most one round of 256
    // workgroups, each walking its share of the instances with the next one's x in flight (a workgroup per instance
    // paid the W_ih tile load and an unpipelined x fetch for every instance: 82 us instead of 25 at 960 instances)
    int ipw = (int)((insts + 255) / 256);
    ipw = ipw < 1 ? 1 : (ipw > 64 ? 64 : ipw);
    const unsigned grid = (unsigned)((insts + ipw - 1) / ipw);
    if (x2 && (n4a < 1 || n4a >= nq_in * 64)) { set_error("lstm_pre_kernel: bad split of the input"); return -1; }
    if (NT == 13) hipLaunchKernelGGL(lstm_pre_kernel<13>, dim3(grid), dim3(64 * 13), 0, st, w, b, nq_in, x, x_stride, gates, g_stride, insts, ipw, x2, x2_stride, n4a);
    else if (NT == 16) hipLaunchKernelGGL(lstm_pre_kernel<16>, dim3(grid), dim3(64 * 16), 0, st, w, b, nq_in, x, x_stride, gates, g_stride, insts, ipw, x2, x2_stride, n4a);
    else { set_error("unsupported LSTM hidden size: %d tiles of 8 units", NT); return -4; }
    return launch_check("lstm_pre_kernel");
}

// Layer-0 gate pre-activations per calendar day, transposed for the windows encoder (LstmSeqArgs::pre_t):
// one workgroup per tile of 32 days, one wave per gate tile.
template <int NT>
__global__ void __launch_bounds__(64 * NT) lstm_pre_t_kernel(const float* w, const float* b, int nq_in, const float* x,
                                                             float* pre_t, long pre_stride) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t ct = blockIdx.x;
    const int toff1[1] = {wave * 64};
    f32x16 acc[1];
    load_bias<1>(acc, b, toff1, lane >> 5);
    gemm_seg<1>(acc, (const float4*)w, lane, toff1, NT * 64, x + ct * (size_t)(4 * nq_in) * 64 + lane, nq_in);
    float* o = pre_t + (size_t)(lane >> 5) * pre_stride + ct * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)(2 * (16 * wave + r)) * pre_stride] = acc[0][r];
}

// two-layer training forward as one launch per layer (few row tiles: the one-wave-per-gate-tile schedule);
// VSV_FWD_FUSED=1 keeps the single two-layer launch
static bool per_layer_forward(int NL, long rowtiles) {
    static const bool fused = [] { const char* e = getenv("VSV_FWD_FUSED"); return e && e[0] == '1'; }();
    return NL == 2 && rowtiles <= kWideMaxRowtiles && !fused;
}

static thread_local hipEvent_t g_prof_ev[2] = {nullptr, nullptr};

static int fill_dropout(LstmSeqArgs& l, const vsv_train_save* sv, int NL, const char* who) {
    l.drop = DropArgs{0.0f, 1.0f, 0u, 0u, 0u, nullptr};
    l.sv_hd = nullptr;
    if (!(sv->drop_p > 0.0f) || NL < 2) return 0;              // nn.LSTM applies dropout between layers only
    if (sv->drop_p >= 1.0f) { set_error("%s: dropout probability must be < 1", who); return -1; }
    if (!sv->hd) { set_error("%s: dropout needs the hd save buffer", who); return -1; }
    l.drop = DropArgs{sv->drop_p, 1.0f / (1.0f - sv->drop_p), (uint32_t)sv->drop_seed, (uint32_t)(sv->drop_seed >> 32),
                      (uint32_t)sv->drop_id, (const uint32_t*)sv->drop_seed_dev};
    l.sv_hd = sv->hd;
    return 0;
}

static void fill_dense(DenseChainArgs& a, int li, const float* packed, const Sec& s) {
    a.layer[li].w = packed + s.w_off;
    a.layer[li].b = packed + s.b_off;
    a.layer[li].nq = s.nq;
    a.layer[li].nt = s.nt;
    a.layer[li].n_relu = s.n_relu;
    for (int t = 0; t < 4; ++t) {
        a.layer[li].q0[t] = s.qlo[t];
        a.layer[li].nqb[t] = s.qhi[t] - s.qlo[t];
    }
}

}  // namespace vsv

using namespace vsv;

#define VSV_LAYOUT_OR_FAIL(fn)                                  \
    Layout lay;                                                  \
    {                                                            \
        const char* msg_ = nullptr;                              \
        if (!dims || make_layout(*dims, lay, &msg_) != 0) {      \
            set_error(fn ": %s", msg_ ? msg_ : "null dims");     \
            return -1;                                           \
        }                                                        \
    }

extern "C" size_t vsv_encode_ws_bytes(const vsv_dims* dims, long n_rows, int T) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_rows < 0 || T < 0) return 0;
    const long rt = (n_rows + 31) / 32;
    const int kx = lay.enc[0].kx > lay.enc[1].kx ? lay.enc[0].kx : lay.enc[1].kx;
    return align256((size_t)rt * T * kx * 64 * sizeof(float)) + align256((size_t)rt * T * 4 * lay.NTm * 64 * sizeof(float)) + 256;
}

// ---- overlapping windows: everything that depends only on the calendar day is computed once per day ----
extern "C" size_t vsv_encode_windows_ws_bytes(const vsv_dims* dims, long n_windows, int T) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_windows < 0 || T < 0) return 0;
    const long rt = (n_windows + 31) / 32, ct = (n_windows + T - 1 + 31) / 32;
    const int kx = lay.enc[0].kx > lay.enc[1].kx ? lay.enc[0].kx : lay.enc[1].kx;
    return align256((size_t)ct * kx * 64 * sizeof(float)) + align256((size_t)32 * lay.NTm * ct * 32 * sizeof(float)) +
           align256((size_t)rt * T * 4 * lay.NTm * 64 * sizeof(float)) + 256;
}

static int encode_windows_impl(const vsv_dims* dims, const float* packed, int which, const float* vol, const float* ex,
                               long n_windows, int T, const float* state_in, float* state_out, float* out, void* ws,
                               size_t ws_bytes, void* stream);

extern "C" int vsv_encode_windows(const vsv_dims* dims, const float* packed, int which, const float* vol, const float* ex,
                                  long n_windows, int T, float* out, void* ws, size_t ws_bytes, void* stream) {
    return encode_windows_impl(dims, packed, which, vol, ex, n_windows, T, nullptr, nullptr, out, ws, ws_bytes, stream);
}

extern "C" int vsv_encode_windows_state(const vsv_dims* dims, const float* packed, int which, const float* vol,
                                        const float* ex, long n_windows, int T, const float* state_in, float* state_out,
                                        float* out, void* ws, size_t ws_bytes, void* stream) {
    return encode_windows_impl(dims, packed, which, vol, ex, n_windows, T, state_in, state_out, out, ws, ws_bytes, stream);
}

static int encode_windows_impl(const vsv_dims* dims, const float* packed, int which, const float* vol, const float* ex,
                               long n_windows, int T, const float* state_in, float* state_out, float* out, void* ws,
                               size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_encode_windows")
    if (which < 0 || which > 1) { set_error("vsv_encode_windows: which must be 0 or 1"); return -1; }
    if (n_windows < 0 || T < 0) { set_error("vsv_encode_windows: negative size"); return -1; }
    if (n_windows == 0 || T == 0) return 0;                       // nothing to do (empty outputs have no address)
    if (!packed || !vol || !out || !ws) { set_error("vsv_encode_windows: null pointer"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_encode_windows: ex_feats_dim > 0 but ex is null"); return -1; }
    const long rt = (n_windows + 31) / 32;
    if (rt > kWideMaxRowtiles) { set_error("vsv_encode_windows: at most %ld windows per call (use vsv_encode beyond)", kWideMaxRowtiles * 32); return -5; }
    if (ws_bytes < vsv_encode_windows_ws_bytes(dims, n_windows, T)) { set_error("vsv_encode_windows: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    const int NT = lay.NTm;
    const long n_cal = n_windows + T - 1, ct = (n_cal + 31) / 32;
    float* xe = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* pre = (float*)((char*)xe + align256((size_t)ct * e.kx * 64 * sizeof(float)));
    float* hbuf = (float*)((char*)pre + align256((size_t)32 * NT * ct * 32 * sizeof(float)));

    // 1. conv embedding of every calendar day (T = 1 "sequences")
    DenseChainArgs a = {};
    a.n_layers = e.n_chain;
    for (int i = 0; i < e.n_chain; ++i) fill_dense(a, i, packed, e.conv[i]);
    a.inj_layer = e.n_conv - 1;
    a.n_rows = n_cal; a.n_rowtiles = ct; a.T = 1;
    a.in_mode = 0; a.in = vol; a.in_rs = 25; a.in_ts = 0; a.in_f = 25;
    a.in_ks = 4 * e.conv[0].nq;
    if (lay.X > 0) { a.inj = ex; a.inj_rs = lay.X; a.inj_ts = 0; a.inj_f0 = e.ex_f0; a.inj_n = lay.X; }
    a.out_mode = 0; a.out = xe; a.out_f = e.E; a.S = 1;
    a.aux_layer = -1;
    int rc = launch_dense_chain(a, st);
    if (rc) return rc;

    // 2. layer 0's b + W_ih . x of every calendar day, transposed
    const float* w0 = packed + e.lstm[0].w_off;
    const float* b0 = packed + e.lstm[0].b_off;
    if (NT == 13) hipLaunchKernelGGL(lstm_pre_t_kernel<13>, dim3((unsigned)ct), dim3(64 * 13), 0, st, w0, b0, e.kx >> 2, xe, pre, ct * 32);
    else if (NT == 16) hipLaunchKernelGGL(lstm_pre_t_kernel<16>, dim3((unsigned)ct), dim3(64 * 16), 0, st, w0, b0, e.kx >> 2, xe, pre, ct * 32);
    else { set_error("unsupported LSTM hidden size: %d tiles of 8 units", NT); return -4; }
    rc = launch_check("lstm_pre_t_kernel");
    if (rc) return rc;

    // 3. LSTM over the windows (row r, step t reads day r + t), top-layer h of every step -> hbuf
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    for (int i = 0; i < lay.NL; ++i) { l.w[i] = packed + e.lstm[i].w_off; l.b[i] = packed + e.lstm[i].b_off; }
    l.kx = e.kx; l.n_rows = n_windows; l.n_rowtiles = rt; l.T = T;
    l.pre_t = pre; l.pre_stride = ct * 32; l.pre_tstride = 1; l.L = lay.L;
    l.hout = hbuf; l.t_out0 = 0;
    l.din.state_in = state_in; l.din.state_div = 1;
    l.state_out = state_out;
    rc = launch_lstm<0>(l, NT, lay.NL, false, st);
    if (rc) return rc;

    // 4. linear head on every step
    DenseChainArgs h = {};
    h.n_layers = 1;
    fill_dense(h, 0, packed, e.head);
    h.n_rows = n_windows; h.n_rowtiles = rt; h.T = T;
    h.in_mode = 1; h.in = hbuf; h.in_ks = 4 * NT;
    h.out_mode = 1; h.out = out; h.out_f = e.nh; h.S = 1;
    h.aux_layer = -1;
    return launch_dense_chain(h, st);
}

// ---- the date-range prefix as a pipeline of LAYER stages (two-layer memories) ---------------------------------
// A recurrent launch holds one CU per 32-row tile for its T sequential steps, and a fused two-layer step costs
// that CU the matrix time of both layers.  Split at the layer boundary -- layer 0's recurrence; layer 1's input
// product b + W_ih . h0 for all steps at once (lstm_pre_kernel, off the time loop); layer 1's recurrence -- the
// two recurrences of consecutive TIME SEGMENTS run on different CUs at the same time (the caller puts the two
// calls on different streams): the prefix's critical path is one layer's steps plus a segment of the other.
// Same products in the same order as the fused launches: results are bit-equal.
static size_t enc_split_bytes(const Layout& lay, long n_windows, int T, size_t off[5]) {
    const long rt = (n_windows + 31) / 32, ct = (n_windows + T - 1 + 31) / 32;
    const int kx = lay.enc[0].kx > lay.enc[1].kx ? lay.enc[0].kx : lay.enc[1].kx;
    size_t cur = 0;
    off[0] = cur; cur += align256((size_t)ct * kx * 64 * sizeof(float));                      // embeddings per day
    off[1] = cur; cur += align256((size_t)32 * lay.NTm * ct * 32 * sizeof(float));            // layer-0 table
    off[2] = cur; cur += align256((size_t)rt * T * 4 * lay.NTm * 64 * sizeof(float));         // h0 of every step
    off[3] = cur; cur += align256((size_t)rt * T * 16 * lay.NTm * 64 * sizeof(float));        // layer-1 pre-activations
    off[4] = cur; cur += align256((size_t)rt * T * 4 * lay.NTm * 64 * sizeof(float));         // h1 of every step
    return cur + 256;
}

extern "C" size_t vsv_encode_windows_split_ws_bytes(const vsv_dims* dims, long n_windows, int T) {
    Layout lay;
    const char* msg = nullptr;
    size_t off[5];
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_windows < 0 || T < 0) return 0;
    return enc_split_bytes(lay, n_windows, T, off);
}

extern "C" int vsv_encode_windows_layer0(const vsv_dims* dims, const float* packed, int which, const float* vol,
                                         const float* ex, long n_windows, int T, const float* state_in,
                                         float* state_out, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_encode_windows_layer0")
    if (which < 0 || which > 1) { set_error("vsv_encode_windows_layer0: which must be 0 or 1"); return -1; }
    if (lay.NL != 2) { set_error("vsv_encode_windows_layer0: the layer split is for two-layer memories"); return -4; }
    if (n_windows < 0 || T < 0) { set_error("vsv_encode_windows_layer0: negative size"); return -1; }
    if (n_windows == 0 || T == 0) return 0;
    if (!packed || !vol || !ws) { set_error("vsv_encode_windows_layer0: null pointer"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_encode_windows_layer0: ex_feats_dim > 0 but ex is null"); return -1; }
    const long rt = (n_windows + 31) / 32;
    if (rt > kWideMaxRowtiles) { set_error("vsv_encode_windows_layer0: at most %ld windows per call", kWideMaxRowtiles * 32); return -5; }
    size_t off[5];
    if (ws_bytes < enc_split_bytes(lay, n_windows, T, off)) { set_error("vsv_encode_windows_layer0: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    const int NT = lay.NTm;
    const long n_cal = n_windows + T - 1, ct = (n_cal + 31) / 32;
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* xe = (float*)(base + off[0]);
    float* pre = (float*)(base + off[1]);
    float* h0 = (float*)(base + off[2]);

    DenseChainArgs a = {};
    a.n_layers = e.n_chain;
    for (int i = 0; i < e.n_chain; ++i) fill_dense(a, i, packed, e.conv[i]);
    a.inj_layer = e.n_conv - 1;
    a.n_rows = n_cal; a.n_rowtiles = ct; a.T = 1;
    a.in_mode = 0; a.in = vol; a.in_rs = 25; a.in_ts = 0; a.in_f = 25;
    a.in_ks = 4 * e.conv[0].nq;
    if (lay.X > 0) { a.inj = ex; a.inj_rs = lay.X; a.inj_ts = 0; a.inj_f0 = e.ex_f0; a.inj_n = lay.X; }
    a.out_mode = 0; a.out = xe; a.out_f = e.E; a.S = 1;
    a.aux_layer = -1;
    int rc = launch_dense_chain(a, st);
    if (rc) return rc;
    const float* w0 = packed + e.lstm[0].w_off;
    const float* b0 = packed + e.lstm[0].b_off;
    if (NT == 13) hipLaunchKernelGGL(lstm_pre_t_kernel<13>, dim3((unsigned)ct), dim3(64 * 13), 0, st, w0, b0, e.kx >> 2, xe, pre, ct * 32);
    else if (NT == 16) hipLaunchKernelGGL(lstm_pre_t_kernel<16>, dim3((unsigned)ct), dim3(64 * 16), 0, st, w0, b0, e.kx >> 2, xe, pre, ct * 32);
    else { set_error("unsupported LSTM hidden size: %d tiles of 8 units", NT); return -4; }
    rc = launch_check("lstm_pre_t_kernel");
    if (rc) return rc;
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    l.w[0] = w0; l.b[0] = b0;
    l.kx = e.kx; l.n_rows = n_windows; l.n_rowtiles = rt; l.T = T;
    l.pre_t = pre; l.pre_stride = ct * 32; l.pre_tstride = 1; l.L = lay.L;
    l.hout = h0; l.t_out0 = 0;
    l.din.state_in = state_in; l.din.state_div = 1;
    l.state_out = state_out;
    l.st_nl = 2; l.st_l0 = 0;
    return launch_lstm<0>(l, NT, 1, false, st);
}

static int encode_windows_layer1_impl(const vsv_dims* dims, const float* packed, int which, long n_windows, int T,
                                      const float* state_in, float* state_out, float* out, int out_f, void* ws,
                                      size_t ws_bytes, void* stream, float* head_tb = nullptr, int head_slots = 0,
                                      bool no_head = false);

extern "C" int vsv_encode_windows_layer1(const vsv_dims* dims, const float* packed, int which, long n_windows, int T,
                                         const float* state_in, float* state_out, float* out, void* ws,
                                         size_t ws_bytes, void* stream) {
    return encode_windows_layer1_impl(dims, packed, which, n_windows, T, state_in, state_out, out, 0, ws, ws_bytes, stream);
}

// out_f: head outputs written per (window, step), 0 = all (the prefix pipeline wants z_mean only: L of 2L).
// The date-range prefix (vsv_context_prefix, fused hand-over) has the decoder read the encoders' outputs in the
// tile-B form the recurrent launch leaves them in: head_tb / head_slots -- the head's first out_f outputs of every
// step as tile-B [inst][head_slots][64], computed inside the recurrent launch (LstmSeqArgs::hd_out), no head launch;
// no_head -- the head is the identity (compress_context off) and the consumer reads h1 from the workspace itself.
static int encode_windows_layer1_impl(const vsv_dims* dims, const float* packed, int which, long n_windows, int T,
                                      const float* state_in, float* state_out, float* out, int out_f, void* ws,
                                      size_t ws_bytes, void* stream, float* head_tb, int head_slots, bool no_head) {
    VSV_LAYOUT_OR_FAIL("vsv_encode_windows_layer1")
    if (which < 0 || which > 1) { set_error("vsv_encode_windows_layer1: which must be 0 or 1"); return -1; }
    if (lay.NL != 2) { set_error("vsv_encode_windows_layer1: the layer split is for two-layer memories"); return -4; }
    if (n_windows < 0 || T < 0) { set_error("vsv_encode_windows_layer1: negative size"); return -1; }
    if (n_windows == 0 || T == 0) return 0;
    if (!packed || (!out && !head_tb && !no_head) || !ws) { set_error("vsv_encode_windows_layer1: null pointer"); return -1; }
    const long rt = (n_windows + 31) / 32;
    if (rt > kWideMaxRowtiles) { set_error("vsv_encode_windows_layer1: at most %ld windows per call", kWideMaxRowtiles * 32); return -5; }
    size_t off[5];
    if (ws_bytes < enc_split_bytes(lay, n_windows, T, off)) { set_error("vsv_encode_windows_layer1: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    const int NT = lay.NTm;
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const float* h0 = (const float*)(base + off[2]);
    float* g1 = (float*)(base + off[3]);
    float* h1 = (float*)(base + off[4]);
    const float* w1 = packed + e.lstm[1].w_off;
    const float* b1 = packed + e.lstm[1].b_off;
    int rc = launch_lstm_pre(NT, w1, b1, NT, h0, (long)4 * NT * 64, g1, (long)16 * NT * 64, rt * T, st);
    if (rc) return rc;
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    l.w[0] = w1; l.b[0] = b1;
    l.kx = 4 * NT; l.n_rows = n_windows; l.n_rowtiles = rt; l.T = T;
    l.pre_g = g1; l.L = lay.L;
    l.hout = h1; l.t_out0 = 0;
    l.din.state_in = state_in; l.din.state_div = 1;
    l.state_out = state_out;
    l.st_nl = 2; l.st_l0 = 1;
    if (head_tb) {
        if (NT != 13 || e.head.nt != 1 || e.head.nq != NT || out_f < 1 || out_f > 16 || head_slots < 1 || head_slots > 8 ||
            !lstm_rec16_applies(l, 1) || lstm_rec4_applies(l)) {
            set_error("vsv_encode_windows_layer1: the head cannot ride in this recurrent launch"); return -4;
        }
        l.hd_w = packed + e.head.w_off; l.hd_b = packed + e.head.b_off;
        l.hd_out = head_tb; l.hd_n = out_f; l.hd_slots = head_slots;
    }
    rc = launch_lstm<0>(l, NT, 1, false, st);
    if (rc || head_tb || no_head) return rc;
    DenseChainArgs h = {};
    h.n_layers = 1;
    fill_dense(h, 0, packed, e.head);
    h.n_rows = n_windows; h.n_rowtiles = rt; h.T = T;
    h.in_mode = 1; h.in = h1; h.in_ks = 4 * NT;
    h.out_mode = 1; h.out = out; h.out_f = (out_f > 0 && out_f < e.nh) ? out_f : e.nh; h.S = 1;
    h.aux_layer = -1;
    return launch_dense_chain(h, st);
}

static size_t dec_split_bytes(const Layout& lay, long n_rows, int T, size_t off[4]) {
    const long rt = (n_rows + 31) / 32;
    off[0] = 0;                                                                                // h0
    off[1] = align256((size_t)rt * T * 4 * lay.NTd * 64 * sizeof(float));                      // layer-1 pre-activations
    off[2] = off[1] + align256((size_t)rt * T * 16 * lay.NTd * 64 * sizeof(float));            // layer-0 pre-activations
    off[3] = off[2] + align256((size_t)rt * T * 16 * lay.NTd * 64 * sizeof(float));            // [ctx | z] in tile-B
    return off[3] + align256((size_t)rt * T * lay.dec.kx * 64 * sizeof(float)) + 256;
}

extern "C" size_t vsv_decode_prefix_split_ws_bytes(const vsv_dims* dims, long n_rows, int T) {
    Layout lay;
    const char* msg = nullptr;
    size_t off[4];
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_rows < 0 || T < 0) return 0;
    return dec_split_bytes(lay, n_rows, T, off);
}

static int decode_prefix_check(const char* who, const Layout& lay, const float* packed, const vsv_decode_in* in, long n_rows,
                               int T, void* ws, size_t ws_bytes, size_t off[4]) {
    if (lay.NL != 2) { set_error("%s: the layer split is for two-layer memories", who); return -4; }
    if (!packed || !in || !ws) { set_error("%s: null pointer", who); return -1; }
    if (n_rows < 1 || n_rows > 0x7fffffffL || T < 1) { set_error("%s: bad sizes", who); return -1; }
    if ((n_rows + 31) / 32 > kWideMaxRowtiles) { set_error("%s: at most %ld rows per call", who, kWideMaxRowtiles * 32); return -5; }
    if (ws_bytes < dec_split_bytes(lay, n_rows, T, off)) { set_error("%s: workspace too small", who); return -1; }
    if (in->gate_in) { set_error("%s: gate_in is for the Monte-Carlo draw, not the prefix", who); return -1; }
    if (in->state_in && in->state_div != 1) { set_error("%s: state_div must be 1", who); return -1; }
    return 0;
}

// ctx_tb / z_tb (the fused hand-over of vsv_context_prefix): the input [ctx_t | z_mean_t] of every step read in two
// pieces where the encoders' recurrent launches left them in tile-B -- the context head's outputs or the context
// memory's h1 ([inst][ctx_slots][64], the first Lc/2 slots) and the posterior head's outputs ([inst][z_slots][64]) --
// instead of assembled by a launch of its own
static int decode_prefix_layer0_impl(const vsv_dims* dims, const float* packed, const vsv_decode_in* in, long n_rows,
                                     int T, float* state_out, void* ws, size_t ws_bytes, void* stream,
                                     const float* ctx_tb, int ctx_slots, const float* z_tb, int z_slots);

extern "C" int vsv_decode_prefix_layer0(const vsv_dims* dims, const float* packed, const vsv_decode_in* in, long n_rows,
                                        int T, float* state_out, void* ws, size_t ws_bytes, void* stream) {
    return decode_prefix_layer0_impl(dims, packed, in, n_rows, T, state_out, ws, ws_bytes, stream, nullptr, 0, nullptr, 0);
}

static int decode_prefix_layer0_impl(const vsv_dims* dims, const float* packed, const vsv_decode_in* in, long n_rows,
                                     int T, float* state_out, void* ws, size_t ws_bytes, void* stream,
                                     const float* ctx_tb, int ctx_slots, const float* z_tb, int z_slots) {
    VSV_LAYOUT_OR_FAIL("vsv_decode_prefix_layer0")
    size_t off[4];
    if (int rc = decode_prefix_check("vsv_decode_prefix_layer0", lay, packed, in, n_rows, T, ws, ws_bytes, off)) return rc;
    if (!ctx_tb && (in->ctx_T != T || in->z_T != T || !in->ctx || !in->zb || in->lv)) {
        set_error("vsv_decode_prefix_layer0: the prefix runs on (ctx_t, z_mean_t) for all T steps"); return -1;
    }
    const DecLayout& dc = lay.dec;
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    l.w[0] = packed + dc.lstm[0].w_off; l.b[0] = packed + dc.lstm[0].b_off;
    l.kx = dc.kx; l.n_rows = n_rows; l.n_rowtiles = (n_rows + 31) / 32; l.T = T;
    l.L = lay.L; l.Lc = lay.Lc; l.din = *in;
    if (l.din.ctx_div < 1) l.din.ctx_div = 1;
    if (l.din.z_div < 1) l.din.z_div = 1;
    if (l.din.S < 1) l.din.S = 1;
    l.din.state_div = 1;
    l.state_out = state_out;
    l.hout = (float*)(base + off[0]); l.t_out0 = 0;
    l.st_nl = 2; l.st_l0 = 0;
    hipStream_t st = (hipStream_t)stream;
    LstmSeqArgs r = l;
    r.pre_g = (const float*)(base + off[2]);
    if (!lstm_rec16_applies(r, 1)) {
        if (ctx_tb) { set_error("vsv_decode_prefix_layer0: the tile-B hand-over needs the layer-stage launches"); return -4; }
        return launch_lstm<1>(l, lay.NTd, 1, false, st);
    }
    // [ctx_t | z_mean_t] of every step, its gate pre-activations as one GEMM over all (row tile, step) instances, and
    // the 16-row recurrent launch
    int rc;
    if (ctx_tb) {
        if (lay.Lc % 2 || z_slots != dc.kx - lay.Lc / 2 || z_slots < 1) { set_error("vsv_decode_prefix_layer0: bad tile-B hand-over"); return -1; }
        rc = launch_lstm_pre(lay.NTd, l.w[0], l.b[0], dc.kx >> 2, ctx_tb, (long)ctx_slots * 64, (float*)(base + off[2]),
                             (long)16 * lay.NTd * 64, l.n_rowtiles * T, st, z_tb, (long)z_slots * 64, (lay.Lc / 2) * 16);
    } else {
        float* x = (float*)(base + off[3]);
        l.sv_x = x;
        const long total = l.n_rowtiles * T * dc.kx * 64;
        if ((total + 255) / 256 > 0x7fffffffL) { set_error("vsv_decode_prefix_layer0: too many instances"); return -2; }
        hipLaunchKernelGGL(decoder_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, l);
        rc = launch_check("decoder_input_kernel");
        if (rc) return rc;
        rc = launch_lstm_pre(lay.NTd, l.w[0], l.b[0], dc.kx >> 2, x, (long)dc.kx * 64, (float*)(base + off[2]), (long)16 * lay.NTd * 64,
                             l.n_rowtiles * T, st);
    }
    if (rc) return rc;
    r.din = vsv_decode_in();
    r.din.state_in = in->state_in; r.din.state_div = 1;
    r.din.gate_out = in->gate_out;
    return launch_lstm<0>(r, lay.NTd, 1, false, st);
}

extern "C" int vsv_decode_prefix_layer1(const vsv_dims* dims, const float* packed, const vsv_decode_in* in, long n_rows,
                                        int T, float* state_out, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_decode_prefix_layer1")
    size_t off[4];
    if (int rc = decode_prefix_check("vsv_decode_prefix_layer1", lay, packed, in, n_rows, T, ws, ws_bytes, off)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const DecLayout& dc = lay.dec;
    const int NT = lay.NTd;
    const long rt = (n_rows + 31) / 32;
    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const float* h0 = (const float*)(base + off[0]);
    float* g1 = (float*)(base + off[1]);
    const float* w1 = packed + dc.lstm[1].w_off;
    const float* b1 = packed + dc.lstm[1].b_off;
    int rc = launch_lstm_pre(NT, w1, b1, NT, h0, (long)4 * NT * 64, g1, (long)16 * NT * 64, rt * T, st);
    if (rc) return rc;
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    l.w[0] = w1; l.b[0] = b1;
    l.kx = 4 * NT; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.pre_g = g1; l.L = lay.L;
    l.din.state_in = in->state_in; l.din.state_div = 1;
    l.din.gate_out = in->gate_out;
    l.state_out = state_out;
    l.t_out0 = T;
    l.st_nl = 2; l.st_l0 = 1;
    return launch_lstm<0>(l, NT, 1, false, st);
}

// ---- the whole date-range prefix as ONE call: every stage launch of every time segment, issued from native code
// (the Python loop over the same entry points spends ~0.15 ms of host time per segment, more than a segment's
// stages take on the GPU) onto the caller's worker streams, ordered by HIP events.
struct PrefixPlan {
    int n_seg;
    size_t seg_off[65];          // per segment: start of its block
    size_t o_wse, o_wsc, o_wsd, o_zm, o_ctx, o_ste, o_stc, o_std;   // offsets inside a segment's block
    size_t wse_bytes, wsd_bytes;
    size_t total;
};

// a head's outputs for every (day, step): row-major (width floats), or -- the fused hand-over -- tile-B of up to 8
// slots per (row tile, step)
static size_t prefix_head_bytes(long n_days, int T, int width) {
    const size_t rows = (size_t)n_days * T * width, tb = (size_t)((n_days + 31) / 32) * T * 8 * 64;
    return align256((rows > tb ? rows : tb) * sizeof(float));
}

// The decoder reads the encoders' outputs where their recurrent launches leave them (no head launches, no input
// assembly): the posterior head's z_mean -- and with compress_context the context head -- ride as the spare gate
// tile of lstm_rec16_kernel<13> and land in tile-B; without compress_context the context embedding is the memory's
// h1 itself.  For the sizes that run the 16-row launches.  VSV_PREFIX_FUSE=0: off.
static bool prefix_fused(const vsv_dims& dims, const Layout& lay, long n_days) {
    static const bool off = [] { const char* e = getenv("VSV_PREFIX_FUSE"); return e && e[0] == '0'; }();
    const int cs = lay.Lc / 2, zs = lay.dec.kx - cs;
    LstmSeqArgs probe = {};
    probe.n_rowtiles = (n_days + 31) / 32;
    probe.pre_g = (const float*)&probe;                          // (any non-null pointer: the test is on the launch's kind)
    const bool ctx_ok = dims.compress_context ? (lay.enc[1].head.nt == 1 && lay.enc[1].head.nq == 13 && lay.Lc <= 16 && cs >= 1 && cs <= 8)
                                              : lay.Lc == lay.Hm;
    return !off && ctx_ok && lay.gru == 0 && lay.NL == 2 && lay.Lc % 2 == 0 && lay.NTm == 13 &&
           lay.enc[0].head.nt == 1 && lay.enc[0].head.nq == 13 && lay.L <= 16 && 2 * zs >= lay.L && zs >= 1 && zs <= 8 &&
           lstm_rec16_applies(probe, 1) && !lstm_rec4_applies(probe);
}

static int prefix_plan(const Layout& lay, long n_days, int C, int n_seg, PrefixPlan& p) {
    if (n_seg < 1 || n_seg > 64 || C < n_seg) return -1;
    p.n_seg = n_seg;
    size_t cur = 0;
    const size_t nse = (size_t)lay.NL * 2 * 8 * lay.NTm, nsd = (size_t)lay.NL * 2 * 8 * lay.NTd;
    for (int i = 0; i < n_seg; ++i) {
        const int T = ((i + 1) * C) / n_seg - (i * C) / n_seg;
        size_t off5[5], off2[4];
        p.seg_off[i] = cur;
        // (the largest segment's sizes serve all: blocks are laid out with the first segment's T or larger)
        const size_t wse = enc_split_bytes(lay, n_days, T, off5), wsd = dec_split_bytes(lay, n_days, T, off2);
        size_t b = 0;
        b += align256(wse);                                    // posterior encoder workspace
        b += align256(wse);                                    // context encoder workspace
        b += align256(wsd);                                    // decoder workspace
        b += prefix_head_bytes(n_days, T, lay.L) + prefix_head_bytes(n_days, T, lay.Lc);   // z_mean, ctx embedding
        b += 2 * align256((size_t)n_days * nse * sizeof(float));            // encoder states
        b += align256((size_t)n_days * nsd * sizeof(float));                // decoder state
        cur += b;
    }
    p.seg_off[n_seg] = cur;
    p.total = cur + 256;
    return 0;
}

extern "C" size_t vsv_context_prefix_ws_bytes(const vsv_dims* dims, long n_days, int ctx_len, int n_seg) {
    Layout lay;
    const char* msg = nullptr;
    PrefixPlan p;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_days < 0 || prefix_plan(lay, n_days, ctx_len, n_seg, p) != 0) return 0;
    return p.total;
}

extern "C" int vsv_context_prefix(const vsv_dims* dims, const float* packed, const float* vol, const float* ex, long n_days,
                                  int ctx_len, int n_seg, void* const* streams, int n_streams, float* state_out,
                                  float* gate_out, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_context_prefix")
    if (lay.NL != 2) { set_error("vsv_context_prefix: the layer split is for two-layer memories"); return -4; }
    if (!packed || !vol || !streams || !state_out || !gate_out || !ws) { set_error("vsv_context_prefix: null pointer"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_context_prefix: ex_feats_dim > 0 but ex is null"); return -1; }
    if (n_days < 1 || (n_days + 31) / 32 > kWideMaxRowtiles) { set_error("vsv_context_prefix: 1..%ld days per call", kWideMaxRowtiles * 32); return -5; }
    if (n_streams != 2 && n_streams != 3 && n_streams != 4 && n_streams != 6) { set_error("vsv_context_prefix: 2, 3, 4 or 6 worker streams"); return -1; }
    PrefixPlan pl;
    if (prefix_plan(lay, n_days, ctx_len, n_seg, pl) != 0) { set_error("vsv_context_prefix: 1 <= n_seg <= min(64, ctx_len)"); return -1; }
    if (ws_bytes < pl.total) { set_error("vsv_context_prefix: workspace too small"); return -1; }
    hipStream_t main = (hipStream_t)stream;
    // stage -> stream: (encoder L0, encoder L1, context L0, context L1, decoder L0, decoder L1); with four
    // streams (HIP's default number of hardware queues) the two encoders' same-layer stages share one
    hipStream_t sq[6];
    for (int k = 0; k < 6; ++k) {
        // (two: the posterior encoder and the decoder on one stream -- it may be the caller's own --, the context
        // encoder beside them on the other: the fewest hardware queues, the schedule for unsegmented prefixes)
        static const int map4[6] = {0, 0, 1, 1, 2, 3}, map3[6] = {0, 0, 1, 1, 2, 2}, map2[6] = {0, 0, 1, 1, 0, 0};
        sq[k] = (hipStream_t)streams[n_streams == 6 ? k : (n_streams == 4 ? map4[k] : (n_streams == 3 ? map3[k] : map2[k]))];
    }
    // every event the call needs is created BEFORE the fork: nothing below can fail between the fork and the join
    // except a launch, and a failed launch still runs through the join (rc is returned after it)
    std::vector<hipEvent_t> ev((size_t)1 + 5 * (size_t)n_seg + (size_t)n_streams, nullptr);
    for (size_t k = 0; k < ev.size(); ++k) {
        const hipError_t e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (size_t j = 0; j < k; ++j) (void)hipEventDestroy(ev[j]);
            set_error("vsv_context_prefix: hipEventCreate: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    hipEvent_t fork = ev[0];
    hipEvent_t* const seg_ev = ev.data() + 1;
    hipEvent_t* const join_ev = ev.data() + 1 + 5 * (size_t)n_seg;
    (void)hipEventRecord(fork, main);
    for (int k = 0; k < n_streams; ++k) (void)hipStreamWaitEvent((hipStream_t)streams[k], fork, 0);

    char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const size_t nse = (size_t)lay.NL * 2 * 8 * lay.NTm;
    const float *st_e = nullptr, *st_c = nullptr, *st_d = nullptr;
    const bool fused = prefix_fused(*dims, lay, n_days);
    const int cslots = lay.Lc / 2, zslots = lay.dec.kx - cslots;
    int rc = 0;
    for (int i = 0; i < n_seg && rc == 0; ++i) {
        const int t0 = (i * ctx_len) / n_seg, T = ((i + 1) * ctx_len) / n_seg - t0;
        const bool last = i == n_seg - 1;
        size_t off5[5], off2[4];
        const size_t wse = enc_split_bytes(lay, n_days, T, off5), wsd = dec_split_bytes(lay, n_days, T, off2);
        char* b = base + pl.seg_off[i];
        void* ws_e = b; b += align256(wse);
        void* ws_c = b; b += align256(wse);
        void* ws_d = b; b += align256(wsd);
        float* zm = (float*)b; b += prefix_head_bytes(n_days, T, lay.L);
        float* cx = (float*)b; b += prefix_head_bytes(n_days, T, lay.Lc);
        float* ne = (float*)b; b += align256((size_t)n_days * nse * sizeof(float));
        float* nc = (float*)b; b += align256((size_t)n_days * nse * sizeof(float));
        float* nd = last ? state_out : (float*)b;
        const float* v = vol + (size_t)25 * t0;
        const float* e = lay.X > 0 ? ex + (size_t)lay.X * t0 : nullptr;
        hipEvent_t* const evs = seg_ev + 5 * (size_t)i;
        // encoders: layer 0, then (other stream) layer 1 + head; the final states of the last segment are not needed
        for (int which = 0; which < 2 && rc == 0; ++which) {
            hipStream_t s0 = sq[2 * which], s1 = sq[2 * which + 1];
            float* so = last ? nullptr : (which ? nc : ne);
            const float* si = which ? st_c : st_e;
            void* w = which ? ws_c : ws_e;
            rc = vsv_encode_windows_layer0(dims, packed, which, v, e, n_days, T, si, so, w, wse, s0);
            if (rc) break;
            (void)hipEventRecord(evs[which], s0);
            (void)hipStreamWaitEvent(s1, evs[which], 0);
            if (fused && which == 0) rc = encode_windows_layer1_impl(dims, packed, 0, n_days, T, si, so, nullptr, lay.L, w, wse, s1, zm, zslots, false);
            else if (fused && dims->compress_context) rc = encode_windows_layer1_impl(dims, packed, 1, n_days, T, si, so, nullptr, lay.Lc, w, wse, s1, cx, cslots, false);
            else if (fused) rc = encode_windows_layer1_impl(dims, packed, 1, n_days, T, si, so, nullptr, 0, w, wse, s1, nullptr, 0, true);
            else rc = encode_windows_layer1_impl(dims, packed, which, n_days, T, si, so, which ? cx : zm, which ? 0 : lay.L, w, wse, s1);
            if (rc) break;
            (void)hipEventRecord(evs[2 + which], s1);
            (void)hipStreamWaitEvent(sq[4], evs[2 + which], 0);
        }
        if (rc == 0) {
            vsv_decode_in din = {};
            din.ctx = cx; din.ctx_rs = (long)T * lay.Lc; din.ctx_T = T; din.ctx_div = 1;
            din.zb = zm; din.z_rs = (long)T * lay.L; din.z_T = T; din.z_div = 1;
            din.S = 1; din.state_in = st_d; din.state_div = 1;
            din.gate_out = last ? gate_out : nullptr;
            if (fused) {
                // (the context memory's h1 of every step: block 4 of its workspace, see enc_split_bytes)
                const float* h1c = (const float*)((char*)(((uintptr_t)ws_c + 255) & ~(uintptr_t)255) + off5[4]);
                if (dims->compress_context) rc = decode_prefix_layer0_impl(dims, packed, &din, n_days, T, nd, ws_d, wsd, sq[4], cx, cslots, zm, zslots);
                else rc = decode_prefix_layer0_impl(dims, packed, &din, n_days, T, nd, ws_d, wsd, sq[4], h1c, 4 * lay.NTm, zm, zslots);
            } else {
                rc = vsv_decode_prefix_layer0(dims, packed, &din, n_days, T, nd, ws_d, wsd, sq[4]);
            }
            if (rc == 0) {
                (void)hipEventRecord(evs[4], sq[4]);
                (void)hipStreamWaitEvent(sq[5], evs[4], 0);
                rc = vsv_decode_prefix_layer1(dims, packed, &din, n_days, T, nd, ws_d, wsd, sq[5]);
            }
        }
        st_e = ne; st_c = nc; st_d = nd;
    }
    // join: the caller's stream continues when every worker stream has drained (also after a failed launch: the
    // segments issued before it are still running on the worker streams)
    for (int k = 0; k < n_streams; ++k) {
        (void)hipEventRecord(join_ev[k], (hipStream_t)streams[k]);
        (void)hipStreamWaitEvent(main, join_ev[k], 0);
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    return rc;
}

// ---- the one entry point that allocates nothing, for callers that cannot call hipMalloc (code under HIP graph
// capture, tight serving loops, harnesses that forbid allocation): vsv_encode_windows with its workspace in a
// file-scope device array instead of a caller buffer.  Each GPU holds its own copy of the array at its own
// address; the address is looked up once per device (first call on that device) and kept.  Host code only
// ever uses that looked-up pointer -- the array's name is a host-side stand-in, not a device address.
static __device__ __align__(256) unsigned char g_noalloc_scratch[VSV_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};

int noalloc_scratch(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("vsv_encode_windows_noalloc: hipGetDevice: %s", hipGetErrorString(e)); return (int)e; }
    if (dev < 0 || dev >= kMaxDevices) { set_error("vsv_encode_windows_noalloc: device %d out of range", dev); return -1; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));          // this device's copy (it is current)
        if (e != hipSuccess || !p) {
            set_error("vsv_encode_windows_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return e != hipSuccess ? (int)e : -1;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int vsv_init_device(void) {
    void* p = nullptr;
    return noalloc_scratch(&p);
}

extern "C" int vsv_encode_windows_noalloc(const vsv_dims* dims, const float* packed, int which, const float* vol,
                                          const float* ex, long n_windows, int T, float* out, void* stream) {
    if (!dims || n_windows < 0 || T < 0) { set_error("vsv_encode_windows_noalloc: bad arguments"); return -1; }
    const size_t need = vsv_encode_windows_ws_bytes(dims, n_windows, T);
    if (need == 0) { set_error("vsv_encode_windows_noalloc: unsupported model shape"); return -1; }
    if (need > (size_t)VSV_NOALLOC_SCRATCH_BYTES) {
        set_error("vsv_encode_windows_noalloc: %ld windows x %d steps need %zu bytes of scratch, the built-in buffer holds "
                  "%zu (use vsv_encode_windows with a caller workspace)", n_windows, T, need, (size_t)VSV_NOALLOC_SCRATCH_BYTES);
        return -6;
    }
    void* ws = nullptr;
    if (int rc = noalloc_scratch(&ws)) return rc;
    // every word of the workspace that a kernel reads is written earlier in the same call (embedding table, gate
    // pre-activations, hidden states): nothing has to start at zero, whatever the previous call left behind
    return encode_windows_impl(dims, packed, which, vol, ex, n_windows, T, nullptr, nullptr, out, ws,
                               (size_t)VSV_NOALLOC_SCRATCH_BYTES, stream);
}

static int encode_impl(const vsv_dims* dims, const float* packed, int which,
                       const float* surf, long surf_row_stride, long surf_t_stride,
                       const float* ex, long ex_row_stride, long ex_t_stride,
                       long n_rows, int T, const float* state_in, long state_div, float* state_out,
                       float* out, void* ws, size_t ws_bytes, void* stream);

// ---- autoregressive rollouts: a generated day's embedding and first-layer input product are the same at
// every later step of the rollout (they do not depend on the day's position in the window), so they are
// computed once when the day is generated (vsv_encode_pre) and the later steps start from the table ----
extern "C" size_t vsv_encode_pre_ws_bytes(const vsv_dims* dims, long n_rows) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_rows < 0) return 0;
    const long rt = (n_rows + 31) / 32;
    const int kx = lay.enc[0].kx > lay.enc[1].kx ? lay.enc[0].kx : lay.enc[1].kx;
    return align256((size_t)rt * kx * 64 * sizeof(float)) + 256;
}

extern "C" int vsv_encode_pre(const vsv_dims* dims, const float* packed, int which,
                              const float* surf, long surf_row_stride, const float* ex, long ex_row_stride,
                              long n_rows, float* pre, long pre_stride, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_encode_pre")
    if (which < 0 || which > 1) { set_error("vsv_encode_pre: which must be 0 or 1"); return -1; }
    if (n_rows < 0) { set_error("vsv_encode_pre: negative size"); return -1; }
    if (n_rows == 0) return 0;
    if (!packed || !surf || !pre || !ws) { set_error("vsv_encode_pre: null pointer"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_encode_pre: ex_feats_dim > 0 but ex is null"); return -1; }
    const long rt = (n_rows + 31) / 32;
    if (pre_stride < rt * 32) { set_error("vsv_encode_pre: pre_stride must cover the rows padded to 32"); return -1; }
    if (ws_bytes < vsv_encode_pre_ws_bytes(dims, n_rows)) { set_error("vsv_encode_pre: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    const int NT = lay.NTm;
    float* xe = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    DenseChainArgs a = {};
    a.n_layers = e.n_chain;
    for (int i = 0; i < e.n_chain; ++i) fill_dense(a, i, packed, e.conv[i]);
    a.inj_layer = e.n_conv - 1;
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = 1;
    a.in_mode = 0; a.in = surf; a.in_rs = surf_row_stride; a.in_ts = 0; a.in_f = 25;
    a.in_ks = 4 * e.conv[0].nq;
    if (lay.X > 0) { a.inj = ex; a.inj_rs = ex_row_stride; a.inj_ts = 0; a.inj_f0 = e.ex_f0; a.inj_n = lay.X; }
    a.out_mode = 0; a.out = xe; a.out_f = e.E; a.S = 1;
    a.aux_layer = -1;
    int rc = launch_dense_chain(a, st);
    if (rc) return rc;
    const float* w0 = packed + e.lstm[0].w_off;
    const float* b0 = packed + e.lstm[0].b_off;
    if (rt > 0x7fffffffL) { set_error("vsv_encode_pre: too many rows"); return -2; }
    if (NT == 13) hipLaunchKernelGGL(lstm_pre_t_kernel<13>, dim3((unsigned)rt), dim3(64 * 13), 0, st, w0, b0, e.kx >> 2, xe, pre, pre_stride);
    else if (NT == 16) hipLaunchKernelGGL(lstm_pre_t_kernel<16>, dim3((unsigned)rt), dim3(64 * 16), 0, st, w0, b0, e.kx >> 2, xe, pre, pre_stride);
    else { set_error("unsupported LSTM hidden size: %d tiles of 8 units", NT); return -4; }
    return launch_check("lstm_pre_t_kernel");
}

extern "C" size_t vsv_encode_from_pre_ws_bytes(const vsv_dims* dims, long n_rows, int T) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_rows < 0 || T < 0) return 0;
    const long rt = (n_rows + 31) / 32;
    return align256((size_t)rt * T * 4 * lay.NTm * 64 * sizeof(float)) + 256;
}

extern "C" int vsv_encode_from_pre(const vsv_dims* dims, const float* packed, int which,
                                   const float* pre, long pre_stride, long pre_tstride, long n_rows, int T,
                                   const float* state_in, long state_div, float* state_out,
                                   float* out, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_encode_from_pre")
    if (which < 0 || which > 1) { set_error("vsv_encode_from_pre: which must be 0 or 1"); return -1; }
    if (n_rows < 0 || T < 0) { set_error("vsv_encode_from_pre: negative size"); return -1; }
    if (n_rows == 0 || T == 0) return 0;
    if (!packed || !pre || !out || !ws) { set_error("vsv_encode_from_pre: null pointer"); return -1; }
    if (state_in && state_div < 1) { set_error("vsv_encode_from_pre: state_div must be >= 1"); return -1; }
    const long rt = (n_rows + 31) / 32;
    if (pre_stride < rt * 32 || pre_tstride < 1) { set_error("vsv_encode_from_pre: bad table strides"); return -1; }
    if (ws_bytes < vsv_encode_from_pre_ws_bytes(dims, n_rows, T)) { set_error("vsv_encode_from_pre: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    float* hbuf = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    for (int i = 0; i < lay.NL; ++i) { l.w[i] = packed + e.lstm[i].w_off; l.b[i] = packed + e.lstm[i].b_off; }
    l.kx = e.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.pre_t = pre; l.pre_stride = pre_stride; l.pre_tstride = pre_tstride; l.L = lay.L;
    l.hout = hbuf; l.t_out0 = 0;
    l.din.state_in = state_in; l.din.state_div = state_in ? state_div : 1;
    l.state_out = state_out;
    // many rows: both layers and the head in one launch on the one-wave-per-row-tile schedule (enc_seq_kernel;
    // VSV_ESEQ=0 keeps lstm_kernel + the dense chain, VSV_ESEQ_MIN the row-tile threshold)
    static const bool eseq_off = [] { const char* v = getenv("VSV_ESEQ"); return v && v[0] == '0'; }();
    static const long eseq_min = [] { const char* v = getenv("VSV_ESEQ_MIN"); return v ? atol(v) : 1024L; }();
    if (!eseq_off && rt >= eseq_min && lay.NL == 2 && lay.NTm == 13 && !lay.gru && !state_out && e.head.nq == 13 &&
        e.head.nt == 1 && e.head.n_relu == 0 && e.nh <= 32) {
        fill_dense(l.head, 0, packed, e.head);
        l.head.n_layers = 1; l.head.out = out; l.head.out_f = e.nh; l.head.n_rows = n_rows; l.head.T = T;
        l.hout = nullptr;
        constexpr int wpb = 4;
        const size_t lds = (size_t)wpb * (size_t)(8 * 13) * 64 * sizeof(float);
        int rc = set_lds(enc_seq_kernel<13>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(enc_seq_kernel<13>, dim3((unsigned)((rt + wpb - 1) / wpb)), dim3(64 * wpb), lds, st, l);
        return launch_check("enc_seq_kernel");
    }
    int rc = launch_lstm<0>(l, lay.NTm, lay.NL, false, st);
    if (rc) return rc;
    DenseChainArgs h = {};
    h.n_layers = 1;
    fill_dense(h, 0, packed, e.head);
    h.n_rows = n_rows; h.n_rowtiles = rt; h.T = T;
    h.in_mode = 1; h.in = hbuf; h.in_ks = 4 * lay.NTm;
    h.out_mode = 1; h.out = out; h.out_f = e.nh; h.S = 1;
    h.aux_layer = -1;
    return launch_dense_chain(h, st);
}

extern "C" int vsv_encode(const vsv_dims* dims, const float* packed, int which,
                          const float* surf, long surf_row_stride, long surf_t_stride,
                          const float* ex, long ex_row_stride, long ex_t_stride,
                          long n_rows, int T, float* out, void* ws, size_t ws_bytes, void* stream) {
    return encode_impl(dims, packed, which, surf, surf_row_stride, surf_t_stride, ex, ex_row_stride, ex_t_stride,
                       n_rows, T, nullptr, 1, nullptr, out, ws, ws_bytes, stream);
}

extern "C" int vsv_encode_state(const vsv_dims* dims, const float* packed, int which,
                                const float* surf, long surf_row_stride, long surf_t_stride,
                                const float* ex, long ex_row_stride, long ex_t_stride,
                                long n_rows, int T, const float* state_in, long state_div, float* state_out,
                                float* out, void* ws, size_t ws_bytes, void* stream) {
    if (state_in && state_div < 1) { set_error("vsv_encode_state: state_div must be >= 1"); return -1; }
    return encode_impl(dims, packed, which, surf, surf_row_stride, surf_t_stride, ex, ex_row_stride, ex_t_stride,
                       n_rows, T, state_in, state_div, state_out, out, ws, ws_bytes, stream);
}

static int encode_impl(const vsv_dims* dims, const float* packed, int which,
                       const float* surf, long surf_row_stride, long surf_t_stride,
                       const float* ex, long ex_row_stride, long ex_t_stride,
                       long n_rows, int T, const float* state_in, long state_div, float* state_out,
                       float* out, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_encode")
    if (which < 0 || which > 1) { set_error("vsv_encode: which must be 0 or 1"); return -1; }
    if (n_rows < 0 || T < 0) { set_error("vsv_encode: negative size"); return -1; }
    if (n_rows == 0 || T == 0) return 0;                          // nothing to do (empty outputs have no address)
    if (!packed || !surf || !out || !ws) { set_error("vsv_encode: null pointer"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_encode: ex_feats_dim > 0 but ex is null"); return -1; }
    if (ws_bytes < vsv_encode_ws_bytes(dims, n_rows, T)) { set_error("vsv_encode: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    const long rt = (n_rows + 31) / 32;
    float* xe = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    float* hbuf = (float*)((char*)xe + align256((size_t)rt * T * e.kx * 64 * sizeof(float)));

    // 1. conv embedding of every (row, t): surface -> [conv+ReLU]* -> [emb | ex] in tile-B layout
    DenseChainArgs a = {};
    a.n_layers = e.n_chain;
    for (int i = 0; i < e.n_chain; ++i) fill_dense(a, i, packed, e.conv[i]);
    a.inj_layer = e.n_conv - 1;
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = T;
    a.in_mode = 0; a.in = surf; a.in_rs = surf_row_stride; a.in_ts = surf_t_stride; a.in_f = 25;
    a.in_ks = 4 * e.conv[0].nq;
    if (lay.X > 0) { a.inj = ex; a.inj_rs = ex_row_stride; a.inj_ts = ex_t_stride; a.inj_f0 = e.ex_f0; a.inj_n = lay.X; }
    a.out_mode = 0; a.out = xe; a.out_f = e.E; a.S = 1;
    a.aux_layer = -1;
    int rc = launch_dense_chain(a, st);
    if (rc) return rc;

    // 2. LSTM over time, top-layer h of every step -> hbuf (tile-B)
    LstmSeqArgs l = {};
    l.gru = lay.gru;
    for (int i = 0; i < lay.NL; ++i) { l.w[i] = packed + e.lstm[i].w_off; l.b[i] = packed + e.lstm[i].b_off; }
    l.kx = e.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.x_tileb = xe; l.L = lay.L;
    l.hout = hbuf; l.t_out0 = 0;
    l.din.state_in = state_in; l.din.state_div = state_in ? state_div : 1;
    l.state_out = state_out;
    rc = launch_lstm<0>(l, lay.NTm, lay.NL, false, st);
    if (rc) return rc;

    // 3. linear head on every step
    DenseChainArgs h = {};
    h.n_layers = 1;
    fill_dense(h, 0, packed, e.head);
    h.n_rows = n_rows; h.n_rowtiles = rt; h.T = T;
    h.in_mode = 1; h.in = hbuf; h.in_ks = 4 * lay.NTm;
    h.out_mode = 1; h.out = out; h.out_f = e.nh; h.S = 1;
    h.aux_layer = -1;
    return launch_dense_chain(h, st);
}

// mc_decode1_kernel serves the Monte-Carlo one-day draw only: one step from a shared per-day state with the
// first-step gates given, Philox (or no) noise, nothing but the decoded surfaces [and ex-features] wanted,
// the production topology (two LSTM layers, 16 gate tiles, four head layers, the first of them full K).
// VSV_MC1=0 keeps the four-waves-per-row-tile kernel (A/B runs).
static bool mc_decode1_applies(const LstmSeqArgs& l, const Layout& lay, int T, int t_out0, const float* state_out,
                               const float* z_out) {
    static const bool off = [] { const char* e = getenv("VSV_MC1"); return e && e[0] == '0'; }();
    if (off) return false;
    const vsv_decode_in& d = l.din;
    const DenseChainArgs& hd = l.head;
    if (T != 1 || t_out0 != 0 || state_out || z_out || d.gate_out) return false;
    if (!d.gate_in || !d.state_in || d.eps_in || d.ctx_T > 0 || d.z_T > 0) return false;
    if (lay.NL != 2 || lay.NTd != 16 || lay.gru || l.kx > 64) return false;
    if (hd.n_layers != 4 || hd.layer[0].nt != 4 || hd.layer[0].nq != 16) return false;
    for (int i = 1; i < 4; ++i)
        if (hd.layer[i].nt < 1 || hd.layer[i].nt > 4 || hd.layer[i].nq != 16) return false;
    for (int i = 0; i < 4; ++i)
        for (int t = 0; t < hd.layer[i].nt; ++t)                   // (the banded tiles clamp into non-empty bands)
            if (hd.layer[i].nqb[t] < 1 || hd.layer[i].q0[t] < 0 || hd.layer[i].q0[t] + hd.layer[i].nqb[t] > 16) return false;
    return true;
}

// dec_seq_kernel: the same one-wave schedule over T steps, for launches with enough row tiles to give every SIMD of
// the chip a wave of its own (VSV_DSEQ_MIN, default 1024 tiles = 32768 rows; below that lstm_kernel's four or
// sixteen waves per row tile finish a short launch sooner).  VSV_DSEQ=0 keeps lstm_kernel (A/B runs).
static bool dec_seq_applies(const LstmSeqArgs& l, const Layout& lay, int T, int t_out0, const float* state_out,
                            const float* z_out) {
    static const bool off = [] { const char* e = getenv("VSV_DSEQ"); return e && e[0] == '0'; }();
    static const long min_tiles = [] { const char* e = getenv("VSV_DSEQ_MIN"); return e ? atol(e) : 1024L; }();
    if (off || l.n_rowtiles < min_tiles) return false;
    const vsv_decode_in& d = l.din;
    const DenseChainArgs& hd = l.head;
    if (T < 1 || t_out0 >= T || state_out || z_out || d.gate_out || l.sv_gates) return false;
    if (lay.NL != 2 || lay.NTd != 16 || lay.gru || l.kx > 16 || (l.kx & 3) || lay.L > 16) return false;
    if (hd.n_layers != 4 || hd.layer[0].nt != 4 || hd.layer[0].nq != 16) return false;
    for (int i = 1; i < 4; ++i)
        if (hd.layer[i].nt < 1 || hd.layer[i].nt > 4 || hd.layer[i].nq != 16) return false;
    for (int i = 0; i < 4; ++i)
        for (int t = 0; t < hd.layer[i].nt; ++t)                   // (band_tile clamps into non-empty bands)
            if (hd.layer[i].nqb[t] < 1 || hd.layer[i].q0[t] < 0 || hd.layer[i].q0[t] + hd.layer[i].nqb[t] > 16) return false;
    return true;
}

extern "C" size_t vsv_decode_ws_bytes(const vsv_dims* dims, long n_rows, int T, int t_out0) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_rows < 0 || T < 0 || t_out0 < 0) return 0;
    return 256;                                                // the head is fused: no intermediate buffer
}

extern "C" int vsv_decode(const vsv_dims* dims, const float* packed, const vsv_decode_in* in,
                          long n_rows, int T, int t_out0, float* state_out, float* z_out,
                          float* surf_out, int out_mode, float* ex_out,
                          void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_decode")
    if (!packed || !in) { set_error("vsv_decode: null pointer"); return -1; }
    if (n_rows < 0 || T < 0 || t_out0 < 0 || t_out0 > T) { set_error("vsv_decode: bad sizes"); return -1; }
    if (n_rows == 0 || T == 0) return 0;
    if (n_rows > 0x7fffffffL) { set_error("vsv_decode: more than 2^31-1 rows in one call"); return -2; }
    const int touts = T - t_out0;
    if (touts > 0 && (!surf_out || !ws)) { set_error("vsv_decode: surf_out/ws null"); return -1; }
    if (touts > 0 && ws_bytes < vsv_decode_ws_bytes(dims, n_rows, T, t_out0)) { set_error("vsv_decode: workspace too small"); return -1; }
    if (in->ctx_T > 0 && (!in->ctx || in->ctx_div < 1)) { set_error("vsv_decode: ctx missing"); return -1; }
    if (in->z_T > 0 && (!in->zb || in->z_div < 1)) { set_error("vsv_decode: zb missing"); return -1; }
    if (in->ctx_T > T || in->z_T > T) { set_error("vsv_decode: ctx_T/z_T exceed T"); return -1; }
    const bool needs_noise = (in->z_T < T && in->std0 != 0.0f) || (in->lv != nullptr && in->z_T > 0);
    if (needs_noise) {
        if (in->eps_in) {
            if (in->eps_t0 < 0 || in->eps_T < 1) { set_error("vsv_decode: bad eps_T/eps_t0"); return -1; }
            const int first = in->lv ? 0 : in->z_T;
            if (in->eps_t0 > first || T - in->eps_t0 > in->eps_T) { set_error("vsv_decode: eps_in does not cover the noisy steps"); return -1; }
        } else if (in->S < 1 || in->eps_t0 < 0 || in->eps_t0 > (in->lv ? 0 : in->z_T)) {
            set_error("vsv_decode: Philox mode needs S >= 1 and eps_t0 <= first noisy step"); return -1;
        }
    }
    if (in->state_in && in->state_div < 1) { set_error("vsv_decode: state_div must be >= 1"); return -1; }
    if (in->gate_in && !in->state_in) { set_error("vsv_decode: gate_in comes with the state_in it was computed from"); return -1; }
    if (out_mode < 0 || out_mode >= 2 + lay.Q) { set_error("vsv_decode: out_mode must be 0, 1 or VSV_OUT_CHANNEL(c) with c < n_quantiles"); return -1; }
    if (out_mode >= 1 && touts > 0 && (in->S < 1 || n_rows % in->S != 0)) { set_error("vsv_decode: sample-minor output needs n_rows %% S == 0"); return -1; }
    const int out_ch = out_mode >= 2 ? out_mode - 2 : -1;         // sample-minor, ONE decoder channel
    hipStream_t st = (hipStream_t)stream;
    const DecLayout& dc = lay.dec;
    const long rt = (n_rows + 31) / 32;

    LstmSeqArgs l = {};
    l.gru = lay.gru;
    for (int i = 0; i < lay.NL; ++i) { l.w[i] = packed + dc.lstm[i].w_off; l.b[i] = packed + dc.lstm[i].b_off; }
    l.kx = dc.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.L = lay.L; l.Lc = lay.Lc; l.din = *in; l.z_out = z_out;
    if (l.din.ctx_div < 1) l.din.ctx_div = 1;
    if (l.din.z_div < 1) l.din.z_div = 1;
    if (l.din.S < 1) l.din.S = 1;
    if (l.din.state_div < 1) l.din.state_div = 1;
    l.state_out = state_out;
    l.t_out0 = t_out0;

    // the decoder head: Linear -> ConvT+ReLU x n -> Conv [, ex Linear x2 riding in spare features]
    DenseChainArgs& a = l.head;
    // (generation runs Linear . first ConvT, which have no activation between them, as the folded layer)
    a.n_layers = dc.n_head - 1;
    for (int i = 0; i < a.n_layers; ++i)
        fill_dense(a, i, packed, i < dc.fold_at ? dc.head[i] : (i == dc.fold_at ? dc.head_fold : dc.head[i + 1]));
    if (out_ch >= 0) fill_dense(a, a.n_layers - 1, packed, dc.head_ch[out_ch]);   // (the other channels are not computed)
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = touts;
    a.out_mode = out_mode == 0 ? 1 : 2; a.out = surf_out; a.out_f = out_ch >= 0 ? 25 : dc.n_surf; a.S = l.din.S;
    a.aux_layer = (lay.X > 0 && ex_out) ? dc.ex_layer - 1 : -1;
    a.aux = ex_out; a.aux_f0 = dc.ex_f0; a.aux_n = lay.X;

    if (a.n_layers < 2 || a.layer[0].nq != lay.NTd || dc.ex_layer != (lay.X > 0 ? dc.fold_at + 1 : -1)) {
        set_error("vsv_decode: unexpected head layout"); return -4;
    }
    hipEvent_t ev0 = g_prof_ev[0], ev1 = g_prof_ev[1];
    g_prof_ev[0] = g_prof_ev[1] = nullptr;
    if (ev0) (void)hipEventRecord(ev0, st);
    int rc;
    if (mc_decode1_applies(l, lay, T, t_out0, state_out, z_out)) {
        // the Monte-Carlo one-day draw: one wave per 32 rows, no barriers (mc_decode1_kernel)
        constexpr int wpb = 4;                                     // (1, 2 or 4 waves per workgroup measure the same)
        const size_t lds = (size_t)wpb * 64 * 64 * sizeof(float);
        rc = set_lds(mc_decode1_kernel<16>, lds);
        if (!rc) {
            hipLaunchKernelGGL(mc_decode1_kernel<16>, dim3((unsigned)((rt + wpb - 1) / wpb)), dim3(64 * wpb), lds, st, l);
            rc = launch_check("mc_decode1_kernel");
        }
    } else if (dec_seq_applies(l, lay, T, t_out0, state_out, z_out)) {
        // many rows, T steps: one wave per 32 rows and SIMD, cell state in registers, no barriers (dec_seq_kernel)
        static const int wpb = [] { const char* e = getenv("VSV_DSEQ_WPB"); const int v = e ? atoi(e) : 4; return (v == 1 || v == 2) ? v : 4; }();
        const size_t lds = (size_t)wpb * (size_t)(l.kx + 8 * 16) * 64 * sizeof(float);
        rc = set_lds(dec_seq_kernel<16>, lds);
        if (!rc) {
            hipLaunchKernelGGL(dec_seq_kernel<16>, dim3((unsigned)((rt + wpb - 1) / wpb)), dim3(64 * wpb), lds, st, l);
            rc = launch_check("dec_seq_kernel");
        }
    } else {
        rc = launch_lstm<1>(l, lay.NTd, lay.NL, touts > 0, st);
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    return rc;
}

#ifdef VSV_STAMPS
extern "C" int vsv_debug_stamps(unsigned long long* host_out, size_t n) {
    if (n > sizeof(g_stamps) / 8) n = sizeof(g_stamps) / 8;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), n * 8);
}
#endif

extern "C" void vsv_profile_next_decode(void* ev_start, void* ev_stop) {
    g_prof_ev[0] = (hipEvent_t)ev_start;
    g_prof_ev[1] = (hipEvent_t)ev_stop;
}

template <int E>
static void launch_percentile_wave(const float* in, long n_vec, int S, const PercArgs& pa, int inner, long group_stride,
                                   float* out, hipStream_t st) {
    hipLaunchKernelGGL(percentile_wave_kernel<E>, dim3((unsigned)((n_vec + 3) / 4)), dim3(256), 0, st, in, n_vec, S, pa,
                       inner, group_stride, out);
}

extern "C" int vsv_percentile(const float* in, long n_vec, int S, const double* q_host, int nq,
                              int inner, long group_stride, float* out, void* stream) {
    if (!in || !out || !q_host) { set_error("vsv_percentile: null pointer"); return -1; }
    if (nq < 1 || nq > 8) { set_error("vsv_percentile: nq must be in [1,8]"); return -1; }
    if (S < 1 || S > 16384) { set_error("vsv_percentile: S must be in [1,16384]"); return -1; }
    if (n_vec < 0 || inner < 1 || n_vec % inner != 0) { set_error("vsv_percentile: n_vec must be a multiple of inner"); return -1; }
    if (group_stride < inner) { set_error("vsv_percentile: group_stride must be >= inner"); return -1; }
    if (n_vec == 0) return 0;
    if ((n_vec + 3) / 4 > 0x7fffffffL) { set_error("vsv_percentile: too many vectors"); return -2; }
    PercArgs pa = {};
    pa.nq = nq;
    for (int i = 0; i < nq; ++i) {
        if (!(q_host[i] >= 0.0 && q_host[i] <= 100.0)) { set_error("vsv_percentile: percentile out of [0,100]"); return -1; }
        const double pos = q_host[i] / 100.0 * (S - 1);
        int lo = (int)floor(pos);
        if (lo > S - 1) lo = S - 1;
        pa.lo[i] = lo;
        pa.fr[i] = (float)(pos - lo);
    }
    hipStream_t st = (hipStream_t)stream;
    if (S <= 1024) {
        // wavefront-level selection: a wave per vector, sorted in registers
        if (S <= 64) launch_percentile_wave<1>(in, n_vec, S, pa, inner, group_stride, out, st);
        else if (S <= 128) launch_percentile_wave<2>(in, n_vec, S, pa, inner, group_stride, out, st);
        else if (S <= 256) launch_percentile_wave<4>(in, n_vec, S, pa, inner, group_stride, out, st);
        else if (S <= 512) launch_percentile_wave<8>(in, n_vec, S, pa, inner, group_stride, out, st);
        else launch_percentile_wave<16>(in, n_vec, S, pa, inner, group_stride, out, st);
        return launch_check("percentile_wave_kernel");
    }
    int P2 = 2;
    while (P2 < S) P2 <<= 1;
    static const bool wg_off = [] { const char* e = getenv("VSV_PERCENTILE_WG"); return e && e[0] == '0'; }();
    if (!wg_off && n_vec <= 0x7fffffffL) {
        // workgroup-level selection: P2 / 2048 waves per vector, sorted in registers, six LDS exchanges at most
        const int W = P2 / (64 * kPwgE);
        const size_t ldsw = (size_t)P2 * sizeof(float);
        if (int rcw = set_lds(percentile_wg_kernel, ldsw)) return rcw;
        if (n_vec > 0)
            hipLaunchKernelGGL(percentile_wg_kernel, dim3((unsigned)n_vec), dim3(64 * W), ldsw, st, in, n_vec, S, W, pa, inner,
                               group_stride, out);
        return launch_check("percentile_wg_kernel");
    }
    const size_t lds = (size_t)P2 * sizeof(float);
    int rc = set_lds(percentile_kernel, lds);
    if (rc) return rc;
    const long maxb = 256L * 16;
    const unsigned blocks = (unsigned)(n_vec < maxb ? n_vec : maxb);
    hipLaunchKernelGGL(percentile_kernel, dim3(blocks), dim3(256), lds, st, in, n_vec, S, P2, pa, inner, group_stride, out);
    return launch_check("percentile_kernel");
}

// factor (0 or 1/(1-p)) of every (row, t, unit) as the training kernels apply it: out (n_rows, T, hidden)
__global__ void __launch_bounds__(256) dropout_mask_kernel(DropArgs d, int layer, long n_rows, int T, int hidden,
                                                           float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;           // one (row, t, tile, hh) group per thread
    const int groups = 2 * ((hidden + 7) / 8);
    if (i >= n_rows * T * groups) return;
    const int g = (int)(i % groups);
    const long rt = i / groups;
    float f[4];
    dropout_factors(d, (uint32_t)(rt / T), (uint32_t)(rt % T), g >> 1, g & 1, layer, f);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int u = 8 * (g >> 1) + 2 * jj + (g & 1);
        if (u < hidden) out[rt * hidden + u] = f[jj];
    }
}

extern "C" int vsv_dropout_mask(float p, uint64_t seed, int id, int layer, long n_rows, int T, int hidden,
                                float* out, void* stream) {
    if (!out) { set_error("vsv_dropout_mask: null pointer"); return -1; }
    if (!(p >= 0.0f && p < 1.0f) || n_rows < 0 || T < 0 || hidden < 1) { set_error("vsv_dropout_mask: bad arguments"); return -1; }
    const long total = n_rows * T * (2 * ((hidden + 7) / 8));
    if (total == 0) return 0;
    const DropArgs d = {p, 1.0f / (1.0f - p), (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)id, nullptr};
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d, layer, n_rows, T, hidden, out);
    return launch_check("dropout_mask_kernel");
}

extern "C" int vsv_gmm_sample(const float* means, const float* chol, const float* cdf, int K, int L, long n,
                              const int* comp_in, const float* eps_in, uint64_t seed, float* z_out, int* comp_out,
                              void* stream) {
    if (!means || !chol || !z_out || (!comp_in && !cdf)) { set_error("vsv_gmm_sample: null pointer"); return -1; }
    if (K < 1 || L < 1 || L > kGmmMaxL || n < 0) { set_error("vsv_gmm_sample: need K >= 1, 1 <= L <= %d, n >= 0", kGmmMaxL); return -1; }
    if (n == 0) return 0;
    const long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffL) { set_error("vsv_gmm_sample: too many samples"); return -2; }
    hipLaunchKernelGGL(gmm_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, means, chol, cdf, K, L, n,
                       comp_in, eps_in, seed, z_out, comp_out);
    return launch_check("gmm_sample_kernel");
}

extern "C" int vsv_philox_normal(uint64_t seed, long day0, long n_days, long S, int step0, int n_steps,
                                 int L, float* out, void* stream) {
    if (!out) { set_error("vsv_philox_normal: null pointer"); return -1; }
    if (n_days < 0 || S < 1 || n_steps < 0 || L < 1) { set_error("vsv_philox_normal: bad sizes"); return -1; }
    const long total = n_days * S * n_steps * L;
    if (total == 0) return 0;
    const long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffL) { set_error("vsv_philox_normal: too many elements"); return -2; }
    hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       seed, day0, S, step0, n_steps, L, total, out);
    return launch_check("philox_normal_kernel");
}

// ---------------------------------------------------------------------------
// training entry points
// ---------------------------------------------------------------------------
// The launches of one encoder stack's training forward: conv embedding chain, per layer the input product for all
// steps + the recurrent launch, latent / compression head.
struct TrainEncPlan {
    DenseChainArgs emb, head;
    LstmSeqArgs l;
    const EncLayout* e;
    long rt;
    int T;
};
static int plan_train_encode(const Layout& lay, const float* packed, int which,
                             const float* surf, long surf_row_stride, long surf_t_stride,
                             const float* ex, long ex_row_stride, long ex_t_stride,
                             long n_rows, int T, float* out, const vsv_train_save* sv, TrainEncPlan& p,
                             bool split_planes = false) {
    if (!packed || !surf || !out || !sv) { set_error("vsv_train_encode: null pointer"); return -1; }
    if (!sv->emb_acts || !sv->xe || !sv->gates || !sv->c || !sv->h || !sv->hout) { set_error("vsv_train_encode: incomplete save set"); return -1; }
    if (which < 0 || which > 1) { set_error("vsv_train_encode: which must be 0 or 1"); return -1; }
    if (n_rows < 1 || T < 1) { set_error("vsv_train_encode: empty batch"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_train_encode: ex_feats_dim > 0 but ex is null"); return -1; }
    const EncLayout& e = lay.enc[which];
    const long rt = (n_rows + 31) / 32;
    p.e = &e; p.rt = rt; p.T = T;

    DenseChainArgs& a = p.emb;
    a = DenseChainArgs();
    a.n_layers = e.n_chain;
    for (int i = 0; i < e.n_chain; ++i) fill_dense(a, i, packed, e.conv[i]);
    a.inj_layer = e.n_conv - 1;
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = T;
    a.in_mode = 0; a.in = surf; a.in_rs = surf_row_stride; a.in_ts = surf_t_stride; a.in_f = 25;
    a.in_ks = 4 * e.conv[0].nq;
    if (lay.X > 0) { a.inj = ex; a.inj_rs = ex_row_stride; a.inj_ts = ex_t_stride; a.inj_f0 = e.ex_f0; a.inj_n = lay.X; }
    a.out_mode = 0; a.out = sv->xe; a.out_f = e.E; a.S = 1;
    a.aux_layer = -1;
    a.save = sv->emb_acts;

    LstmSeqArgs& l = p.l;
    l = LstmSeqArgs();
    l.gru = lay.gru;
    for (int i = 0; i < lay.NL; ++i) { l.w[i] = packed + e.lstm[i].w_off; l.b[i] = packed + e.lstm[i].b_off; }
    l.kx = e.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.x_tileb = sv->xe; l.L = lay.L;
    l.hout = sv->hout; l.t_out0 = 0;
    l.sv_gates = sv->gates; l.sv_c = sv->c; l.sv_h = sv->h;
    if (int rcd = fill_dropout(l, sv, lay.NL, "vsv_train_encode")) return rcd;

    DenseChainArgs& h = p.head;
    h = DenseChainArgs();
    h.n_layers = 1;
    fill_dense(h, 0, packed, e.head);
    h.n_rows = n_rows; h.n_rowtiles = rt; h.T = T;
    h.in_mode = 1; h.in = sv->hout; h.in_ks = 4 * lay.NTm;
    h.out_mode = 1; h.out = out; h.out_f = e.nh; h.S = 1;
    h.aux_layer = -1;
    if (split_planes && which == 0) {
        // posterior encoder: z_mean and z_log_var as two planes (n_rows, T, L) instead of one (n_rows, T, 2L) --
        // the decoder reads them as separate tensors, and a strided slice of the pair would have to be copied out
        h.out_f = lay.L;
        h.aux = out + (size_t)n_rows * T * lay.L; h.aux_f0 = lay.L; h.aux_n = lay.L; h.aux_layer = 0;
    }
    return 0;
}
// layer L of a per-layer forward: the input product's arguments and the recurrent launch's
static void train_encode_layer(const Layout& lay, const TrainEncPlan& p, const vsv_train_save* sv, int L, PreArgs& pre, LstmSeqArgs& r) {
    const int NT = lay.NTm;
    const LstmSeqArgs& l = p.l;
    const long gs = (long)lay.NL * 16 * NT * 64, hs = (long)lay.NL * 4 * NT * 64;
    pre.x = (L == 0) ? sv->xe : (l.sv_hd ? l.sv_hd : sv->h);
    pre.x_stride = (L == 0) ? (long)p.e->kx * 64 : (l.sv_hd ? (long)4 * NT * 64 : hs);
    pre.nq_in = (L == 0) ? p.e->kx >> 2 : NT;
    pre.w = l.w[L]; pre.b = l.b[L];
    pre.gates = sv->gates + (size_t)L * 16 * NT * 64; pre.g_stride = gs;
    pre.n_inst = p.rt * p.T; pre.ipw = 0;
    r = l;
    r.w[0] = l.w[L]; r.b[0] = l.b[L]; r.kx = 4 * pre.nq_in;
    r.rec = 1; r.sv_nl = 2; r.sv_l0 = L;
    r.hout = (L == 1) ? sv->hout : nullptr;
}
static int run_train_encode(const Layout& lay, const TrainEncPlan& p, const vsv_train_save* sv, hipStream_t st) {
    int rc = launch_dense_chain(p.emb, st);
    if (rc) return rc;
    if (per_layer_forward(lay.NL, p.rt)) {
        for (int L = 0; L < 2 && !rc; ++L) {
            PreArgs pre;
            LstmSeqArgs r;
            train_encode_layer(lay, p, sv, L, pre, r);
            rc = launch_lstm_pre(lay.NTm, pre.w, pre.b, pre.nq_in, pre.x, pre.x_stride, pre.gates, pre.g_stride, pre.n_inst, st);
            if (rc) return rc;
            rc = launch_lstm<0>(r, lay.NTm, 1, false, st);
        }
    } else {
        rc = launch_lstm<0>(p.l, lay.NTm, lay.NL, false, st);
    }
    if (rc) return rc;
    return launch_dense_chain(p.head, st);
}

extern "C" int vsv_train_encode(const vsv_dims* dims, const float* packed, int which,
                                const float* surf, long surf_row_stride, long surf_t_stride,
                                const float* ex, long ex_row_stride, long ex_t_stride,
                                long n_rows, int T, float* out, const vsv_train_save* sv, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_train_encode")
    TrainEncPlan p;
    if (int rc = plan_train_encode(lay, packed, which, surf, surf_row_stride, surf_t_stride, ex, ex_row_stride, ex_t_stride,
                                   n_rows, T, out, sv, p)) return rc;
    return run_train_encode(lay, p, sv, (hipStream_t)stream);
}

// Both encoders of a training step in one call: the posterior encoder (which = 0) over T0 steps and the context
// encoder (which = 1) over T1 steps of the same batch.  With few row tiles (a batch of 64: sixteen 4-row tiles per
// stack) every launch of a stack has a twin in the other stack, and each pair goes out as ONE launch whose
// blockIdx.y picks the stack -- six launches instead of twelve, the two stacks side by side on the chip without a
// second stream.  Larger batches (or other memories) run the two stacks one after the other on `stream`: same
// results either way (the kernels and their arguments are those of vsv_train_encode).
extern "C" int vsv_train_encode_pair(const vsv_dims* dims, const float* packed,
                                     const float* surf, long surf_row_stride, long surf_t_stride,
                                     const float* ex, long ex_row_stride, long ex_t_stride,
                                     long n_rows, int T0, int T1, float* out0, float* out1,
                                     const vsv_train_save* sv0, const vsv_train_save* sv1, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_train_encode_pair")
    hipStream_t st = (hipStream_t)stream;
    TrainEncPlan p[2];
    const int Ts[2] = {T0, T1};
    float* outs[2] = {out0, out1};
    const vsv_train_save* svs[2] = {sv0, sv1};
    for (int k = 0; k < 2; ++k)
        if (int rc = plan_train_encode(lay, packed, k, surf, surf_row_stride, surf_t_stride, ex, ex_row_stride, ex_t_stride,
                                       n_rows, Ts[k], outs[k], svs[k], p[k], true)) return rc;
    static const bool off = [] { const char* e = getenv("VSV_ENC_PAIR"); return e && e[0] == '0'; }();
    const long rt = p[0].rt;
    bool pair = !off && lay.NL == 2 && !lay.gru && lay.NTm == 13 && per_layer_forward(lay.NL, rt) && 8 * rt <= 128;
    for (int k = 0; k < 2 && pair; ++k) {
        PreArgs pre;
        LstmSeqArgs r;
        train_encode_layer(lay, p[k], svs[k], 0, pre, r);
        pair = lstm_rec16_applies(r, 1) && lstm_rec4_applies(r) && rt * Ts[k] <= 512 && p[k].emb.in_ks <= kDenseSlots;
    }
    if (!pair) {
        for (int k = 0; k < 2; ++k)
            if (int rc = run_train_encode(lay, p[k], svs[k], st)) return rc;
        return 0;
    }
    const long nmax = rt * (Ts[0] > Ts[1] ? Ts[0] : Ts[1]);
    const size_t lds1 = (size_t)kDenseSlots * 64 * sizeof(float);
    hipLaunchKernelGGL(dense_chain_wg_pair_kernel, dim3((unsigned)nmax, 2), dim3(256), lds1, st, p[0].emb, p[1].emb);
    if (int rc = launch_check("dense_chain_wg_pair_kernel")) return rc;
    for (int L = 0; L < 2; ++L) {
        PreArgs pre[2];
        LstmSeqArgs r[2];
        unsigned grid = 0;
        for (int k = 0; k < 2; ++k) {
            train_encode_layer(lay, p[k], svs[k], L, pre[k], r[k]);
            int ipw = (int)((pre[k].n_inst + 255) / 256);
            pre[k].ipw = ipw < 1 ? 1 : (ipw > 64 ? 64 : ipw);
            const unsigned g = (unsigned)((pre[k].n_inst + pre[k].ipw - 1) / pre[k].ipw);
            grid = g > grid ? g : grid;
        }
        hipLaunchKernelGGL(lstm_pre_pair_kernel<13>, dim3(grid, 2), dim3(64 * 13), 0, st, pre[0], pre[1]);
        if (int rc = launch_check("lstm_pre_pair_kernel")) return rc;
        const size_t lds = (size_t)2 * 1 * 4 * 144 * sizeof(float);
        for (int k = 0; k < 2; ++k)
            if (int rc = rec_spans_fit(r[k], 13)) return rc;
        hipLaunchKernelGGL((lstm_rec4_pair_kernel<13, true, 1>), dim3((unsigned)(8 * rt), 2), dim3(512), lds, st, r[0], r[1]);
        if (int rc = launch_check("lstm_rec4_pair_kernel")) return rc;
    }
    hipLaunchKernelGGL(dense_chain_wg_pair_kernel, dim3((unsigned)nmax, 2), dim3(256), lds1, st, p[0].head, p[1].head);
    return launch_check("dense_chain_wg_pair_kernel");
}

extern "C" int vsv_train_decode(const vsv_dims* dims, const float* packed, const vsv_decode_in* in,
                                long n_rows, int T, int t_out0, float* z_out, float* surf_out, float* ex_out,
                                const vsv_train_save* sv, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_train_decode")
    if (!packed || !in || !sv || !surf_out) { set_error("vsv_train_decode: null pointer"); return -1; }
    if (!sv->xe || !sv->gates || !sv->c || !sv->h || !sv->hout || !sv->head_acts) { set_error("vsv_train_decode: incomplete save set"); return -1; }
    if (n_rows < 1 || n_rows > 0x7fffffffL || T < 1 || t_out0 < 0 || t_out0 >= T) { set_error("vsv_train_decode: bad sizes"); return -1; }
    if (in->ctx_T > 0 && (!in->ctx || in->ctx_div < 1)) { set_error("vsv_train_decode: ctx missing"); return -1; }
    if (in->z_T > 0 && (!in->zb || in->z_div < 1)) { set_error("vsv_train_decode: zb missing"); return -1; }
    if (in->ctx_T > T || in->z_T > T) { set_error("vsv_train_decode: ctx_T/z_T exceed T"); return -1; }
    if (in->state_in || in->gate_in || in->gate_out) { set_error("vsv_train_decode: training starts from the zero state"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const DecLayout& dc = lay.dec;
    const long rt = (n_rows + 31) / 32;
    const int touts = T - t_out0;

    LstmSeqArgs l = {};
    l.gru = lay.gru;
    for (int i = 0; i < lay.NL; ++i) { l.w[i] = packed + dc.lstm[i].w_off; l.b[i] = packed + dc.lstm[i].b_off; }
    l.kx = dc.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.L = lay.L; l.Lc = lay.Lc; l.din = *in; l.z_out = z_out;
    if (l.din.ctx_div < 1) l.din.ctx_div = 1;
    if (l.din.z_div < 1) l.din.z_div = 1;
    if (l.din.S < 1) l.din.S = 1;
    l.din.state_div = 1;
    l.hout = sv->hout; l.t_out0 = t_out0;
    l.sv_gates = sv->gates; l.sv_c = sv->c; l.sv_h = sv->h; l.sv_x = sv->xe;
    if (int rcd = fill_dropout(l, sv, lay.NL, "vsv_train_decode")) return rcd;
    int rc;
    if (per_layer_forward(lay.NL, rt)) {
        const int NT = lay.NTd;
        static const bool l0_fused = [] { const char* e = getenv("VSV_DEC_L0_PRE"); return e && e[0] == '0'; }();
        if (l0_fused || rt * T * dc.kx * 64 > 0x7fffffffL * 256) {
            // layer 0 keeps its (narrow) input product: the latent draw is assembled inside the kernel
            LstmSeqArgs r0 = l;
            r0.sv_nl = 2; r0.sv_l0 = 0; r0.hout = nullptr;
            rc = launch_lstm<1>(r0, NT, 1, false, st);
        } else {
            // input assembly, pre-activation GEMM and the recurrent-only launch, like layer 1
            const long total = rt * T * dc.kx * 64;
            hipLaunchKernelGGL(decoder_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, l);
            rc = launch_check("decoder_input_kernel");
            if (rc) return rc;
            rc = launch_lstm_pre(NT, l.w[0], l.b[0], dc.kx >> 2, sv->xe, (long)dc.kx * 64, sv->gates, (long)lay.NL * 16 * NT * 64,
                                 rt * T, st);
            if (rc) return rc;
            LstmSeqArgs r0 = l;
            r0.rec = 1; r0.sv_nl = 2; r0.sv_l0 = 0; r0.hout = nullptr;
            r0.z_out = nullptr; r0.sv_x = nullptr;
            rc = launch_lstm<0>(r0, NT, 1, false, st);
        }
        if (rc) return rc;
        const float* x = l.sv_hd ? l.sv_hd : sv->h;
        const long xs = l.sv_hd ? (long)4 * NT * 64 : (long)lay.NL * 4 * NT * 64;
        rc = launch_lstm_pre(NT, l.w[1], l.b[1], NT, x, xs, sv->gates + (size_t)16 * NT * 64, (long)lay.NL * 16 * NT * 64,
                             rt * T, st);
        if (rc) return rc;
        LstmSeqArgs r1 = l;
        r1.w[0] = l.w[1]; r1.b[0] = l.b[1]; r1.kx = 4 * NT;
        r1.rec = 1; r1.sv_nl = 2; r1.sv_l0 = 1;
        r1.z_out = nullptr; r1.sv_x = nullptr;
        rc = launch_lstm<0>(r1, NT, 1, false, st);
    } else {
        rc = launch_lstm<1>(l, lay.NTd, lay.NL, false, st);
    }
    if (rc) return rc;

    DenseChainArgs a = {};
    a.n_layers = dc.n_head;
    for (int i = 0; i < dc.n_head; ++i) fill_dense(a, i, packed, dc.head[i]);
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = touts;
    a.in_mode = 1; a.in = sv->hout; a.in_ks = 4 * lay.NTd;
    a.out_mode = 1; a.out = surf_out; a.out_f = dc.n_surf; a.S = 1;
    a.aux_layer = (lay.X > 0 && ex_out) ? dc.ex_layer : -1;
    a.aux = ex_out; a.aux_f0 = dc.ex_f0; a.aux_n = lay.X;
    a.save = sv->head_acts;
    return launch_dense_chain(a, st);
}

extern "C" int vsv_dense_backward(const vsv_dense_bwd* d, void* stream) {
    if (!d || !d->in) { set_error("vsv_dense_backward: null pointer"); return -1; }
    if (d->n_layers < 1 || d->n_layers > VSV_MAX_CHAIN) { set_error("vsv_dense_backward: 1..%d layers", VSV_MAX_CHAIN); return -1; }
    if (d->n_rows < 1 || d->T < 1 || d->n_rowtiles != (d->n_rows + 31) / 32) { set_error("vsv_dense_backward: bad sizes"); return -1; }
    // (the kernels clamp feature indices into [0, in_f) and [add_f0, add_f0 + add_n) instead of branching around loads)
    if (d->in_mode == 0 && d->in_f < 1) { set_error("vsv_dense_backward: in_f must be >= 1"); return -1; }
    for (int i = 0; i < d->n_layers; ++i)
        if (d->layer[i].add && d->layer[i].add_n < 1) { set_error("vsv_dense_backward: layer %d adds an empty range", i); return -1; }
    for (int i = 0; i < d->n_layers; ++i) {
        const vsv_dense_bwd_layer& ly = d->layer[i];
        if (ly.ntT < 0 || ly.ntT > kHeadTiles || (ly.ntT > 0 && (!ly.wT || ly.nqT < 1 || ly.nqT > 4 * kHeadTiles))) {
            set_error("vsv_dense_backward: layer %d has an unsupported shape", i); return -1;
        }
    }
    const long insts = d->n_rowtiles * d->T;
    const long blocks = (insts + 3) / 4;
    if (blocks > 0x7fffffffL) { set_error("vsv_dense_backward: too many tiles"); return -2; }
    static const bool wg_off = [] { const char* e = getenv("VSV_DENSE_BWD_WG"); return e && e[0] == '0'; }();
    if (insts <= 1024 && !wg_off) {
        // few instances: a workgroup per instance, its four waves splitting slots and output tiles
        const size_t lds1 = (size_t)kDenseSlots * 64 * sizeof(float);
        hipLaunchKernelGGL(dense_bwd_wg_kernel, dim3((unsigned)insts), dim3(256), lds1, (hipStream_t)stream, *d);
        return launch_check("dense_bwd_wg_kernel");
    }
    const size_t lds = (size_t)4 * kDenseSlots * 64 * sizeof(float);
    hipLaunchKernelGGL(dense_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, *d);
    return launch_check("dense_bwd_kernel");
}

// ---------------------------------------------------------------------------
// weight gradients: dW = G^T . X over the (row, t) instances the backward kernels left in HBM,
// written straight to the packed-blob gradient.  G [inst][a][32 rows] (a = 2*slot + half, the tile-B
// layout read as a plain array) and X [inst][b][32 rows] are both K-contiguous per output index, so
// lane (i = lane & 31, kh = lane >> 5) of a 32x32x2 MFMA loads 16 consecutive rows of its a (and of
// its b) per instance as four float4 and the 32 rows go through 16 MFMAs (k pair j = rows j, 16 + j).
// One workgroup per (32 a) x (64 b) tile of one problem, its four waves splitting the instances;
// up to VSV_MAX_WGRAD problems (layers) per launch.
// ---------------------------------------------------------------------------
struct WgradArgs {
    vsv_wgrad w;
    int tile0[VSV_MAX_WGRAD + 1];
};

__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs a) {
    typedef const __attribute__((address_space(4))) WgradArgs* KP;
    const KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    __shared__ float red[3][2][16][64];           // partial tiles of waves 1..3
    __shared__ float redb[3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;                   // the launcher sizes the grid to the tile count
    const int n = kp->w.n;
    int pi = 0;
    while (pi + 1 < n && tile >= kp->tile0[pi + 1]) ++pi;                     // block-uniform
    const float* g = kp->w.p[pi].g;
    const long g_stride = kp->w.p[pi].g_stride;
    const int B0 = kp->w.p[pi].n_b[0], B1 = kp->w.p[pi].n_b[1];
    const int NI = kp->w.p[pi].n_inst, Tw = kp->w.p[pi].T;
    const int nbg = (B0 + B1 + 63) / 64;
    const int lt = tile - kp->tile0[pi];
    const int at = lt / nbg, bg = lt - at * nbg;
    const int i32 = lane & 31, kh = lane >> 5;

    const float* ga = g + (size_t)(at * 32 + i32) * 32 + kh * 16;
    const float* xb[2];
    long xs[2];
    int sh[2];
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = bg * 64 + j * 32 + i32;
        ok[j] = b < B0 + B1;
        const int bc = ok[j] ? b : 0;                                       // out-of-range columns read column 0, never stored
        const int seg = (bc >= B0) ? 1 : 0;
        const int bl = seg ? bc - B0 : bc;
        xb[j] = (seg ? kp->w.p[pi].x[1] : kp->w.p[pi].x[0]) + (size_t)bl * 32 + kh * 16;
        xs[j] = seg ? kp->w.p[pi].x_stride[1] : kp->w.p[pi].x_stride[0];
        sh[j] = seg ? kp->w.p[pi].shift[1] : kp->w.p[pi].shift[0];
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float bsum = 0.0f;
    // The four waves of the block split the instances (i = wave, wave + 4, ...).  Software pipeline: the
    // next instance's operands are in flight while this one's go through the MFMAs; loads are
    // unconditional (a shifted read at t = 0 reads the instance itself) and the zero for t = 0 is applied
    // as a factor afterwards, so nothing waits on the loads before the MFMAs issue.
    float4 av[4], bv[2][4], an[4], bn[2][4];
    float mk[2], mn[2];
    auto fetch = [&](int i, float4 (&fa)[4], float4 (&fb)[2][4], float (&m)[2]) {
        const int t = i % Tw;
        const float4* gp4 = (const float4*)(ga + (size_t)i * g_stride);
#pragma unroll
        for (int q = 0; q < 4; ++q) fa[q] = gp4[q];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool dead = sh[j] && t == 0;
            m[j] = dead ? 0.0f : 1.0f;
            const float4* xp = (const float4*)(xb[j] + (size_t)(i - (dead ? 0 : sh[j])) * xs[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) fb[j][q] = xp[q];
        }
    };
    if (wave < NI) fetch(wave, av, bv, mk);
    for (int i = wave; i < NI; i += 4) {
        if (i + 4 < NI) fetch(i + 4, an, bn, mn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float aq[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bq[4] = {bv[j][q].x * mk[j], bv[j][q].y * mk[j], bv[j][q].z * mk[j], bv[j][q].w * mk[j]};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j] = mfma32(aq[e], bq[e], acc[j]);
            }
            bsum += (aq[0] + aq[1]) + (aq[2] + aq[3]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = an[q];
            bv[0][q] = bn[0][q];
            bv[1][q] = bn[1][q];
        }
        mk[0] = mn[0];
        mk[1] = mn[1];
    }
    // fixed-order reduction over the four waves (deterministic)
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][j][r][lane] = acc[j][r];
        redb[wave - 1][lane] = bsum;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] += red[w][j][r][lane];
        bsum += redb[w][lane];
    }
    float* gpo = kp->w.gp;
    const long w_off = kp->w.p[pi].w_off;
    const int nt = kp->w.p[pi].nt;
    // D register r of lane: a = 32*at + 8*(r>>2) + 4*kh + (r&3), b = 64*bg + 32*j + i32
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!ok[j]) continue;
        const int b = bg * 64 + j * 32 + i32;
        const long colpart = w_off + (long)(b >> 3) * nt * 256 + (b & 1) * 128 + ((b >> 1) & 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int av_ = at * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
            const int s_ = av_ >> 1, hh = av_ & 1;
            const int nf = 32 * (s_ >> 4) + 8 * ((s_ >> 2) & 3) + 4 * hh + (s_ & 3);      // slot order -> feature
            gpo[colpart + (long)(nf >> 5) * 256 + (nf & 31) * 4] = acc[j][r];
        }
    }
    if (bg == 0 && kp->w.p[pi].bias) {
        bsum += __shfl_xor(bsum, 32);
        if (kh == 0) {
            const int av_ = at * 32 + i32;
            const int s_ = av_ >> 1, hh = av_ & 1;
            const int nf = 32 * (s_ >> 4) + 8 * ((s_ >> 2) & 3) + 4 * hh + (s_ & 3);
            const int tq = nf >> 5, iq = nf & 31;
            gpo[kp->w.p[pi].b_off + (tq * 2 + ((iq >> 2) & 1)) * 16 + 4 * (iq >> 3) + (iq & 3)] = bsum;
        }
    }
}

extern "C" int vsv_weight_grads(const vsv_wgrad* d, void* stream) {
    if (!d || !d->gp) { set_error("vsv_weight_grads: null pointer"); return -1; }
    if (d->n < 1 || d->n > VSV_MAX_WGRAD) { set_error("vsv_weight_grads: 1..%d problems per launch", VSV_MAX_WGRAD); return -1; }
    WgradArgs a;
    a.w = *d;
    long tiles = 0;
    for (int i = 0; i < d->n; ++i) {
        const vsv_wgrad_prob& p = d->p[i];
        if (!p.g || !p.x[0] || (p.n_b[1] > 0 && !p.x[1])) { set_error("vsv_weight_grads: problem %d: null pointer", i); return -1; }
        if (p.n_a < 32 || p.n_a % 32 || p.n_b[0] < 1 || p.n_b[1] < 0 || p.n_inst < 1 || p.T < 1 || p.n_inst % p.T ||
            p.nt < 1 || p.n_a > 32 * p.nt || p.g_stride < (long)p.n_a * 32 || p.x_stride[0] < (long)p.n_b[0] * 32 ||
            (p.n_b[1] > 0 && p.x_stride[1] < (long)p.n_b[1] * 32) || (p.shift[0] | p.shift[1]) & ~1) {
            set_error("vsv_weight_grads: problem %d has an unsupported shape", i); return -1;
        }
        a.tile0[i] = (int)tiles;
        tiles += (long)(p.n_a / 32) * ((p.n_b[0] + p.n_b[1] + 63) / 64);
        if (tiles > 0x3fffffffL) { set_error("vsv_weight_grads: too many tiles"); return -2; }
    }
    for (int i = d->n; i <= VSV_MAX_WGRAD; ++i) a.tile0[i] = (int)tiles;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return launch_check("wgrad_kernel");
}

template <int NT, int NL>
static int launch_lstm_bwd(const vsv_lstm_bwd& a, hipStream_t st) {
    const size_t lds = (size_t)NL * 16 * NT * 64 * sizeof(float);
    int rc = set_lds(lstm_bwd_kernel<NT, NL>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((lstm_bwd_kernel<NT, NL>), dim3((unsigned)a.n_rowtiles), dim3(256), lds, st, a);
    return launch_check("lstm_bwd_kernel");
}

template <int NT>
static int launch_lstm_bwd16(const vsv_lstm_bwd& a, hipStream_t st) {
    constexpr int KS = 16 * ((8 * NT + 15) / 16);
    const size_t lds = (size_t)2 * KS * 64 * sizeof(float);
    int rc = set_lds(lstm_bwd16_kernel<NT>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((lstm_bwd16_kernel<NT>), dim3((unsigned)(2 * a.n_rowtiles)), dim3(512), lds, st, a);
    return launch_check("lstm_bwd16_kernel");
}

// one nn.LSTM layer per launch, few row tiles: the 16-row schedule (VSV_BWD16=0: lstm_bwd_kernel's 32-row tiles)
static bool lstm_bwd16_applies(const vsv_lstm_bwd& a, int layers) {
    static const bool off = [] { const char* e = getenv("VSV_BWD16"); return e && e[0] == '0'; }();
    return !off && layers == 1 && a.gru == 0 && a.nt_x == 0 && 2 * a.n_rowtiles <= 256;
}

extern "C" int vsv_lstm_backward(const vsv_lstm_bwd* d, int hidden, int layers, void* stream) {
    if (!d || !d->sv_gates || !d->sv_c || !d->dgates || !d->wT_hh[0]) { set_error("vsv_lstm_backward: null pointer"); return -1; }
    if (d->n_rows < 1 || d->T < 1 || d->t_out0 < 0 || d->t_out0 > d->T || d->n_rowtiles != (d->n_rows + 31) / 32 ||
        d->n_rowtiles > 0x7fffffffL) { set_error("vsv_lstm_backward: bad sizes"); return -1; }
    if (d->nt_x < 0 || d->nt_x > 4 || (d->nt_x > 0 && !d->wT_in[0])) { set_error("vsv_lstm_backward: bad nt_x"); return -1; }
    if (layers == 2 && (!d->wT_hh[1] || !d->wT_in[1])) { set_error("vsv_lstm_backward: layer 1 weights missing"); return -1; }
    if (d->drop_p < 0.0f || d->drop_p >= 1.0f) { set_error("vsv_lstm_backward: dropout probability must be in [0,1)"); return -1; }
    if (d->sv_nl < 0 || (d->sv_nl > 0 && (d->sv_l0 < 0 || d->sv_l0 + layers > d->sv_nl))) { set_error("vsv_lstm_backward: bad sv_nl / sv_l0"); return -1; }
    const int NT = (hidden + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    if (lstm_bwd16_applies(*d, layers)) {
        // 4-row tiles while eight of them per 32 rows still find a CU each (VSV_BWD4=0: 16-row tiles)
        static const bool no4 = [] { const char* e = getenv("VSV_BWD4"); return e && e[0] == '0'; }();
        if (!no4 && 8 * d->n_rowtiles <= 256) {
            const size_t lds = (size_t)(4 * 4 * 144 + 8 * 4 * 64) * sizeof(float);
            if (NT == 13) hipLaunchKernelGGL((lstm_bwd4_kernel<13>), dim3((unsigned)(8 * d->n_rowtiles)), dim3(512), lds, st, *d);
            else if (NT == 16) hipLaunchKernelGGL((lstm_bwd4_kernel<16>), dim3((unsigned)(8 * d->n_rowtiles)), dim3(512), lds, st, *d);
            if (NT == 13 || NT == 16) return launch_check("lstm_bwd4_kernel");
        }
        if (NT == 13) return launch_lstm_bwd16<13>(*d, st);
        if (NT == 16) return launch_lstm_bwd16<16>(*d, st);
    }
    if (NT == 13) return layers == 2 ? launch_lstm_bwd<13, 2>(*d, st) : launch_lstm_bwd<13, 1>(*d, st);
    if (NT == 16) return layers == 2 ? launch_lstm_bwd<16, 2>(*d, st) : launch_lstm_bwd<16, 1>(*d, st);
    set_error("vsv_lstm_backward: unsupported hidden size %d", hidden);
    return -4;
}

extern "C" int vsv_tileb_gemm_launch(const vsv_tileb_gemm* d, void* stream) {
    if (!d || !d->wT || !d->g || (!d->out_tile && !d->out_rows)) { set_error("vsv_tileb_gemm_launch: null pointer"); return -1; }
    if (d->nqT < 1 || d->ntT < 1 || d->ntT > 4 || d->T < 1 || d->n_rows < 1 || d->n_rowtiles != (d->n_rows + 31) / 32 ||
        d->g_stride < (long)d->nqT * 4 * 64 || (d->out_rows && d->out_f < 1)) {
        set_error("vsv_tileb_gemm_launch: bad sizes"); return -1;
    }
    if (d->drop_p < 0.0f || d->drop_p >= 1.0f) { set_error("vsv_tileb_gemm_launch: dropout probability must be in [0,1)"); return -1; }
    const long insts = d->n_rowtiles * d->T;
    if (insts > 0x7fffffffL) { set_error("vsv_tileb_gemm_launch: too many instances"); return -2; }
    if (d->nqT > 64) { set_error("vsv_tileb_gemm_launch: more than 64 K quads"); return -4; }
    size_t lds = (size_t)d->nqT * 4 * 64 * sizeof(float);
    if (insts * d->ntT <= 1024) {
        // few instances: one workgroup per (instance, output tile), K split over its four waves
        if (lds < (size_t)3 * 16 * 64 * sizeof(float)) lds = (size_t)3 * 16 * 64 * sizeof(float);
        if (int rc = set_lds(tileb_gemm_kernel<true>, lds)) return rc;
        hipLaunchKernelGGL(tileb_gemm_kernel<true>, dim3((unsigned)(insts * d->ntT)), dim3(256), lds, (hipStream_t)stream, *d);
    } else {
        if (int rc = set_lds(tileb_gemm_kernel<false>, lds)) return rc;
        hipLaunchKernelGGL(tileb_gemm_kernel<false>, dim3((unsigned)insts), dim3(256), lds, (hipStream_t)stream, *d);
    }
    return launch_check("tileb_gemm_kernel");
}

// ---- the backward launches of TWO independent stacks as one launch each (the two encoders of a training step at
// few row tiles: see vsv_train_encode_pair).  Each *_pair entry point takes the two argument sets of the matching
// single entry point; when both would run the few-instance kernel they go out as one launch whose blockIdx.y picks
// the set, otherwise as two launches one after the other on `stream`.  Same results either way.
static bool pair_off() {
    static const bool off = [] { const char* e = getenv("VSV_ENC_PAIR"); return e && e[0] == '0'; }();
    return off;
}
extern "C" int vsv_dense_backward_pair(const vsv_dense_bwd* d0, const vsv_dense_bwd* d1, void* stream) {
    static const bool wg_off = [] { const char* e = getenv("VSV_DENSE_BWD_WG"); return e && e[0] == '0'; }();
    const bool ok = d0 && d1 && !pair_off() && !wg_off && d0->n_rowtiles * d0->T <= 1024 && d1->n_rowtiles * d1->T <= 1024 &&
                    d0->n_rowtiles * d0->T >= 1 && d1->n_rowtiles * d1->T >= 1;
    if (!ok) {
        if (int rc = vsv_dense_backward(d0, stream)) return rc;
        return vsv_dense_backward(d1, stream);
    }
    // (argument checks of the single entry point, per set)
    for (const vsv_dense_bwd* d : {d0, d1}) {
        if (!d->in || d->n_layers < 1 || d->n_layers > VSV_MAX_CHAIN || d->n_rows < 1 || d->T < 1 ||
            d->n_rowtiles != (d->n_rows + 31) / 32 || (d->in_mode == 0 && d->in_f < 1)) {
            set_error("vsv_dense_backward_pair: bad arguments"); return -1;
        }
        for (int i = 0; i < d->n_layers; ++i) {
            const vsv_dense_bwd_layer& ly = d->layer[i];
            if ((ly.add && ly.add_n < 1) || ly.ntT < 0 || ly.ntT > kHeadTiles ||
                (ly.ntT > 0 && (!ly.wT || ly.nqT < 1 || ly.nqT > 4 * kHeadTiles))) {
                set_error("vsv_dense_backward_pair: layer %d has an unsupported shape", i); return -1;
            }
        }
    }
    const long n0 = d0->n_rowtiles * d0->T, n1 = d1->n_rowtiles * d1->T;
    const size_t lds1 = (size_t)kDenseSlots * 64 * sizeof(float);
    hipLaunchKernelGGL(dense_bwd_wg_pair_kernel, dim3((unsigned)(n0 > n1 ? n0 : n1), 2), dim3(256), lds1, (hipStream_t)stream, *d0, *d1);
    return launch_check("dense_bwd_wg_pair_kernel");
}

extern "C" int vsv_lstm_backward_pair(const vsv_lstm_bwd* d0, const vsv_lstm_bwd* d1, int hidden, int layers, void* stream) {
    static const bool no4 = [] { const char* e = getenv("VSV_BWD4"); return e && e[0] == '0'; }();
    const int NT = (hidden + 7) / 8;
    bool ok = d0 && d1 && !pair_off() && !no4 && NT == 13 && layers == 1;
    for (const vsv_lstm_bwd* d : {d0, d1}) {
        if (!ok) break;
        ok = d->sv_gates && d->sv_c && d->dgates && d->wT_hh[0] && d->n_rows >= 1 && d->T >= 1 && d->t_out0 >= 0 && d->t_out0 <= d->T &&
             d->n_rowtiles == (d->n_rows + 31) / 32 && d->nt_x == 0 && d->drop_p >= 0.0f && d->drop_p < 1.0f &&
             !(d->sv_nl < 0 || (d->sv_nl > 0 && (d->sv_l0 < 0 || d->sv_l0 + layers > d->sv_nl))) &&
             lstm_bwd16_applies(*d, layers) && 8 * d->n_rowtiles <= 256;
    }
    if (!ok) {
        if (int rc = vsv_lstm_backward(d0, hidden, layers, stream)) return rc;
        return vsv_lstm_backward(d1, hidden, layers, stream);
    }
    const long n0 = 8 * d0->n_rowtiles, n1 = 8 * d1->n_rowtiles;
    const size_t lds = (size_t)(4 * 4 * 144 + 8 * 4 * 64) * sizeof(float);
    hipLaunchKernelGGL((lstm_bwd4_pair_kernel<13>), dim3((unsigned)(n0 > n1 ? n0 : n1), 2), dim3(512), lds, (hipStream_t)stream, *d0, *d1);
    return launch_check("lstm_bwd4_pair_kernel");
}

extern "C" int vsv_tileb_gemm_launch_pair(const vsv_tileb_gemm* d0, const vsv_tileb_gemm* d1, void* stream) {
    bool ok = d0 && d1 && !pair_off();
    for (const vsv_tileb_gemm* d : {d0, d1}) {
        if (!ok) break;
        ok = d->wT && d->g && (d->out_tile || d->out_rows) && d->nqT >= 1 && d->nqT <= 64 && d->ntT >= 1 && d->ntT <= 4 && d->T >= 1 &&
             d->n_rows >= 1 && d->n_rowtiles == (d->n_rows + 31) / 32 && d->g_stride >= (long)d->nqT * 4 * 64 &&
             !(d->out_rows && d->out_f < 1) && d->drop_p >= 0.0f && d->drop_p < 1.0f && d->n_rowtiles * d->T * d->ntT <= 1024;
    }
    if (!ok) {
        if (int rc = vsv_tileb_gemm_launch(d0, stream)) return rc;
        return vsv_tileb_gemm_launch(d1, stream);
    }
    const int nq = d0->nqT > d1->nqT ? d0->nqT : d1->nqT;
    size_t lds = (size_t)nq * 4 * 64 * sizeof(float);
    if (lds < (size_t)3 * 16 * 64 * sizeof(float)) lds = (size_t)3 * 16 * 64 * sizeof(float);
    if (int rc = set_lds(tileb_gemm_pair_kernel, lds)) return rc;
    const long n0 = d0->n_rowtiles * d0->T * d0->ntT, n1 = d1->n_rowtiles * d1->T * d1->ntT;
    hipLaunchKernelGGL(tileb_gemm_pair_kernel, dim3((unsigned)(n0 > n1 ? n0 : n1), 2), dim3(256), lds, (hipStream_t)stream, *d0, *d1);
    return launch_check("tileb_gemm_pair_kernel");
}

// Backward of the reparameterisation z = mu + exp(0.5 logvar) * eps (:199) and of the decoder-input cat (:1066):
// from dx (B, T, 2L) = d loss / d [ctx_pad | z]
//   d_enc (B, T, 2L) = [ dz + d_zm | dz * 0.5 * (z - mu) + d_zl ]   (d_zm / d_zl: the KL term's gradients; dz = dx[.., L:] + d_z)
//   d_ctx (B, Tc, L) = dx[:, :Tc, :L] [* ctx_mask (B, Tc)]
__global__ void __launch_bounds__(256) reparam_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ z,
                                                          const float* __restrict__ zm, const float* __restrict__ d_zm,
                                                          const float* __restrict__ d_zl, const float* __restrict__ d_z,
                                                          const float* __restrict__ ctx_mask, long B, int T, int Tc, int L,
                                                          int Lc, float* __restrict__ d_enc, float* __restrict__ d_ctx) {
    const int W = Lc + L;                                          // dx row: [ctx part (Lc) | z part (L)]
    const long n = B * T * L;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int f = (int)(i % L);
        const long bt = i / L;
        float dz = dx[bt * W + Lc + f];
        if (d_z) dz += d_z[i];
        d_enc[bt * 2 * L + f] = d_zm ? dz + d_zm[i] : dz;
        const float dl = dz * 0.5f * (z[i] - zm[i]);
        d_enc[bt * 2 * L + L + f] = d_zl ? dl + d_zl[i] : dl;
    }
    const long nc = B * Tc * Lc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nc; i += (long)gridDim.x * 256) {
        const int f = (int)(i % Lc);
        const long bt = i / Lc;
        const int t = (int)(bt % Tc);
        const long b = bt / Tc;
        const float g = dx[(b * T + t) * W + f];
        d_ctx[i] = ctx_mask ? g * ctx_mask[bt] : g;
    }
}

extern "C" int vsv_reparam_backward(const float* dx, const float* z, const float* zm, const float* d_zm, const float* d_zl,
                                    const float* d_z, const float* ctx_mask, long B, int T, int Tc, int L, int Lc,
                                    float* d_enc, float* d_ctx, void* stream) {
    if (!dx || !z || !zm || !d_enc || !d_ctx) { set_error("vsv_reparam_backward: null pointer"); return -1; }
    if (B < 1 || T < 1 || Tc < 0 || Tc > T || L < 1 || Lc < 1) { set_error("vsv_reparam_backward: bad sizes"); return -1; }
    const long n = B * T * (L > Lc ? L : Lc);
    const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dx, z, zm, d_zm, d_zl, d_z,
                       ctx_mask, B, T, Tc, L, Lc, d_enc, d_ctx);
    return launch_check("reparam_bwd_kernel");
}

// ---- live parameters <-> packed blob, on the device ------------------------------------------------------------
// A training step re-packs the blob from the nn.Parameters after every optimizer step and turns the blob's gradient
// back into parameter gradients.  Both are pure index maps (vsv/engine.py::build_gather_maps); as torch index ops
// they cost ~10 launches each way.  Here: one launch each, the parameters addressed through a device table of their
// data pointers.  A map entry is (tensor << 24 | element), 0 = nothing.
__global__ void __launch_bounds__(256) pack_params_kernel(const float* const* __restrict__ ptrs, const unsigned* __restrict__ map_a,
                                                          const unsigned* __restrict__ map_b, const float* __restrict__ cst,
                                                          float* __restrict__ packed, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned a = map_a[i], b = map_b[i];
        float v = cst ? cst[i] : 0.0f;
        if (a) v += ptrs[(a >> 24) - 1][a & 0xffffffu];
        if (b) v += ptrs[(b >> 24) - 1][b & 0xffffffu];
        packed[i] = v;
    }
}

extern "C" int vsv_pack_params(const void* const* param_ptrs, const unsigned* map_a, const unsigned* map_b, const float* cst,
                               float* packed, long n, void* stream) {
    if (!param_ptrs || !map_a || !map_b || !packed || n < 1) { set_error("vsv_pack_params: null pointer or empty blob"); return -1; }
    const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(pack_params_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float* const*)param_ptrs,
                       map_a, map_b, cst, packed, n);
    return launch_check("pack_params_kernel");
}

// Parameter gradients from the blob's gradient: flat[j] = sum of gp over the blob positions parameter element j was
// packed to (CSR rows in ascending blob position: a fixed summation order, unlike an atomic scatter), and per
// workgroup the sum of squares of what it wrote, for the gradient-norm clip.
__global__ void __launch_bounds__(256) param_grads_kernel(const float* __restrict__ gp, const int* __restrict__ row_ptr,
                                                          const int* __restrict__ cols, long n, float* __restrict__ flat,
                                                          float* __restrict__ partials) {
    __shared__ float red[4];
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    float g = 0.0f;
    if (j < n) {
        // (a conv tap sits at up to 25 blob positions: eight independent loads in flight, fixed order)
        const int e = row_ptr[j + 1];
        int k = row_ptr[j];
        for (; k + 8 <= e; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gp[cols[k + u]];
            g += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; k < e; ++k) g += gp[cols[k]];
        flat[j] = g;
    }
    float q = g * g;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int vsv_param_grads(const float* gp, const int* row_ptr, const int* cols, long n, float* flat, float* partials,
                               void* stream) {
    if (!gp || !row_ptr || !cols || !flat || !partials || n < 1) { set_error("vsv_param_grads: null pointer or no parameters"); return -1; }
    if ((n + 255) / 256 > 0x7fffffffL) { set_error("vsv_param_grads: too many parameters"); return -2; }
    hipLaunchKernelGGL(param_grads_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gp, row_ptr, cols,
                       n, flat, partials);
    return launch_check("param_grads_kernel");
}

// clip_grad_norm_ (reference :1129) on the flat gradient: every workgroup adds the n_partials partial sums of squares
// in the same order (so all of them get the same coefficient, run to run), then scales its share in place by
// min(1, max_norm / (norm + 1e-6)).  norm_out (may be null) receives the total norm.
__global__ void __launch_bounds__(256) clip_scale_kernel(float* __restrict__ flat, long n, const float* __restrict__ partials,
                                                         int n_partials, float max_norm, float* __restrict__ norm_out) {
    __shared__ float red[4];
    float q = 0.0f;
    for (int i = threadIdx.x; i < n_partials; i += 256) q += partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
    const float coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
    if (coef < 1.0f)
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) flat[i] *= coef;
}

extern "C" int vsv_clip_grad_norm(float* flat, long n, const float* partials, int n_partials, float max_norm, float* norm_out,
                                  void* stream) {
    if (!flat || !partials || n < 1 || n_partials < 1) { set_error("vsv_clip_grad_norm: null pointer or empty"); return -1; }
    if (!(max_norm > 0.0f)) { set_error("vsv_clip_grad_norm: max_norm must be positive"); return -1; }
    const unsigned blocks = (unsigned)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, flat, n, partials, n_partials, max_norm,
                       norm_out);
    return launch_check("clip_scale_kernel");
}

// torch.optim.AdamW's step (the reference's optimizer.step(), :1131, for the optimizer its trainers build) over all
// parameters in one launch: the tensors (parameter, exp_avg, exp_avg_sq: the optimizer's own state, updated in
// place) are addressed through device pointer tables, the gradient is the flat buffer vsv_param_grads wrote.  The
// arithmetic follows torch's foreach implementation step by step:
//   p *= 1 - lr * wd;  m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;
//   p += -step_size * (m / (sqrt(v) / bc2_sqrt + eps)),   step_size = lr / (1 - b1^t),  bc2_sqrt = sqrt(1 - b2^t)
struct AdamwArgs {
    float* const* p; float* const* m; float* const* v;
    const float* grad; const int* starts; int n_tensors; long n;
    float decay, w1, b2, w2, eps, neg_step_size, bc2_sqrt;
};

__global__ void __launch_bounds__(256) adamw_kernel(const AdamwArgs a) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    int lo = 0, hi = a.n_tensors;                                  // starts[lo] <= i < starts[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((long)a.starts[mid] <= i) lo = mid; else hi = mid;
    }
    const long o = i - a.starts[lo];
    const float g = a.grad[i];
    float p = a.p[lo][o], m = a.m[lo][o], v = a.v[lo][o];
    p *= a.decay;
    m = m + (g - m) * a.w1;
    v = v * a.b2;
    v = v + (a.w2 * g) * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + a.neg_step_size * (m / denom);
    a.p[lo][o] = p; a.m[lo][o] = m; a.v[lo][o] = v;
}

extern "C" int vsv_adamw_step(void* const* params, void* const* exp_avg, void* const* exp_avg_sq, const float* grad_flat,
                              const int* starts, int n_tensors, long n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, long step, void* stream) {
    if (!params || !exp_avg || !exp_avg_sq || !grad_flat || !starts) { set_error("vsv_adamw_step: null pointer"); return -1; }
    if (n_tensors < 1 || n < 1 || step < 1) { set_error("vsv_adamw_step: need tensors, elements and step >= 1"); return -1; }
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0)) {
        set_error("vsv_adamw_step: bad hyper-parameters"); return -1;
    }
    AdamwArgs a;
    a.p = (float* const*)params; a.m = (float* const*)exp_avg; a.v = (float* const*)exp_avg_sq;
    a.grad = grad_flat; a.starts = starts; a.n_tensors = n_tensors; a.n = n;
    // (the scalars as torch computes them: Python doubles rounded to float where the foreach kernels take them)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    a.decay = (float)(1.0 - lr * weight_decay);
    a.w1 = (float)(1.0 - beta1);
    a.b2 = (float)beta2;
    a.w2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.neg_step_size = (float)(-(lr / bc1));
    a.bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return launch_check("adamw_kernel");
}

extern "C" int vsv_loss_grad(const vsv_loss* d, void* stream) {
    if (!d || !d->pred || !d->target || !d->zm || !d->zl || !d->d_pred || !d->d_zm || !d->d_zl || !d->sums) {
        set_error("vsv_loss_grad: null pointer"); return -1;
    }
    if (d->B < 1 || d->H < 1 || d->T < d->H || d->Q < 1 || d->Q > 8 || d->L < 1) { set_error("vsv_loss_grad: bad sizes"); return -1; }
    if (d->Hs != 0 && d->Hs < d->H) { set_error("vsv_loss_grad: Hs < H"); return -1; }
    if (d->surf_l2 && d->Q != 1) { set_error("vsv_loss_grad: the MSE surface loss is the one-channel decoder's (Q = 1)"); return -1; }
    if (d->ex_pred && (!d->ex_target || !d->d_ex || d->X < 1 || d->ex_n < 1 || d->ex_n > d->X)) { set_error("vsv_loss_grad: bad ex arguments"); return -1; }
    if ((d->loss_vec != nullptr) != (d->mix != nullptr) || (d->loss_vec != nullptr) != (d->ticket != nullptr)) {
        set_error("vsv_loss_grad: mix, loss_vec and ticket go together"); return -1;
    }
    hipLaunchKernelGGL(loss_grad_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, *d);
    return launch_check("loss_grad_kernel");
}
