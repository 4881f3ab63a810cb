// gfx950 kernels of libvsv_hip.so and their launchers.
//
// Execution model: one wave owns a tile of 16 independent rows
// (batch index m = lane & 15).  Every layer is y^T[N x 16] = W[N x K] . x^T[K x 16]
// on v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), with the WEIGHTS as the A
// operand (pre-packed fragments, streamed from L2) and the ACTIVATIONS as the B
// operand.  The D layout (lane (g,m) holds rows 4g..4g+3 of a tile for row m)
// coincides with the B layout of the next layer (lane (g,m) supplies k = 4s+g),
// so activations never move between lanes: each lane keeps its own slice of
// every activation vector in a private column of LDS ("priv[k*64 + lane]").
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>

#include "vsv_layout.h"

namespace vsv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc[t] += W[tile t] . B for nq k-quads.  w: this lane's float4 of quad 0, tile 0
// (packed [nq][NT][64][4]); bsrc: this lane's column of the activation array at the
// first k-step (LDS, stride 64 floats per k-step).
// `nts` is the tile stride of the packed layer (its total tile count), so a caller can
// compute a sub-range of NT tiles by offsetting w.
template <int NT>
__device__ __forceinline__ void tile_gemm(f32x4 (&acc)[NT], const float4* __restrict__ w,
                                          const float* bsrc, int nq, int nts = NT) {
    constexpr int G = (NT < 4) ? NT : 4;
#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const float b0 = bsrc[(4 * q + 0) * 64];
        const float b1 = bsrc[(4 * q + 1) * 64];
        const float b2 = bsrc[(4 * q + 2) * 64];
        const float b3 = bsrc[(4 * q + 3) * 64];
        const float4* wq = w + (size_t)q * nts * 64;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += G) {
            float4 a[G];
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (t0 + i < NT) a[i] = wq[(t0 + i) * 64];
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (t0 + i < NT) acc[t0 + i] = mfma4(a[i].x, b0, acc[t0 + i]);
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (t0 + i < NT) acc[t0 + i] = mfma4(a[i].y, b1, acc[t0 + i]);
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (t0 + i < NT) acc[t0 + i] = mfma4(a[i].z, b2, acc[t0 + i]);
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (t0 + i < NT) acc[t0 + i] = mfma4(a[i].w, b3, acc[t0 + i]);
        }
    }
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x4 (&acc)[NT], const float4* __restrict__ b) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 v = b[t * 64];
        acc[t] = f32x4{v.x, v.y, v.z, v.w};
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller: one N(0,1) value per (sample, day, step, f).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float philox_normal_elem(uint32_t k0, uint32_t k1, uint32_t sample,
                                                    uint32_t day, uint32_t step, int f) {
    uint32_t c0 = sample, c1 = day, c2 = step, c3 = (uint32_t)(f >> 2);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const int p = (f & 3) >> 1;
    const uint32_t xa = p ? c2 : c0, xb = p ? c3 : c1;
    const float u1 = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-07f;   // 2^-23
    const float u2 = ((float)(xb >> 9) + 0.5f) * 1.1920928955078125e-07f;
    const float r = sqrtf(-2.0f * logf(u1));
    const float th = 6.283185307179586f * u2;
    return (f & 1) ? r * sinf(th) : r * cosf(th);
}

// ---------------------------------------------------------------------------
// dense_chain: a stack of dense layers (conv embeddings, decoder head) applied
// to independent (row, t) instances.  One wave per (row tile, t).
// ---------------------------------------------------------------------------
struct DenseLayerArg {
    const float* w;
    const float* b;
    int nq, nt, n_relu;
};

struct DenseChainArgs {
    int n_layers;
    DenseLayerArg layer[kMaxDenseLayers];
    long n_rows;
    long n_rowtiles;
    int T;
    // input
    int in_mode;            // 0 row-major gather, 1 tile-B
    const float* in;
    long in_rs, in_ts;      // row-major strides (floats)
    int in_f;               // row-major: real features
    int in_ks;              // k-steps to fill (multiple of 4 for mode 0; = 4*nt_prev for mode 1)
    // injection into the LAST layer's output features [inj_f0, inj_f0+inj_n)
    const float* inj;
    long inj_rs, inj_ts;
    int inj_f0, inj_n;
    // output of the last layer
    int out_mode;           // 0 tile-B, 1 row-major (row, t, f), 2 sample-minor (row/S, t, f, row%S)
    float* out;
    int out_f;
    long S;
    // aux output: features [aux_f0, aux_f0+aux_n) of layer aux_layer -> aux (row, t, aux_n)
    int aux_layer;
    float* aux;
    int aux_f0, aux_n;
};

template <int NT>
__device__ __forceinline__ void dense_layer(const DenseChainArgs& a, int li, const float* ibuf, float* obuf,
                                            int lane, long row, bool row_ok, int t, long tile) {
    const DenseLayerArg& ly = a.layer[li];
    f32x4 acc[NT];
    load_bias<NT>(acc, (const float4*)ly.b + lane);
    tile_gemm<NT>(acc, (const float4*)ly.w + lane, ibuf, ly.nq);
    const int g = lane >> 4;
    const bool last = (li == a.n_layers - 1);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * tt + 4 * g + r;
            float v = acc[tt][r];
            if (f < ly.n_relu) v = fmaxf(v, 0.0f);
            if (last && a.inj && f >= a.inj_f0 && f < a.inj_f0 + a.inj_n)
                v = a.inj[row * a.inj_rs + (long)t * a.inj_ts + (f - a.inj_f0)];
            if (li == a.aux_layer && a.aux && row_ok && f >= a.aux_f0 && f < a.aux_f0 + a.aux_n)
                a.aux[(row * a.T + t) * a.aux_n + (f - a.aux_f0)] = v;
            if (!last) {
                obuf[(4 * tt + r) * 64] = v;
            } else if (a.out_mode == 0) {
                a.out[((size_t)tile * (4 * NT) + 4 * tt + r) * 64 + lane] = v;
            } else if (row_ok && f < a.out_f) {
                if (a.out_mode == 1)
                    a.out[((size_t)row * a.T + t) * a.out_f + f] = v;
                else
                    a.out[((((size_t)(row / a.S)) * a.T + t) * a.out_f + f) * a.S + (row % a.S)] = v;
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2) dense_chain_kernel(DenseChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * 4 + wave;
    if (tile >= a.n_rowtiles * a.T) return;
    const long rowtile = tile / a.T;
    const int t = (int)(tile % a.T);
    const int g = lane >> 4, m = lane & 15;
    long row = rowtile * 16 + m;
    const bool row_ok = row < a.n_rows;
    if (!row_ok) row = a.n_rows - 1;
    float* buf0 = smem + (size_t)wave * (2 * 32 * 64) + lane;
    float* buf1 = buf0 + 32 * 64;

    if (a.in_mode == 0) {
        const float* src = a.in + row * a.in_rs + (long)t * a.in_ts;
        for (int s = 0; s < a.in_ks; ++s) {
            const int f = 4 * s + g;
            buf0[s * 64] = (f < a.in_f) ? src[f] : 0.0f;
        }
    } else {
        const float* src = a.in + (size_t)tile * a.in_ks * 64 + lane;
        for (int s = 0; s < a.in_ks; ++s) buf0[s * 64] = src[s * 64];
        for (int s = a.in_ks; s < 4 * a.layer[0].nq; ++s) buf0[s * 64] = 0.0f;   // zero the last quad's pad
    }
    float* ib = buf0;
    float* ob = buf1;
    for (int li = 0; li < a.n_layers; ++li) {
        switch (a.layer[li].nt) {
            case 1: dense_layer<1>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            case 2: dense_layer<2>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            case 3: dense_layer<3>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            case 4: dense_layer<4>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            case 5: dense_layer<5>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            case 6: dense_layer<6>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            case 7: dense_layer<7>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
            default: dense_layer<8>(a, li, ib, ob, lane, row, row_ok, t, tile); break;
        }
        float* tmp = ib; ib = ob; ob = tmp;
    }
}

// ---------------------------------------------------------------------------
// lstm_seq: a 1- or 2-layer LSTM walked over T steps for a tile of 16 rows per
// wave, with (XMODE 0) the x part read from a tile-B tensor written by
// dense_chain (encoders) or (XMODE 1) assembled from ctx / z / noise (decoder).
// ---------------------------------------------------------------------------
struct LstmSeqArgs {
    const float* w[2];
    const float* b[2];
    int nq[2];
    int kx;                 // k-steps of the x part
    long n_rows, n_rowtiles;
    int T;
    // XMODE 0
    const float* x_tileb;   // [(rowtile*T + t)*kx + s][64]
    // XMODE 1
    int L;
    vsv_decode_in din;
    float* z_out;           // (n_rows, T, L) or null
    // state
    float* state_out;       // (n_rows, NL, 2, 4*NT) or null
    // per-step top-layer h in tile-B layout for t >= t_out0: [(rowtile*(T-t_out0) + t-t_out0)*NT + s][64]
    float* hout;
    int t_out0;
    // fused linear head on the top-layer h (encoders): out (n_rows, T, nh)
    const float* head_w;
    const float* head_b;
    int head_nq, head_nt, nh;
    float* head_out;
};

__device__ __forceinline__ float decoder_input(const LstmSeqArgs& a, long row, int t, int f) {
    const vsv_decode_in& d = a.din;
    const int L = a.L;
    if (f < L) {
        return (t < d.ctx_T) ? d.ctx[(row / d.ctx_div) * d.ctx_rs + (long)t * L + f] : 0.0f;
    }
    f -= L;
    if (f >= L) return 0.0f;
    float base, scale;
    if (t < d.z_T) {
        const long idx = (row / d.z_div) * d.z_rs + (long)t * L + f;
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
        if (d.eps_in) {
            eps = d.eps_in[((size_t)row * d.eps_T + te) * L + f];
        } else {
            eps = philox_normal_elem((uint32_t)d.seed, (uint32_t)(d.seed >> 32), (uint32_t)(row % d.S),
                                     (uint32_t)(d.day0 + row / d.S), (uint32_t)(d.step0 + te), f);
        }
        v = base + scale * eps;
    }
    if (a.z_out && row < a.n_rows) a.z_out[((size_t)row * a.T + t) * L + f] = v;
    return v;
}

// gates of tiles [t0, t0+NC) of layer l: bias already in acc
template <int NC, int XMODE>
__device__ __forceinline__ void lstm_gemm(f32x4 (&acc)[NC], const LstmSeqArgs& a, int l, int t0, int nts,
                                          long rowtile, int t, int lane, const float* priv, const float* hbase) {
    const float4* w = (const float4*)a.w[l] + (size_t)t0 * 64 + lane;
    if (XMODE == 0 && l == 0) {
        // x part straight from the tile-B tensor in global memory (kx is a multiple of 4)
        const int kx = a.kx;
        const float* xe = a.x_tileb + ((size_t)(rowtile * a.T + t) * kx) * 64 + lane;
        tile_gemm<NC>(acc, w, xe, kx / 4, nts);
        tile_gemm<NC>(acc, w + (size_t)(kx / 4) * nts * 64, hbase, a.nq[0] - kx / 4, nts);
    } else {
        tile_gemm<NC>(acc, w, (l == 0) ? priv : hbase, a.nq[l], nts);   // layer 1 reads [h0 | h1]
    }
}

// LSTM cell for NC tiles (one unit per tile per lane): gates i,f,g,o in acc[s][0..3]
template <int NC>
__device__ __forceinline__ void lstm_cell(const f32x4 (&acc)[NC], float* c, float (&hn)[NC]) {
#pragma unroll
    for (int s = 0; s < NC; ++s) {
        const float ig = sigmoidf_(acc[s][0]);
        const float fg = sigmoidf_(acc[s][1]);
        const float gg = tanhf(acc[s][2]);
        const float og = sigmoidf_(acc[s][3]);
        const float cn = fg * c[s] + ig * gg;
        c[s] = cn;
        hn[s] = og * tanhf(cn);
    }
}

// fused linear head on the top-layer h (encoders): NTH output tiles, natural order
template <int NTH>
__device__ __forceinline__ void lstm_head(const LstmSeqArgs& a, const float* htop, int lane, long row,
                                          bool row_ok, int t) {
    f32x4 acc2[NTH];
    load_bias<NTH>(acc2, (const float4*)a.head_b + lane);
    tile_gemm<NTH>(acc2, (const float4*)a.head_w + lane, htop, a.head_nq);
    if (!row_ok) return;
    const int g = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < NTH; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * tt + 4 * g + r;
            if (f < a.nh) a.head_out[((size_t)row * a.T + t) * a.nh + f] = acc2[tt][r];
        }
}

template <int NT, int NL, int XMODE>
__global__ void __launch_bounds__(256, 2) lstm_seq_kernel(LstmSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long rowtile = (long)blockIdx.x * 4 + wave;
    if (rowtile >= a.n_rowtiles) return;
    const int g = lane >> 4, m = lane & 15;
    const long row_raw = rowtile * 16 + m;
    const bool row_ok = row_raw < a.n_rows;
    const long row = row_ok ? row_raw : a.n_rows - 1;
    const int kx = a.kx;
    const int xk = (XMODE == 0) ? 0 : kx;                  // x k-steps staged in LDS (decoder only)
    const int ks_total = xk + NL * NT + 4;                 // + zeroed pad for the last quad
    float* priv = smem + (size_t)wave * ks_total * 64 + lane;
    float* hbase = priv + (size_t)xk * 64;                 // h of layer l at hbase[(l*NT + s)*64]
    constexpr int HP = 4 * NT;

    float c[NL][NT];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            float h0 = 0.0f, c0 = 0.0f;
            if (XMODE == 1 && a.din.state_in) {
                const float* st = a.din.state_in + ((size_t)(row / a.din.state_div) * NL + l) * 2 * HP;
                h0 = st[4 * s + g];
                c0 = st[HP + 4 * s + g];
            }
            hbase[(l * NT + s) * 64] = h0;
            c[l][s] = c0;
        }
    }
    for (int s = 0; s < 4; ++s) hbase[(NL * NT + s) * 64] = 0.0f;

    const int t_outs = a.T - a.t_out0;
    for (int t = 0; t < a.T; ++t) {
        // ---- x part of layer 0 -> priv[0 .. kx)
        if (XMODE == 1) {
            for (int s = 0; s < kx; ++s) priv[s * 64] = decoder_input(a, row, t, 4 * s + g);
        }
        // ---- layers.  The gate tiles are computed in chunks of <= TC tiles so that the
        // accumulators of one chunk (4 registers per tile) and both layers' cell states
        // stay inside the 256-register budget of 2 waves per SIMD.  h of a layer is
        // overwritten only after ALL of its chunks have accumulated (chunk 0's new h
        // waits in registers meanwhile).
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            constexpr int TC = 16;
            constexpr int NC0 = (NT < TC) ? NT : TC;          // tiles in chunk 0
            constexpr int NC1 = NT - NC0;                     // tiles in chunk 1 (0 if none)
            float* hl = hbase + (size_t)l * NT * 64;
            float hn0[NC0];
            {
                f32x4 acc[NC0];
                load_bias<NC0>(acc, (const float4*)a.b[l] + lane);
                lstm_gemm<NC0, XMODE>(acc, a, l, 0, NT, rowtile, t, lane, priv, hbase);
                lstm_cell<NC0>(acc, &c[l][0], hn0);
            }
            if (NC1 > 0) {
                constexpr int NC1s = NC1 > 0 ? NC1 : 1;
                f32x4 acc[NC1s];
                float hn1[NC1s];
                load_bias<NC1s>(acc, (const float4*)a.b[l] + NC0 * 64 + lane);
                lstm_gemm<NC1s, XMODE>(acc, a, l, NC0, NT, rowtile, t, lane, priv, hbase);
                lstm_cell<NC1s>(acc, &c[l][NC0], hn1);
#pragma unroll
                for (int s = 0; s < NC1s; ++s) hl[(NC0 + s) * 64] = hn1[s];
            }
#pragma unroll
            for (int s = 0; s < NC0; ++s) hl[s * 64] = hn0[s];
        }
        const float* htop = hbase + (size_t)(NL - 1) * NT * 64;
        // ---- outputs
        if (a.hout && t >= a.t_out0) {
            float* ho = a.hout + ((size_t)(rowtile * t_outs + (t - a.t_out0)) * NT) * 64 + lane;
            for (int s = 0; s < NT; ++s) ho[s * 64] = htop[s * 64];
        }
        if (a.head_w) {
            if (a.head_nt == 2)
                lstm_head<2>(a, htop, lane, row, row_ok, t);
            else
                lstm_head<1>(a, htop, lane, row, row_ok, t);
        }
    }
    if (a.state_out && row_ok) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            float* st = a.state_out + ((size_t)row * NL + l) * 2 * HP;
#pragma unroll
            for (int s = 0; s < NT; ++s) {
                st[4 * s + g] = hbase[(l * NT + s) * 64];
                st[HP + 4 * s + g] = c[l][s];
            }
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
                                                         PercArgs pa, int inner, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int tid = threadIdx.x;
    for (long v = blockIdx.x; v < n_vec; v += gridDim.x) {
        const float* src = in + (size_t)v * S;
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
    const long tiles = a.n_rowtiles * a.T;
    if (tiles <= 0) return 0;
    const size_t lds = 4 * 2 * 32 * 64 * sizeof(float);
    const long blocks = (tiles + 3) / 4;
    if (blocks > 0x7fffffffL) { set_error("dense_chain: too many tiles"); return -2; }
    hipLaunchKernelGGL(dense_chain_kernel, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return launch_check("dense_chain_kernel");
}

template <int NT, int NL, int XMODE>
static int launch_lstm_t(const LstmSeqArgs& a, hipStream_t st) {
    const size_t lds = (size_t)4 * ((XMODE == 0 ? 0 : a.kx) + NL * NT + 4) * 64 * sizeof(float);
    int rc = set_lds(lstm_seq_kernel<NT, NL, XMODE>, lds);
    if (rc) return rc;
    const long blocks = (a.n_rowtiles + 3) / 4;
    if (blocks > 0x7fffffffL) { set_error("lstm_seq: too many rows"); return -2; }
    hipLaunchKernelGGL((lstm_seq_kernel<NT, NL, XMODE>), dim3((unsigned)blocks), dim3(256), lds, st, a);
    return launch_check("lstm_seq_kernel");
}

template <int XMODE>
static int launch_lstm(const LstmSeqArgs& a, int NT, int NL, hipStream_t st) {
    if (a.n_rowtiles <= 0 || a.T <= 0) return 0;
#define VSV_CASE(nt)                                                         \
    case nt:                                                                  \
        return NL == 2 ? launch_lstm_t<nt, 2, XMODE>(a, st) : launch_lstm_t<nt, 1, XMODE>(a, st);
    switch (NT) {
        VSV_CASE(25)
        VSV_CASE(32)
        default:
            set_error("unsupported LSTM hidden size: %d tiles of 4 units (built: 25 = hidden 97..100, 32 = hidden 125..128)", NT);
            return -4;
    }
#undef VSV_CASE
}

static thread_local hipEvent_t g_prof_ev[2] = {nullptr, nullptr};

static void fill_dense(DenseChainArgs& a, int li, const float* packed, const Sec& s) {
    a.layer[li].w = packed + s.w_off;
    a.layer[li].b = packed + s.b_off;
    a.layer[li].nq = s.nq;
    a.layer[li].nt = s.nt;
    a.layer[li].n_relu = s.n_relu;
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
    const long rt = (n_rows + 15) / 16;
    const int kx = lay.enc[0].kx > lay.enc[1].kx ? lay.enc[0].kx : lay.enc[1].kx;
    return align256((size_t)rt * T * kx * 64 * sizeof(float)) + 256;
}

extern "C" int vsv_encode(const vsv_dims* dims, const float* packed, int which,
                          const float* surf, long surf_row_stride, long surf_t_stride,
                          const float* ex, long ex_row_stride, long ex_t_stride,
                          long n_rows, int T, float* out, void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_encode")
    if (!packed || !surf || !out || !ws) { set_error("vsv_encode: null pointer"); return -1; }
    if (which < 0 || which > 1) { set_error("vsv_encode: which must be 0 or 1"); return -1; }
    if (n_rows < 0 || T < 0) { set_error("vsv_encode: negative size"); return -1; }
    if (lay.X > 0 && !ex) { set_error("vsv_encode: ex_feats_dim > 0 but ex is null"); return -1; }
    if (n_rows == 0 || T == 0) return 0;
    if (ws_bytes < vsv_encode_ws_bytes(dims, n_rows, T)) { set_error("vsv_encode: workspace too small"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const EncLayout& e = lay.enc[which];
    const long rt = (n_rows + 15) / 16;
    float* xe = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);

    DenseChainArgs a = {};
    a.n_layers = e.n_conv;
    for (int i = 0; i < e.n_conv; ++i) fill_dense(a, i, packed, e.conv[i]);
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = T;
    a.in_mode = 0; a.in = surf; a.in_rs = surf_row_stride; a.in_ts = surf_t_stride; a.in_f = 25;
    a.in_ks = 4 * e.conv[0].nq;
    if (lay.X > 0) { a.inj = ex; a.inj_rs = ex_row_stride; a.inj_ts = ex_t_stride; a.inj_f0 = e.ex_f0; a.inj_n = lay.X; }
    a.out_mode = 0; a.out = xe; a.out_f = e.E; a.S = 1;
    a.aux_layer = -1;
    int rc = launch_dense_chain(a, st);
    if (rc) return rc;

    LstmSeqArgs l = {};
    for (int i = 0; i < lay.NL; ++i) {
        l.w[i] = packed + e.lstm[i].w_off; l.b[i] = packed + e.lstm[i].b_off; l.nq[i] = e.lstm[i].nq;
    }
    l.kx = e.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.x_tileb = xe; l.L = lay.L;
    l.t_out0 = T;
    l.head_w = packed + e.head.w_off; l.head_b = packed + e.head.b_off; l.head_nq = e.head.nq; l.head_nt = e.head.nt; l.nh = e.nh;
    l.head_out = out;
    return launch_lstm<0>(l, lay.NTm, lay.NL, st);
}

extern "C" size_t vsv_decode_ws_bytes(const vsv_dims* dims, long n_rows, int T, int t_out0) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0 || n_rows < 0 || T < 0) return 0;
    const long rt = (n_rows + 15) / 16;
    const int touts = T - t_out0 > 0 ? T - t_out0 : 0;
    return align256((size_t)rt * touts * lay.NTd * 64 * sizeof(float)) + 256;
}

extern "C" int vsv_decode(const vsv_dims* dims, const float* packed, const vsv_decode_in* in,
                          long n_rows, int T, int t_out0, float* state_out, float* z_out,
                          float* surf_out, int out_mode, float* ex_out,
                          void* ws, size_t ws_bytes, void* stream) {
    VSV_LAYOUT_OR_FAIL("vsv_decode")
    if (!packed || !in) { set_error("vsv_decode: null pointer"); return -1; }
    if (n_rows < 0 || T < 0 || t_out0 < 0 || t_out0 > T) { set_error("vsv_decode: bad sizes"); return -1; }
    if (n_rows == 0 || T == 0) return 0;
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
    if (out_mode < 0 || out_mode > 1) { set_error("vsv_decode: out_mode must be 0 or 1"); return -1; }
    if (out_mode == 1 && touts > 0 && (in->S < 1 || n_rows % in->S != 0)) { set_error("vsv_decode: out_mode 1 needs n_rows %% S == 0"); return -1; }
    hipStream_t st = (hipStream_t)stream;
    const DecLayout& dc = lay.dec;
    const long rt = (n_rows + 15) / 16;
    float* hbuf = ws ? (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255) : nullptr;

    LstmSeqArgs l = {};
    for (int i = 0; i < lay.NL; ++i) {
        l.w[i] = packed + dc.lstm[i].w_off; l.b[i] = packed + dc.lstm[i].b_off; l.nq[i] = dc.lstm[i].nq;
    }
    l.kx = dc.kx; l.n_rows = n_rows; l.n_rowtiles = rt; l.T = T;
    l.L = lay.L; l.din = *in; l.z_out = z_out;
    if (l.din.ctx_div < 1) l.din.ctx_div = 1;
    if (l.din.z_div < 1) l.din.z_div = 1;
    if (l.din.S < 1) l.din.S = 1;
    if (l.din.state_div < 1) l.din.state_div = 1;
    l.state_out = state_out;
    l.hout = touts > 0 ? hbuf : nullptr;
    l.t_out0 = t_out0;
    hipEvent_t ev0 = g_prof_ev[0], ev1 = g_prof_ev[1];
    g_prof_ev[0] = g_prof_ev[1] = nullptr;
    if (ev0) (void)hipEventRecord(ev0, st);
    int rc = launch_lstm<1>(l, lay.NTd, lay.NL, st);
    if (ev1) (void)hipEventRecord(ev1, st);
    if (rc || touts == 0) return rc;

    DenseChainArgs a = {};
    a.n_layers = dc.n_head;
    for (int i = 0; i < dc.n_head; ++i) fill_dense(a, i, packed, dc.head[i]);
    a.n_rows = n_rows; a.n_rowtiles = rt; a.T = touts;
    a.in_mode = 1; a.in = hbuf; a.in_ks = lay.NTd;
    a.out_mode = out_mode == 0 ? 1 : 2; a.out = surf_out; a.out_f = dc.n_surf; a.S = l.din.S;
    a.aux_layer = (lay.X > 0 && ex_out) ? dc.ex_layer : -1;
    a.aux = ex_out; a.aux_f0 = dc.ex_f0; a.aux_n = lay.X;
    return launch_dense_chain(a, st);
}

extern "C" void vsv_profile_next_decode(void* ev_start, void* ev_stop) {
    g_prof_ev[0] = (hipEvent_t)ev_start;
    g_prof_ev[1] = (hipEvent_t)ev_stop;
}

extern "C" int vsv_percentile(const float* in, long n_vec, int S, const double* q_host, int nq,
                              int inner, float* out, void* stream) {
    if (!in || !out || !q_host) { set_error("vsv_percentile: null pointer"); return -1; }
    if (nq < 1 || nq > 8) { set_error("vsv_percentile: nq must be in [1,8]"); return -1; }
    if (S < 1 || S > 16384) { set_error("vsv_percentile: S must be in [1,16384]"); return -1; }
    if (n_vec < 0 || inner < 1 || n_vec % inner != 0) { set_error("vsv_percentile: n_vec must be a multiple of inner"); return -1; }
    if (n_vec == 0) return 0;
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
    int P2 = 2;
    while (P2 < S) P2 <<= 1;
    const size_t lds = (size_t)P2 * sizeof(float);
    int rc = set_lds(percentile_kernel, lds);
    if (rc) return rc;
    const long maxb = 256L * 16;
    const unsigned blocks = (unsigned)(n_vec < maxb ? n_vec : maxb);
    hipLaunchKernelGGL(percentile_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, in, n_vec, S, P2, pa, inner, out);
    return launch_check("percentile_kernel");
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
