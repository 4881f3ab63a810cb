// Host side of libvsv_hip.so: shape validation, blob layout and weight packing.
// No device code here.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>

#include "vsv_layout.h"

namespace vsv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

static Sec alloc_sec(size_t& cur, int nq, int nt, int n_out, int n_relu) {
    Sec s;
    s.nq = nq;
    s.nt = nt;
    s.n_out = n_out;
    s.n_relu = n_relu;
    s.w_off = cur;
    cur += (size_t)nq * nt * 256;
    s.b_off = cur;
    cur += (size_t)nt * 32;
    for (int t = 0; t < 4; ++t) { s.qlo[t] = 0; s.qhi[t] = nq; }
    return s;
}

// Hidden features of the decoder head's conv layers are stored PIXEL-major (packed index p = pixel * C +
// channel <-> reference feature channel * 25 + pixel; the ex-feature riders behind them keep their place):
// a 3x3 conv / conv-transpose output pixel reads its 3x3 neighbourhood, so in this order an output tile's
// non-zero weights form a contiguous band of input features.
static inline int px2orig(int p, int C) { return p < 25 * C ? (p % C) * 25 + p / C : p; }

// band of K quads (8 input features each) read by the 32 outputs of tile t of a cin -> cout conv layer
static void conv_band(Sec& s, int cin, int cout) {
    for (int t = 0; t < s.nt && t < 4; ++t) {
        int lo = 1 << 30, hi = -1;
        for (int p = 32 * t; p < 32 * t + 32 && p < 25 * cout; ++p) {
            const int y = (p / cout) / 5, x = (p / cout) % 5;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= 5 || xx < 0 || xx >= 5) continue;
                    const int f0 = (yy * 5 + xx) * cin;
                    lo = f0 < lo ? f0 : lo;
                    hi = f0 + cin - 1 > hi ? f0 + cin - 1 : hi;
                }
        }
        if (hi < 0) { s.qlo[t] = 0; s.qhi[t] = 1; continue; }
        s.qlo[t] = lo / 8;
        s.qhi[t] = hi / 8 + 1;
        if (s.qhi[t] > s.nq) s.qhi[t] = s.nq;
    }
}

int make_layout(const vsv_dims& d, Layout& lay, const char** msg) {
    static thread_local char why[256];
    auto fail = [&](const char* m) {
        snprintf(why, sizeof(why), "%s", m);
        if (msg) *msg = why;
        return -1;
    };
    lay = Layout();
    lay.L = d.latent_dim;
    lay.X = d.ex_feats_dim;
    lay.Hm = d.mem_hidden;
    lay.NL = d.mem_layers;
    lay.Q = d.n_quantiles;
    if (lay.L < 1 || lay.L > 16) return fail("latent_dim must be in [1,16]");
    if (lay.X < 0 || lay.X > 16) return fail("ex_feats_dim must be in [0,16]");
    if (lay.NL < 1 || lay.NL > 2) return fail("mem_layers must be 1 or 2");
    if (lay.Hm < 4 || lay.Hm > 128) return fail("mem_hidden must be in [4,128]");
    if (lay.Q < 1 || lay.Q * 25 > kMaxFeat) return fail("n_quantiles*25 must be <= 128");
    if (d.interaction_layers < 0 || d.interaction_layers + 1 > VSV_MAX_INTERACT) return fail("interaction_layers must be in [0,2]");
    lay.NI = d.interaction_layers > 0 ? d.interaction_layers + 1 : 0;
    if (d.mem_type < 0 || d.mem_type > 2) return fail("mem_type must be 0 (lstm), 1 (gru) or 2 (rnn)");
    lay.gru = d.mem_type;
    lay.Lc = d.compress_context ? lay.L : lay.Hm;
    if (d.n_conv < 1 || d.n_conv > VSV_MAX_CONV || d.n_ctx_conv < 1 || d.n_ctx_conv > VSV_MAX_CONV)
        return fail("1..4 conv layers supported");
    if (d.n_conv + lay.NI > kMaxDenseLayers || d.n_ctx_conv + lay.NI > kMaxDenseLayers || 2 + d.n_conv + lay.NI > kMaxDenseLayers)
        return fail("too many conv + interaction layers for one dense chain (8)");
    lay.dense = d.use_dense_surface != 0;
    // feature count of an embedding / head layer with `c` channels (conv: c maps of 5x5) or `c` units (dense)
    const int per = lay.dense ? 1 : 25;
    for (int i = 0; i < d.n_conv; ++i)
        if (d.surface_hidden[i] < 1 || d.surface_hidden[i] * per + (lay.dense ? 0 : lay.X) > kMaxFeat)
            return fail(lay.dense ? "surface_hidden[i] must be in [1,128]" : "surface_hidden[i]*25 + ex_feats_dim must be <= 128");
    for (int i = 0; i < d.n_ctx_conv; ++i)
        if (d.ctx_surface_hidden[i] < 1 || d.ctx_surface_hidden[i] * per + (lay.dense ? 0 : lay.X) > kMaxFeat)
            return fail(lay.dense ? "ctx_surface_hidden[i] must be in [1,128]" : "ctx_surface_hidden[i]*25 + ex_feats_dim must be <= 128");
    // gate tiles of 8 hidden units.  The recurrent kernels are built for 13 and 16 tiles (hidden 97..104 and
    // 121..128, the widths of every reference config); any other width runs on the next built size: the
    // extra units have all-zero weights and biases, so their gates are sigmoid(0) / tanh(0), their c and h
    // stay exactly 0 and nothing reads them (zero columns downstream)
    auto built_tiles = [](int nt) { return nt <= 13 ? 13 : 16; };
    lay.NTm = built_tiles(cdiv(lay.Hm, 8));
    const int c_last = d.surface_hidden[d.n_conv - 1];
    if (lay.X > 0 && !lay.dense) {
        // the kernels look for the ex-feature slots only in the last (4th) tile of a layer
        const int cc = d.ctx_surface_hidden[d.n_ctx_conv - 1];
        const int c0 = d.n_conv > 1 ? d.surface_hidden[d.n_conv - 2] : c_last;
        if (c_last * 25 < 96 || cc * 25 < 96 || c0 * 25 < 96)
            return fail("with ex_feats_dim > 0 the last conv layers need >= 4 channels (ex slots live in features 96..127)");
    }
    if (lay.X > 0 && lay.dense) {
        // dense surfaces: the ex-feature slots sit at kDenseEx0 = 96 .. 96 + X of their layers (the last tile, where
        // the kernels look for them), the surface units below them
        const int cc = d.ctx_surface_hidden[d.n_ctx_conv - 1];
        const int c0 = d.n_conv > 1 ? d.surface_hidden[d.n_conv - 2] : 25;
        if (c_last > kDenseEx0 || cc > kDenseEx0 || c0 > kDenseEx0)
            return fail("use_dense_surface with ex_feats_dim > 0: the last embedding layers must be <= 96 wide");
    }
    lay.Hd = c_last * per + lay.X;
    lay.NTd = built_tiles(cdiv(lay.Hd, 8));

    size_t cur = 0;
    for (int which = 0; which < 2; ++which) {
        EncLayout& e = lay.enc[which];
        const int* ch = which ? d.ctx_surface_hidden : d.surface_hidden;
        e.n_conv = which ? d.n_ctx_conv : d.n_conv;
        int nt_prev = 0;
        // first ex-feature slot of the embedding: right behind the surface features, or (dense surfaces) at 96
        e.ex_f0 = (lay.dense && lay.X > 0) ? kDenseEx0 : ch[e.n_conv - 1] * per;
        const int e_ext = e.ex_f0 + lay.X;                     // extent of the embedding in feature slots
        for (int i = 0; i < e.n_conv; ++i) {
            const bool last = (i == e.n_conv - 1);
            const int n_out = last ? e_ext : ch[i] * per;
            const int nt = cdiv(n_out, 32);
            const int nq = (i == 0) ? cdiv(cdiv(25, 2), 4) : 4 * nt_prev;
            e.conv[i] = alloc_sec(cur, nq, nt, n_out, ch[i] * per);
            nt_prev = nt;
        }
        e.E = ch[e.n_conv - 1] * per + lay.X;
        e.n_chain = e.n_conv;
        for (int j = 0; j < lay.NI; ++j) {
            // interaction Linear j: E -> E, ReLU on all but the final one
            const int nt = cdiv(e_ext, 32);
            e.conv[e.n_chain++] = alloc_sec(cur, 4 * nt_prev, nt, e_ext, j < lay.NI - 1 ? e_ext : 0);
            nt_prev = nt;
        }
        e.kx = 16 * nt_prev;
        e.lstm[0] = alloc_sec(cur, e.kx / 4 + lay.NTm, lay.NTm, 4 * lay.Hm, 0);
        if (lay.NL == 2) e.lstm[1] = alloc_sec(cur, 2 * lay.NTm, lay.NTm, 4 * lay.Hm, 0);
        e.nh = which ? lay.Lc : 2 * lay.L;
        e.head = alloc_sec(cur, lay.NTm, cdiv(e.nh, 32), e.nh, 0);
    }
    DecLayout& dc = lay.dec;
    dc.kx = 4 * cdiv(lay.Lc + lay.L, 8);              // Lc + L features, 2 per k-step, rounded up to a quad
    dc.lstm[0] = alloc_sec(cur, dc.kx / 4 + lay.NTd, lay.NTd, 4 * lay.Hd, 0);
    if (lay.NL == 2) dc.lstm[1] = alloc_sec(cur, 2 * lay.NTd, lay.NTd, 4 * lay.Hd, 0);
    // head: Linear(Hd -> c_last*25 [+X]) ; ConvT x n_conv ; Conv -> Q*25.
    // All layers but the last are padded to kHeadTiles tiles (zero rows) so the fused
    // decoder kernel can chain them through registers with compile-time shapes.
    int nh = 0;
    for (int j = 0; j < lay.NI; ++j)       // interaction Linears on the LSTM output: Hd -> Hd
        dc.head[nh++] = alloc_sec(cur, j == 0 ? lay.NTd : 4 * kHeadTiles, kHeadTiles, lay.Hd, j < lay.NI - 1 ? lay.Hd : 0);
    {
        const int n_out = (lay.dense && lay.X > 0) ? kDenseEx0 + lay.X : c_last * per + lay.X;
        dc.head[nh++] = alloc_sec(cur, lay.NI > 0 ? 4 * kHeadTiles : lay.NTd, kHeadTiles, n_out, 0);
    }
    int cin = c_last;
    for (int i = 0; i < d.n_conv; ++i) {
        // dec_deconv_i / dec_dense_i maps reversed(surface_hidden[:-1])[i]; dec_final keeps the channel count
        // (conv) or maps to the 25 grid points (dense, :401-404)
        const int cout = (i < d.n_conv - 1) ? d.surface_hidden[d.n_conv - 2 - i] : (lay.dense ? 25 : cin);
        const bool carries_ex = (i == 0 && lay.X > 0);
        const int f_ex = lay.dense ? kDenseEx0 : cout * 25;
        const int n_out = carries_ex ? f_ex + lay.X : cout * per;
        if (carries_ex) {
            dc.ex_layer = nh;
            dc.ex_f0 = f_ex;
        }
        dc.head[nh++] = alloc_sec(cur, 4 * kHeadTiles, kHeadTiles, n_out, cout * per);
        if (i > 0 && !lay.dense) conv_band(dc.head[nh - 1], cin, cout);      // (layer 0 carries the ex riders and is folded)
        cin = cout;
    }
    dc.head[nh++] = alloc_sec(cur, 4 * kHeadTiles, cdiv(lay.Q * 25, 32), lay.Q * 25, 0);
    dc.n_head = nh;
    dc.n_surf = lay.Q * 25;
    // the folded Linear . ConvT layer (appended, so every other offset stays where it was)
    dc.fold_at = lay.NI;
    dc.head_fold = alloc_sec(cur, dc.head[dc.fold_at].nq, kHeadTiles, dc.head[dc.fold_at + 1].n_out,
                             dc.head[dc.fold_at + 1].n_relu);
    // the output Conv one channel at a time (appended likewise)
    for (int c = 0; c < lay.Q; ++c) dc.head_ch[c] = alloc_sec(cur, 4 * kHeadTiles, 1, 25, 0);
    lay.total = cur;
    return 0;
}

// ---- generic fragment packer ------------------------------------------------
using OutMap = std::function<int(int t, int i)>;            // -> output row or -1
using InMap = std::function<int(int s, int kg)>;            // -> input column or -1
using WFn = std::function<float(int row, int col)>;
using BFn = std::function<float(int row)>;

static void pack_sec(float* blob, const Sec& s, const OutMap& fo, const InMap& fi, const WFn& W, const BFn& B) {
    float* w = blob + s.w_off;
    for (int q = 0; q < s.nq; ++q)
        for (int t = 0; t < s.nt; ++t)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int r = fo(t, l & 31), c = fi(4 * q + j, l >> 5);
                    w[(((size_t)q * s.nt + t) * 64 + l) * 4 + j] = (r >= 0 && c >= 0) ? W(r, c) : 0.0f;
                }
    float* b = blob + s.b_off;
    for (int t = 0; t < s.nt; ++t)
        for (int hh = 0; hh < 2; ++hh)
            for (int reg = 0; reg < 16; ++reg) {
                const int r = fo(t, 8 * (reg >> 2) + 4 * hh + (reg & 3));
                b[((size_t)t * 2 + hh) * 16 + reg] = (r >= 0) ? B(r) : 0.0f;
            }
}

static OutMap out_nat32(int n_out) {
    return [=](int t, int i) { const int f = 32 * t + i; return f < n_out ? f : -1; };
}
// LSTM gates: tile t holds units 8t..8t+7; D row i = 4*unit_local + gate
static OutMap out_gate(int H) {
    return [=](int t, int i) { const int u = 8 * t + (i >> 2); return u < H ? (i & 3) * H + u : -1; };
}
static InMap in_nat(int n_in) {
    return [=](int s, int hh) { const int f = 2 * s + hh; return f < n_in ? f : -1; };
}
static OutMap out_px(int n_out, int C) {
    return [=](int t, int i) { const int f = 32 * t + i; return f < n_out ? px2orig(f, C) : -1; };
}
static InMap in_perm(int n_in);
static InMap in_perm_px(int n_in, int C) {
    return [=](int s, int hh) {
        const int f = 32 * (s >> 4) + 8 * ((s >> 2) & 3) + 4 * hh + (s & 3);
        return f < n_in ? px2orig(f, C) : -1;
    };
}
static InMap in_perm(int n_in) {
    return [=](int s, int hh) {
        const int f = 32 * (s >> 4) + 8 * ((s >> 2) & 3) + 4 * hh + (s & 3);
        return f < n_in ? f : -1;
    };
}
// Dense surfaces (use_dense_surface): a layer's h surface units sit at slots 0..h-1 and the X ex-feature riders at
// slots x0..x0+X-1 (x0 = kDenseEx0 >= h); index = position in the reference's [surface | ex] concatenation.
static inline int gap2orig(int f, int h, int X, int x0) { return f < h ? f : (f >= x0 && f < x0 + X ? h + (f - x0) : -1); }
static OutMap out_gap(int h, int X, int x0) {
    return [=](int t, int i) { return gap2orig(32 * t + i, h, X, x0); };
}
static InMap in_perm_gap(int h, int X, int x0) {
    return [=](int s, int hh) { return gap2orig(32 * (s >> 4) + 8 * ((s >> 2) & 3) + 4 * hh + (s & 3), h, X, x0); };
}

// Dense (Toeplitz) matrix of a 3x3, stride 1, pad 1 conv on a 5x5 grid.
// transposed=false: nn.Conv2d weight (cout,cin,3,3); true: nn.ConvTranspose2d weight (cin,cout,3,3).
static std::vector<float> conv_matrix(const float* w, int cin, int cout, bool transposed) {
    std::vector<float> M((size_t)cout * 25 * cin * 25, 0.0f);
    for (int co = 0; co < cout; ++co)
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x)
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int yi = transposed ? y + 1 - ky : y + ky - 1;
                            const int xi = transposed ? x + 1 - kx : x + kx - 1;
                            if (yi < 0 || yi >= 5 || xi < 0 || xi >= 5) continue;
                            const float v = transposed ? w[((ci * cout + co) * 3 + ky) * 3 + kx]
                                                       : w[((co * cin + ci) * 3 + ky) * 3 + kx];
                            M[(size_t)(co * 25 + y * 5 + x) * (cin * 25) + ci * 25 + yi * 5 + xi] = v;
                        }
    return M;
}

// nn.GRU in the four gate slots of a hidden unit: (r, z, n_x, n_h).  r and z sum their input and
// recurrent parts like an LSTM gate; the candidate's two halves stay apart -- n_x = W_in x + b_in has
// zero recurrent columns, n_h = W_hn h + b_hn zero input columns -- because the cell needs
// n = tanh(n_x + r * n_h).  Row code g*H + u as for the LSTM (g = slot).
static void pack_gru(float* blob, const Sec* secs, const vsv_lstm_raw& raw, int NL, int H, int NT, int in0,
                     int kx, const InMap& x_map) {
    for (int l = 0; l < NL; ++l) {
        const float *wih = raw.w_ih[l], *whh = raw.w_hh[l], *bih = raw.b_ih[l], *bhh = raw.b_hh[l];
        const int n_in = (l == 0) ? in0 : H;
        const int kin = (l == 0) ? kx : 4 * NT;
        const InMap imap = (l == 0) ? x_map : in_nat(H);
        const InMap hmap = in_nat(H);
        InMap fi = [=](int s, int hh) {
            if (s < kin) return imap(s, hh);
            const int c = hmap(s - kin, hh);
            return c < 0 ? -1 : n_in + c;
        };
        WFn W = [=](int r, int c) {
            const int g = r / H, u = r % H;
            const int src = (g < 2 ? g : 2) * H + u;                    // row of the 3H-row matrices
            if (c < n_in) return g == 3 ? 0.0f : wih[(size_t)src * n_in + c];
            return g == 2 ? 0.0f : whh[(size_t)src * H + (c - n_in)];
        };
        BFn B = [=](int r) {
            const int g = r / H, u = r % H;
            if (g < 2) return bih[g * H + u] + bhh[g * H + u];
            return g == 2 ? bih[2 * H + u] : bhh[2 * H + u];
        };
        pack_sec(blob, secs[l], out_gate(H), fi, W, B);
    }
}

// nn.RNN (tanh) in slot 0 of a hidden unit's four gate slots: pre = W_ih x + b_ih + W_hh h + b_hh, h' = tanh(pre);
// slots 1-3 are zero rows (the gate GEMM keeps the LSTM's shape, the cell reads slot 0 only).
static void pack_rnn(float* blob, const Sec* secs, const vsv_lstm_raw& raw, int NL, int H, int NT, int in0,
                     int kx, const InMap& x_map) {
    for (int l = 0; l < NL; ++l) {
        const float *wih = raw.w_ih[l], *whh = raw.w_hh[l], *bih = raw.b_ih[l], *bhh = raw.b_hh[l];
        const int n_in = (l == 0) ? in0 : H;
        const int kin = (l == 0) ? kx : 4 * NT;
        const InMap imap = (l == 0) ? x_map : in_nat(H);
        const InMap hmap = in_nat(H);
        InMap fi = [=](int s, int hh) {
            if (s < kin) return imap(s, hh);
            const int c = hmap(s - kin, hh);
            return c < 0 ? -1 : n_in + c;
        };
        WFn W = [=](int r, int c) {
            const int g = r / H, u = r % H;
            if (g != 0) return 0.0f;
            return c < n_in ? wih[(size_t)u * n_in + c] : whh[(size_t)u * H + (c - n_in)];
        };
        BFn B = [=](int r) { return r < H ? bih[r] + bhh[r] : 0.0f; };
        pack_sec(blob, secs[l], out_gate(H), fi, W, B);
    }
}

static void pack_lstm(float* blob, const Sec* secs, const vsv_lstm_raw& raw, int NL, int H, int NT, int in0,
                      int kx, const InMap& x_map) {
    for (int l = 0; l < NL; ++l) {
        const float *wih = raw.w_ih[l], *whh = raw.w_hh[l], *bih = raw.b_ih[l], *bhh = raw.b_hh[l];
        const int n_in = (l == 0) ? in0 : H;
        const int kin = (l == 0) ? kx : 4 * NT;        // k-steps of the input part
        const InMap imap = (l == 0) ? x_map : in_nat(H);
        const InMap hmap = in_nat(H);
        // combined column space: [0, n_in) -> w_ih, [n_in, n_in+H) -> w_hh
        InMap fi = [=](int s, int hh) {
            if (s < kin) return imap(s, hh);
            const int c = hmap(s - kin, hh);
            return c < 0 ? -1 : n_in + c;
        };
        WFn W = [=](int r, int c) { return c < n_in ? wih[(size_t)r * n_in + c] : whh[(size_t)r * H + (c - n_in)]; };
        BFn B = [=](int r) { return bih[r] + bhh[r]; };
        pack_sec(blob, secs[l], out_gate(H), fi, W, B);
    }
}

}  // namespace vsv

using namespace vsv;

extern "C" int vsv_abi_version(void) { return VSV_ABI_VERSION; }
extern "C" const char* vsv_last_error(void) { return g_err; }

extern "C" size_t vsv_packed_floats(const vsv_dims* dims) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0) {
        set_error("vsv_packed_floats: %s", msg ? msg : "null dims");
        return 0;
    }
    return lay.total;
}

extern "C" size_t vsv_state_floats(const vsv_dims* dims) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0) {
        set_error("vsv_state_floats: %s", msg ? msg : "null dims");
        return 0;
    }
    return (size_t)lay.NL * 2 * 8 * lay.NTd;
}

extern "C" size_t vsv_enc_state_floats(const vsv_dims* dims) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0) {
        set_error("vsv_enc_state_floats: %s", msg ? msg : "null dims");
        return 0;
    }
    return (size_t)lay.NL * 2 * 8 * lay.NTm;
}

extern "C" size_t vsv_gate_floats(const vsv_dims* dims) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0) {
        set_error("vsv_gate_floats: %s", msg ? msg : "null dims");
        return 0;
    }
    return (size_t)lay.NL * lay.NTd * 32;
}

extern "C" int vsv_pack_weights(const vsv_dims* dims, const vsv_raw_weights* raw, float* out) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || !raw || !out) {
        set_error("vsv_pack_weights: null argument");
        return -1;
    }
    if (make_layout(*dims, lay, &msg) != 0) {
        set_error("vsv_pack_weights: %s", msg);
        return -1;
    }
    const vsv_dims& d = *dims;
    memset(out, 0, lay.total * sizeof(float));
    const int L = lay.L, X = lay.X, Hm = lay.Hm;

    for (int which = 0; which < 2; ++which) {
        const EncLayout& e = lay.enc[which];
        const int* ch = which ? d.ctx_surface_hidden : d.surface_hidden;
        const float* const* cw = which ? raw->ctx_conv_w : raw->enc_conv_w;
        const float* const* cb = which ? raw->ctx_conv_b : raw->enc_conv_b;
        const int h_last = ch[e.n_conv - 1];
        // slot -> column of the reference's [surface embedding | ex-features] concatenation (:188)
        const InMap emb_in = lay.dense ? in_perm_gap(h_last, X, e.ex_f0) : in_perm(e.E);
        const OutMap emb_out = lay.dense ? out_gap(h_last, X, e.ex_f0) : out_nat32(e.E);
        if (lay.dense) {
            // Flatten + Linear(in -> surface_hidden[i]) + ReLU (:84-93, :237-246); nn.Linear weights are (out, in)
            int n_in = 25;
            for (int i = 0; i < e.n_conv; ++i) {
                if (!cw[i] || !cb[i]) { set_error("vsv_pack_weights: missing dense embedding weight"); return -1; }
                const int n_out = ch[i];
                const float *w = cw[i], *bias = cb[i];
                WFn W = [=](int r, int c) { return w[(size_t)r * n_in + c]; };
                BFn B = [=](int r) { return bias[r]; };
                // (the ex-feature slots of the last layer stay zero rows: out map limited to n_out)
                pack_sec(out, e.conv[i], out_nat32(n_out), i == 0 ? in_nat(n_in) : in_perm(n_in), W, B);
                n_in = n_out;
            }
        } else {
            int cin = 1;
            for (int i = 0; i < e.n_conv; ++i) {
                if (!cw[i] || !cb[i]) { set_error("vsv_pack_weights: missing conv weight"); return -1; }
                const int cout = ch[i];
                std::vector<float> M = conv_matrix(cw[i], cin, cout, false);
                const int n_in = cin * 25, n_out = cout * 25;
                const float* bias = cb[i];
                WFn W = [&, n_in](int r, int c) { return M[(size_t)r * n_in + c]; };
                BFn B = [=](int r) { return bias[r / 25]; };
                // rows >= n_out (ex-feature slots of the last layer) stay zero: out map limited to n_out
                pack_sec(out, e.conv[i], out_nat32(n_out), i == 0 ? in_nat(n_in) : in_perm(n_in), W, B);
                cin = cout;
            }
        }
        for (int j = 0; j < lay.NI; ++j) {
            const float* w = (which ? raw->ctx_int_w : raw->enc_int_w)[j];
            const float* b = (which ? raw->ctx_int_b : raw->enc_int_b)[j];
            if (!w || !b) { set_error("vsv_pack_weights: missing interaction weight"); return -1; }
            const int E = e.E;
            WFn W = [=](int r, int c) { return w[(size_t)r * E + c]; };
            BFn B = [=](int r) { return b[r]; };
            pack_sec(out, e.conv[e.n_conv + j], emb_out, emb_in, W, B);
        }
        const vsv_lstm_raw& mem = which ? raw->ctx_mem : raw->enc_mem;
        (lay.gru == 2 ? pack_rnn : lay.gru ? pack_gru : pack_lstm)(out, e.lstm, mem, lay.NL, Hm, lay.NTm, e.E, e.kx, emb_in);
        if (which == 0) {
            const float *wm = raw->z_mean_w, *bm = raw->z_mean_b, *wv = raw->z_log_var_w, *bv = raw->z_log_var_b;
            WFn W = [=](int r, int c) { return r < L ? wm[(size_t)r * Hm + c] : wv[(size_t)(r - L) * Hm + c]; };
            BFn B = [=](int r) { return r < L ? bm[r] : bv[r - L]; };
            pack_sec(out, e.head, out_nat32(2 * L), in_nat(Hm), W, B);
        } else {
            const float *wc = raw->ctx_comp_w, *bc = raw->ctx_comp_b;
            if (d.compress_context) {
                if (!wc || !bc) { set_error("vsv_pack_weights: missing final_compression"); return -1; }
                WFn W = [=](int r, int c) { return wc[(size_t)r * Hm + c]; };
                BFn B = [=](int r) { return bc[r]; };
                pack_sec(out, e.head, out_nat32(L), in_nat(Hm), W, B);
            } else {
                // compress_context = False (:363-366 absent): the embedding is the memory's output itself -- an
                // identity "head", so the kernels' chain (tile-B h -> row-major embedding) stays the same
                WFn W = [=](int r, int c) { return r == c ? 1.0f : 0.0f; };
                BFn B = [=](int) { return 0.0f; };
                pack_sec(out, e.head, out_nat32(Hm), in_nat(Hm), W, B);
            }
        }
    }

    const DecLayout& dc = lay.dec;
    const int Hd = lay.Hd;
    (lay.gru == 2 ? pack_rnn : lay.gru ? pack_gru : pack_lstm)(out, dc.lstm, raw->dec_mem, lay.NL, Hd, lay.NTd, lay.Lc + L, dc.kx,
                                                                 in_nat(lay.Lc + L));
    int hi = 0;
    const int c_last = d.surface_hidden[d.n_conv - 1];
    for (int j = 0; j < lay.NI; ++j) {
        const float *w = raw->dec_int_w[j], *b = raw->dec_int_b[j];
        if (!w || !b) { set_error("vsv_pack_weights: missing decoder interaction weight"); return -1; }
        WFn W = [=](int r, int c) { return w[(size_t)r * Hd + c]; };
        BFn B = [=](int r) { return b[r]; };
        pack_sec(out, dc.head[hi++], out_nat32(Hd), j == 0 ? in_nat(Hd) : in_perm(Hd), W, B);
    }
    WFn W_lin;
    BFn B_lin;
    const int per = lay.dense ? 1 : 25;
    const int n_lin = c_last * per + X;
    // where a head layer with `c` channels / units keeps its features: conv layers pixel-major with the ex riders
    // right behind, dense layers in natural order with the riders at kDenseEx0
    auto head_out = [&](int n_surf, int n_ex, int c) -> OutMap {
        return lay.dense ? out_gap(n_surf, n_ex, kDenseEx0) : out_px(n_surf + n_ex, c);
    };
    auto head_in = [&](int n_surf, int n_ex, int c) -> InMap {
        return lay.dense ? in_perm_gap(n_surf, n_ex, kDenseEx0) : in_perm_px(n_surf + n_ex, c);
    };
    {   // Linear: rows [surface_decoder_input ; ex_feats_decoder_input]
        const int ns = c_last * per;
        const float *ws = raw->dec_in_w, *bs = raw->dec_in_b, *we = raw->ex_in_w, *be = raw->ex_in_b;
        if (X > 0 && (!we || !be)) { set_error("vsv_pack_weights: missing ex_feats_decoder_input"); return -1; }
        W_lin = [=](int r, int c) { return r < ns ? ws[(size_t)r * Hd + c] : we[(size_t)(r - ns) * Hd + c]; };
        B_lin = [=](int r) { return r < ns ? bs[r] : be[r - ns]; };
        pack_sec(out, dc.head[hi++], head_out(ns, X, c_last), lay.NI > 0 ? in_perm(Hd) : in_nat(Hd), W_lin, B_lin);
    }
    int cin = c_last;
    for (int i = 0; i < d.n_conv; ++i) {
        const int cout = (i < d.n_conv - 1) ? d.surface_hidden[d.n_conv - 2 - i] : (lay.dense ? 25 : cin);
        if (!raw->dec_deconv_w[i] || !raw->dec_deconv_b[i]) { set_error("vsv_pack_weights: missing deconv weight"); return -1; }
        std::vector<float> M;
        if (!lay.dense) M = conv_matrix(raw->dec_deconv_w[i], cin, cout, true);
        const float* wd = raw->dec_deconv_w[i];                  // dense: nn.Linear (cout, cin), :397-404
        const int n_in = cin * per, n_out = cout * per;
        const float* bias = raw->dec_deconv_b[i];
        const bool dense = lay.dense;
        const bool carries_ex = (i == 0 && X > 0);
        const float *w2 = raw->ex_out_w, *b2 = raw->ex_out_b;
        if (carries_ex && (!w2 || !b2)) { set_error("vsv_pack_weights: missing ex_feats_decoder"); return -1; }
        // ex path rides in features [n_out, n_out+X) of this layer: ex = W2 * ex_in + b2 (no ReLU)
        WFn W = [&, n_in, n_out, dense, wd](int r, int c) {
            if (r < n_out) return c < n_in ? (dense ? wd[(size_t)r * n_in + c] : M[(size_t)r * n_in + c]) : 0.0f;
            return c >= n_in ? w2[(size_t)(r - n_out) * X + (c - n_in)] : 0.0f;
        };
        BFn B = [=](int r) { return r < n_out ? bias[dense ? r : r / 25] : b2[r - n_out]; };
        const int x_in = (i == 0) ? X : 0, x_out = carries_ex ? X : 0;
        pack_sec(out, dc.head[hi++], head_out(n_out, x_out, cout), head_in(n_in, x_in, cin), W, B);
        if (i == 0) {
            // inference: this layer applied to the Linear above, as one matrix (float64 accumulation)
            const int n_f = n_out + x_out;
            std::vector<float> F((size_t)n_f * Hd), bF(n_f);
            for (int r = 0; r < n_f; ++r) {
                double bacc = B(r);
                std::vector<double> row(Hd, 0.0);
                for (int k = 0; k < n_lin; ++k) {
                    const double a = W(r, k);
                    if (a == 0.0) continue;
                    bacc += a * B_lin(k);
                    for (int c = 0; c < Hd; ++c) row[c] += a * W_lin(k, c);
                }
                bF[r] = (float)bacc;
                for (int c = 0; c < Hd; ++c) F[(size_t)r * Hd + c] = (float)row[c];
            }
            WFn Wf = [&](int r, int c) { return F[(size_t)r * Hd + c]; };
            BFn Bf = [&](int r) { return bF[r]; };
            pack_sec(out, dc.head_fold, head_out(n_out, x_out, cout), lay.NI > 0 ? in_perm(Hd) : in_nat(Hd), Wf, Bf);
        }
        cin = cout;
    }
    {
        std::vector<float> M;
        if (!lay.dense) M = conv_matrix(raw->dec_out_w, cin, lay.Q, false);
        const float* wo = raw->dec_out_w;                        // dense: nn.Linear (Q*25, 25), :407
        const int n_in = cin * per;
        const float* bias = raw->dec_out_b;
        const bool dense = lay.dense;
        WFn W = [&, n_in, dense, wo](int r, int c) { return dense ? wo[(size_t)r * n_in + c] : M[(size_t)r * n_in + c]; };
        BFn B = [=](int r) { return bias[dense ? r : r / 25]; };
        pack_sec(out, dc.head[hi++], out_nat32(lay.Q * 25), head_in(n_in, 0, cin), W, B);
        for (int c = 0; c < lay.Q; ++c) {
            OutMap one = [=](int t, int i) { return (t == 0 && i < 25) ? c * 25 + i : -1; };
            pack_sec(out, dc.head_ch[c], one, head_in(n_in, 0, cin), W, B);
        }
    }
    // the bands are structural: nothing outside them may be non-zero
    for (int li = 0; li < dc.n_head; ++li) {
        const Sec& sc = dc.head[li];
        for (int q = 0; q < sc.nq; ++q)
            for (int t = 0; t < sc.nt && t < 4; ++t) {
                if (q >= sc.qlo[t] && q < sc.qhi[t]) continue;
                const float* w = out + sc.w_off + ((size_t)q * sc.nt + t) * 256;
                for (int e = 0; e < 256; ++e)
                    if (w[e] != 0.0f) { set_error("vsv_pack_weights: weight outside the band of head layer %d", li); return -3; }
            }
    }
    return 0;
}

// Diagnostic: dump the blob layout as int64 records of 6 (w_off, b_off, nq, nt, n_relu, n_out).
// Order: for which in {enc, ctx}: chain[0..7], lstm[0..1], head; then dec: lstm[0..1], head[0..7];
// then scalars: enc.kx, ctx.kx, enc.nh, ctx.nh, enc.E, enc.ex_f0, dec.kx, dec.n_head, dec.ex_layer,
// dec.ex_f0, NTm, NTd, Hd, enc.n_chain, ctx.n_chain, NI.  Returns the number of int64 written (or needed when out is NULL).
extern "C" int vsv_debug_layout(const vsv_dims* dims, long long* out, int cap) {
    Layout lay;
    const char* msg = nullptr;
    if (!dims || make_layout(*dims, lay, &msg) != 0) {
        set_error("vsv_debug_layout: %s", msg ? msg : "null dims");
        return -1;
    }
    std::vector<long long> v;
    auto put = [&](const Sec& s) {
        v.push_back((long long)s.w_off); v.push_back((long long)s.b_off); v.push_back(s.nq);
        v.push_back(s.nt); v.push_back(s.n_relu); v.push_back(s.n_out);
    };
    for (int w = 0; w < 2; ++w) {
        for (int i = 0; i < kMaxDenseLayers; ++i) put(lay.enc[w].conv[i]);
        put(lay.enc[w].lstm[0]); put(lay.enc[w].lstm[1]); put(lay.enc[w].head);
    }
    put(lay.dec.lstm[0]); put(lay.dec.lstm[1]);
    for (int i = 0; i < kMaxDenseLayers; ++i) put(lay.dec.head[i]);
    const long long sc[] = {lay.enc[0].kx, lay.enc[1].kx, lay.enc[0].nh, lay.enc[1].nh, lay.enc[0].E,
                            lay.enc[0].ex_f0, lay.dec.kx, lay.dec.n_head, lay.dec.ex_layer, lay.dec.ex_f0,
                            lay.NTm, lay.NTd, lay.Hd, lay.enc[0].n_chain, lay.enc[1].n_chain, lay.NI};
    for (long long x : sc) v.push_back(x);
    put(lay.dec.head_fold);
    v.push_back(lay.dec.fold_at);
    for (int c = 0; c < lay.Q; ++c) put(lay.dec.head_ch[c]);
    if (out) {
        if (cap < (int)v.size()) { set_error("vsv_debug_layout: buffer too small"); return -1; }
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    }
    return (int)v.size();
}
