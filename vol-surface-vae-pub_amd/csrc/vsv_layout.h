// Layout of the packed weight blob shared by the host packer (vsv_pack.cpp)
// and the kernel launchers (vsv_kernels.hip).
//
// Every layer is stored as MFMA A-operand fragments for v_mfma_f32_32x32x2_f32
// (D[32 x 32] += A[32 x 2] . B[2 x 32]; A = weights, B = a tile of 32 batch rows):
//   weights  [nq][nt][64 lanes][4] : lane l, element j of quad q, tile t holds
//                                    W[out(t, l & 31)][in(4q + j, l >> 5)]
//            (k-step s = 4q + j covers two input features, one per lane half)
//   bias     [nt][2][16]           : (t, hh, reg) holds bias[out(t, 8*(reg>>2) + 4*hh + (reg&3))]
// so that D = A.B lands as D[row 8*(reg>>2) + 4*hh + (reg&3) of tile t][batch m]
// in lane (hh = l >> 5, m = l & 31), accumulator register reg.  The B operand of
// the NEXT layer wants B[k-step s][half hh][m] in the same lane, so a layer's
// output registers are the next layer's B operands with no lane movement when
// the next layer orders its k-steps as s = 16t + reg  <->  feature
// 32t + 8*(reg>>2) + 4*hh + (reg&3)  ("PERM32" input order below).
// LSTM layers put (i,f,g,o) of one hidden unit in the 4 consecutive D rows a lane
// holds: tile row i = 4*(unit & 7) + gate, so lane (hh, m) owns units
// 8t + 2jj + hh (jj = reg >> 2) and its new h lands in natural k-step order
// (k-step 4t + jj, half hh  <->  unit 2*(4t + jj) + hh).
#pragma once
#include <cstddef>
#include "../../include/vsv_hip.h"

namespace vsv {

constexpr int kMaxDenseLayers = 8;
constexpr int kMaxFeat = 128;      // widest activation vector (4 tiles of 32)
constexpr int kMaxChannels = 5;  // n_quantiles * 25 <= kMaxFeat
constexpr int kHeadTiles = 4;      // decoder-head layers (all but the last) are padded to 4 tiles
constexpr int kDenseEx0 = 96;      // use_dense_surface: first ex-feature slot of a layer that carries them (its last tile)

struct Sec {                       // one packed layer
    size_t w_off = 0, b_off = 0;   // offsets in floats into the blob
    int nq = 0;                    // k quads (4 k-steps = 8 input features each)
    int nt = 0;                    // output tiles (32 output rows each)
    int n_relu = 0;                // output features < n_relu get ReLU (dense layers)
    int n_out = 0;                 // real output features
    // K quads [qlo, qhi) that can be non-zero for output tile t (conv layers of the decoder head in
    // pixel-major feature order: a tile's outputs read a band of input pixels); [0, nq) otherwise
    int qlo[4] = {0, 0, 0, 0}, qhi[4] = {0, 0, 0, 0};
};

struct EncLayout {
    int n_conv = 0;                // conv layers
    int n_chain = 0;               // conv layers + interaction Linears: the dense chain in front of the LSTM
    Sec conv[kMaxDenseLayers];     // the chain; the ex-features are injected into the output of conv[n_conv-1]
    Sec lstm[2];                   // k-space: layer 0 = [x: kx | h0: 4*NTm], layer 1 = [h0 | h1]
    Sec head;                      // [z_mean; z_log_var] (which=0) or final_compression / identity (which=1)
    int nh = 0;                    // head outputs: 2L or Lc
    int E = 0;                     // LSTM input features: c_last*25 + X (dense surfaces: c_last + X)
    int kx = 0;                    // k-steps of the LSTM x part (PERM32 order): 16 * tiles of the chain's last layer
    int ex_f0 = 0;                 // first ex-feature slot in the embedding (c_last*25; dense surfaces: kDenseEx0)
};

struct DecLayout {
    Sec lstm[2];                   // k-space: layer 0 = [x: kx | h0: 4*NTd], layer 1 = [h0 | h1]
    int kx = 0;                    // k-steps of [ctx | z] in natural order: ceil((Lc + L)/2) rounded up to 4
    int n_head = 0;
    Sec head[kMaxDenseLayers];     // [interaction Linears,] Linear, ConvT x n_conv, Conv
    // inference only: head[fold_at] (Linear) and head[fold_at + 1] (first ConvT [+ ex Linear]) have no
    // activation between them (reference :495-503), so generation runs their product as ONE layer
    Sec head_fold;
    int fold_at = -1;
    // inference only: the output Conv restricted to ONE decoder channel (25 outputs in one tile), for
    // callers that reduce a single channel of the Monte-Carlo draw (vsv_decode, VSV_OUT_CHANNEL)
    Sec head_ch[kMaxChannels];
    int ex_layer = -1;             // head layer whose features [ex_f0, ex_f0+X) are the ex output
    int ex_f0 = 0;
    int n_surf = 0;                // Q*25
};

struct Layout {
    int L = 0, X = 0, Hm = 0, NL = 0, Q = 0;
    int Lc = 0;                    // context-embedding width: L (compress_context) or Hm
    int NI = 0;                    // interaction Linears per module: 0, or interaction_layers + 1
    bool dense = false;            // use_dense_surface: Linear stacks in place of the conv / conv-transpose stacks
    int gru = 0;                   // memory cell: 0 nn.LSTM, 1 nn.GRU (gate slots r, z, n_x, n_h), 2 nn.RNN (slot 0)
    int Hd = 0;                    // decoder LSTM hidden = c_last*25 + X
    int NTm = 0, NTd = 0;          // gate tiles (8 hidden units each) of encoder / decoder LSTMs
    EncLayout enc[2];              // 0 = encoder, 1 = ctx_encoder
    DecLayout dec;
    size_t total = 0;              // floats
};

// Returns 0 or a negative error code; msg (optional) receives the reason.
int make_layout(const vsv_dims& d, Layout& lay, const char** msg);

void set_error(const char* fmt, ...);

}  // namespace vsv
