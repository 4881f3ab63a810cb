// Layout of the packed weight blob shared by the host packer (vsv_pack.cpp)
// and the kernel launchers (vsv_kernels.hip).
//
// Every layer is stored as MFMA A-operand fragments for v_mfma_f32_16x16x4_f32:
//   weights  [nq][nt][64 lanes][4]  : lane l, element j of quad q, tile t holds
//                                     W[out(t, l&15)][in(4q+j, l>>4)]
//   bias     [nt][64 lanes][4]      : lane l, element r holds bias[out(t, 4*(l>>4)+r)]
// so that D = A*B lands as D[row 4g+r of tile t][batch m] in lane (g = l>>4,
// m = l&15), register r.  Because the B operand of the NEXT layer wants
// B[k = 4s+g][m] in the same lane, a layer's output registers are the next
// layer's B operands with no lane movement when the next layer orders its
// k-steps as s = 4t + r  <->  feature 16t + 4g + r  ("PERM" input order below).
#pragma once
#include <cstddef>
#include "../../include/vsv_hip.h"

namespace vsv {

constexpr int kMaxDenseLayers = 6;
constexpr int kMaxFeat = 128;      // widest activation vector (8 tiles of 16)

struct Sec {                       // one packed layer
    size_t w_off = 0, b_off = 0;   // offsets in floats into the blob
    int nq = 0;                    // k quads (4 k-steps = 16 input features each)
    int nt = 0;                    // output tiles (16 output rows each)
    int n_relu = 0;                // output features < n_relu get ReLU (dense layers)
    int n_out = 0;                 // real output features
};

struct EncLayout {
    int n_conv = 0;
    Sec conv[VSV_MAX_CONV];
    Sec lstm[2];
    Sec head;                      // [z_mean; z_log_var] (which=0) or final_compression (which=1)
    int nh = 0;                    // head outputs: 2L or L
    int E = 0;                     // LSTM input features: c_last*25 + X
    int kx = 0;                    // k-steps of the LSTM x part (PERM order): 4*ceil(E/16)
    int ex_f0 = 0;                 // first ex-feature slot in the embedding (c_last*25)
};

struct DecLayout {
    Sec lstm[2];
    int kx = 0;                    // ceil(2L/4), natural order
    int n_head = 0;
    Sec head[kMaxDenseLayers];
    int ex_layer = -1;             // head layer whose features [ex_f0, ex_f0+X) are the ex output
    int ex_f0 = 0;
    int n_surf = 0;                // Q*25
};

struct Layout {
    int L = 0, X = 0, Hm = 0, NL = 0, Q = 0;
    int Hd = 0;                    // decoder LSTM hidden = c_last*25 + X
    int NTm = 0, NTd = 0;          // hidden tiles (4 units each) of encoder / decoder LSTMs
    EncLayout enc[2];              // 0 = encoder, 1 = ctx_encoder
    DecLayout dec;
    size_t total = 0;              // floats
};

// Returns 0 or a negative error code; msg (optional) receives the reason.
int make_layout(const vsv_dims& d, Layout& lay, const char** msg);

void set_error(const char* fmt, ...);

}  // namespace vsv
