/*
 * vsv_hip.h -- C ABI of libvsv_hip.so: the CVAEMemRand hot path on MI355X (gfx950).
 *
 * The reference (0xC000005/vol-surface-vae-pub) is pure PyTorch and has no FFI:
 * its "operator boundary" for this path is the Python class
 * vae/cvae_with_mem_randomized.py:CVAEMemRand and the torch.nn ops it calls.
 * Each entry point below therefore cites the reference Python call site whose
 * arithmetic it replaces.  All citations are file:line into the reference.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types; every data pointer is a
 *     DEVICE pointer to float32 unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*;
 *   - return value: 0 ok, <0 argument error (vsv_last_error() has the text),
 *     >0 a hipError_t from a launch or a runtime call;
 *   - no entry point allocates or frees DEVICE memory or synchronises the device;
 *     scratch space comes in through `ws` (size from the matching *_ws_bytes
 *     query) and calls on different streams do not interact -- with these
 *     exceptions, each repeated at its declaration:
 *       . vsv_context_prefix creates and destroys HIP events (host objects) and
 *         issues work onto the worker streams it is given; it joins them back
 *         into `stream` on every return path;
 *       . vsv_encode_windows_noalloc uses a device array built into the library,
 *         one per GPU, shared by all calls on that GPU (overlapping calls race);
 *   - state the library keeps between calls, all of it host-side:
 *       . per THREAD: the text behind vsv_last_error() and the two events armed
 *         by vsv_profile_next_decode (consumed by that thread's next vsv_decode);
 *       . per DEVICE: the address of the built-in scratch array, looked up once
 *         (vsv_init_device or the first vsv_encode_windows_noalloc on that device,
 *         under a process-wide mutex);
 *       . per PROCESS: the environment switches (VSV_DSEQ, VSV_ESEQ, VSV_*_WG ...:
 *         kernel-choice overrides for A/B timing, INTEGRATION.md lists them) are
 *         read once, at the first call that consults them;
 *     nothing else: no handles, no caches, no weights are retained;
 *   - rows are independent sequences (a training sample, a day's context window,
 *     or one (day, MC sample) pair).
 */
#ifndef VSV_HIP_H
#define VSV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSV_ABI_VERSION 23
#define VSV_MAX_CONV 4
#define VSV_MAX_CHAIN 8
#define VSV_MAX_INTERACT 3

/* Model shape: the subset of the reference's config dict that fixes tensor
 * shapes (cvae_with_mem_randomized.py:969-1020; state_dict layout SURVEY.md
 * appendix A): conv (or, use_dense_surface, Linear) surface embedding on a 5x5 grid, LSTM / GRU / RNN
 * memory, compress_context on or off, ex_feats_hidden=None;
 * interaction_layers 0..2 (the Linear+ReLU "interaction" stacks of :131-142, :283-294, :452-463,
 * turned on by the v3 conditional-prior trainer). */
typedef struct vsv_dims {
    int latent_dim;                    /* L: config["latent_dim"]            */
    int ex_feats_dim;                  /* X: config["ex_feats_dim"] (0 or >0) */
    int mem_hidden;                    /* config["mem_hidden"]               */
    int mem_layers;                    /* config["mem_layers"]: 1 or 2       */
    int n_conv;                        /* len(config["surface_hidden"])      */
    int surface_hidden[VSV_MAX_CONV];  /* conv channels, encoder + decoder   */
    int n_ctx_conv;                    /* len(config["ctx_surface_hidden"])  */
    int ctx_surface_hidden[VSV_MAX_CONV];
    int n_quantiles;                   /* Q = len(config["quantiles"])       */
    int interaction_layers;            /* config["interaction_layers"] or 0: n Linear+ReLU + 1 Linear, width kept */
    int mem_type;                      /* config["mem_type"]: 0 "lstm" (nn.LSTM), 1 "gru" (nn.GRU), 2 anything else (nn.RNN, tanh), :154-159 */
    int compress_context;              /* config["compress_context"]: 1 = the context encoder ends in Linear(mem_hidden
                                          -> latent_dim) (:363-366); 0 = its embedding is the memory's output itself
                                          (mem_hidden wide: the class default, :1014-1015) and the decoder memory reads
                                          [ctx (mem_hidden) | z (latent_dim)] */
    int use_dense_surface;             /* config["use_dense_surface"]: 1 = the surface embeddings are Flatten + Linear+ReLU
                                          stacks (:84-93, :237-246) and the decoder head Linear stacks (:394-407);
                                          surface_hidden / ctx_surface_hidden then hold layer WIDTHS, the embedding
                                          is surface_hidden[-1] wide and the decoder memory surface_hidden[-1] +
                                          ex_feats_dim (:375-380).  The raw-weight slots *_conv_w / dec_deconv_w /
                                          dec_out_w then take the nn.Linear weights (out, in) of enc_dense_i /
                                          ctx_enc_dense_i / dec_dense_i, dec_final / dec_output */
} vsv_dims;

/* Raw parameters in PyTorch state_dict layout (HOST pointers, float32,
 * contiguous).  Names follow SURVEY.md appendix A.  With mem_type 1 the memory
 * tensors have nn.GRU's shapes (3*hidden rows: r, z, n). */
typedef struct vsv_lstm_raw {
    const float *w_ih[2], *w_hh[2], *b_ih[2], *b_hh[2];   /* per layer */
} vsv_lstm_raw;

typedef struct vsv_raw_weights {
    /* encoder.* */
    const float *enc_conv_w[VSV_MAX_CONV], *enc_conv_b[VSV_MAX_CONV];
    vsv_lstm_raw enc_mem;
    const float *z_mean_w, *z_mean_b, *z_log_var_w, *z_log_var_b;
    /* ctx_encoder.* */
    const float *ctx_conv_w[VSV_MAX_CONV], *ctx_conv_b[VSV_MAX_CONV];
    vsv_lstm_raw ctx_mem;
    const float *ctx_comp_w, *ctx_comp_b;                 /* final_compression */
    /* decoder.* */
    vsv_lstm_raw dec_mem;
    const float *dec_in_w, *dec_in_b;                     /* surface_decoder_input */
    const float *dec_deconv_w[VSV_MAX_CONV], *dec_deconv_b[VSV_MAX_CONV]; /* dec_deconv_i..., dec_final last */
    const float *dec_out_w, *dec_out_b;                   /* dec_output (Conv2d) */
    const float *ex_in_w, *ex_in_b, *ex_out_w, *ex_out_b; /* ex_feats_decoder_input / ex_feats_decoder (X>0) */
    /* {enc,ctx,dec}_interact_linear_0.. and _final (interaction_layers + 1 Linears each, width x width) */
    const float *enc_int_w[VSV_MAX_INTERACT], *enc_int_b[VSV_MAX_INTERACT];
    const float *ctx_int_w[VSV_MAX_INTERACT], *ctx_int_b[VSV_MAX_INTERACT];
    const float *dec_int_w[VSV_MAX_INTERACT], *dec_int_b[VSV_MAX_INTERACT];
} vsv_raw_weights;

int vsv_abi_version(void);
/* Text of the last argument error raised on the calling thread. */
const char *vsv_last_error(void);

/* ---- weights -------------------------------------------------------------
 * The kernels read weights as ready-made A operands of v_mfma_f32_32x32x2_f32: a
 * layer with K inputs and N outputs is stored as [k-quad = ceil(K/8)][tile = ceil(N/32)][64 lanes][4]
 * floats (1 KiB per (k-quad, tile) block: one float4 per lane feeds the four k-pair MFMAs of
 * the quad; lane l holds output row 32*tile + l%32, k = 8*quad + 2*j + l/32 in float j), biases
 * behind the fragments; conv layers are expanded to their dense 5x5 Toeplitz matrices and
 * only the k-quads inside each tile's band are multiplied.  vsv_pack_weights converts a
 * state_dict once on the host; the caller uploads the blob and passes its device
 * address as `packed` to every compute call.  Replaces: nn.Module parameter
 * storage read by nn.Conv2d/nn.ConvTranspose2d/nn.LSTM/nn.Linear at
 * cvae_with_mem_randomized.py:100,155,196-197,252,307,343,383,413,421,428,476. */
size_t vsv_packed_floats(const vsv_dims *dims);
int vsv_pack_weights(const vsv_dims *dims, const vsv_raw_weights *raw, float *packed_host);

/* ---- encoders --------------------------------------------------------------
 * which = 0: CVAEMemRandEncoder.forward (cvae_with_mem_randomized.py:162-197):
 *            conv embedding + LSTM + z_mean / z_log_var heads;
 *            out is (n_rows, T, 2L): [..., :L] = z_mean, [..., L:] = z_log_var.
 * which = 1: CVAECtxMemRandEncoder.forward (:314-344): out is (n_rows, T, L).
 * surface element (row, t, p) is surf[row*surf_row_stride + t*surf_t_stride + p],
 * p < 25 -- so overlapping day windows of one calendar array can be encoded in
 * place (row stride 25, t stride 25).  ex may be NULL when X == 0. */
size_t vsv_encode_ws_bytes(const vsv_dims *dims, long n_rows, int T);
int vsv_encode(const vsv_dims *dims, const float *packed, int which,
               const float *surf, long surf_row_stride, long surf_t_stride,
               const float *ex, long ex_row_stride, long ex_t_stride,
               long n_rows, int T, float *out,
               void *ws, size_t ws_bytes, void *stream);

/* vsv_encode continuing from / leaving a memory state: state_in (n_rows / state_div states, each
 * vsv_enc_state_floats floats = layers x (h | c) of mem_hidden padded to a multiple of 8, layout as the
 * decoder's in vsv_decode_in) or null for the zero state; row r starts from state r / state_div; state_out
 * (n_rows states) or null.  Used by the autoregressive rollouts (generate_autoregressive_sequence,
 * :717-807): the leading part of a row's window that is still the day's original context is encoded once
 * per day and all samples of the day continue from that state. */
size_t vsv_enc_state_floats(const vsv_dims *dims);
int vsv_encode_state(const vsv_dims *dims, const float *packed, int which,
                     const float *surf, long surf_row_stride, long surf_t_stride,
                     const float *ex, long ex_row_stride, long ex_t_stride,
                     long n_rows, int T, const float *state_in, long state_div, float *state_out,
                     float *out, void *ws, size_t ws_bytes, void *stream);

/* Autoregressive rollouts (generate_autoregressive_sequence, :717-807): a generated day's conv embedding and
 * first-layer input product do not depend on its position in the window, so they are computed once, when the
 * day is generated -- vsv_encode_pre: rows surf (n_rows x 25, row stride given) [, ex] -> pre
 * [32 * NTm gate values][pre_stride >= rows padded to 32] -- and the encoders of later steps run from the
 * tables of their T days -- vsv_encode_from_pre: day t's table at pre + t * pre_tstride; state arguments as in
 * vsv_encode_state.  Results equal vsv_encode_state on the same surfaces. */
size_t vsv_encode_pre_ws_bytes(const vsv_dims *dims, long n_rows);
int vsv_encode_pre(const vsv_dims *dims, const float *packed, int which,
                   const float *surf, long surf_row_stride, const float *ex, long ex_row_stride,
                   long n_rows, float *pre, long pre_stride, void *ws, size_t ws_bytes, void *stream);
size_t vsv_encode_from_pre_ws_bytes(const vsv_dims *dims, long n_rows, int T);
int vsv_encode_from_pre(const vsv_dims *dims, const float *packed, int which,
                        const float *pre, long pre_stride, long pre_tstride, long n_rows, int T,
                        const float *state_in, long state_div, float *state_out,
                        float *out, void *ws, size_t ws_bytes, void *stream);

/* The same encoders over OVERLAPPING windows of one calendar series -- the date-range backfill loops of
 * generate_surfaces.py:60-102 / eval_scripts/eval_utils.py:42-70, where window r is days [r, r+T) of
 * vol (n_windows + T - 1, 25) [, ex (n_windows + T - 1, X)], both contiguous.  Results equal
 * vsv_encode(surf = vol, row stride 25, t stride 25, ...); the conv embedding and the first memory
 * layer's input product depend only on the calendar day and are computed once per day instead of once
 * per (window, step).  At most 8192 windows per call (-5 beyond: use vsv_encode). */
size_t vsv_encode_windows_ws_bytes(const vsv_dims *dims, long n_windows, int T);
int vsv_encode_windows(const vsv_dims *dims, const float *packed, int which, const float *vol, const float *ex,
                       long n_windows, int T, float *out, void *ws, size_t ws_bytes, void *stream);
/* ... continuing from / leaving one memory state per window (vsv_enc_state_floats each; null = zero state /
 * not wanted): a window's T steps can be run in segments (segment starting at step t0: vol + 25*t0,
 * ex + X*t0), so that the decoder's context prefix of one segment overlaps the encoders' next segment. */
int vsv_encode_windows_state(const vsv_dims *dims, const float *packed, int which, const float *vol, const float *ex,
                             long n_windows, int T, const float *state_in, float *state_out,
                             float *out, void *ws, size_t ws_bytes, void *stream);

/* The date-range prefix as a pipeline of LAYER stages (two-layer memories; -4 otherwise).  The same computation as
 * vsv_encode_windows_state, resp. as a prefix-only vsv_decode (t_out0 = T) on (ctx_t, z_mean_t), split at the
 * layer boundary -- still the forwards of cvae_with_mem_randomized.py:162-197 / :314-344 / :482-490 inside the
 * date loops of generate_surfaces.py:60-102 --, bit-equal results:
 *   *_layer0: [conv embedding and layer-0 table per calendar day,] layer 0's recurrence; leaves h0 of every
 *             step in ws and writes layer 0's slot of state_out [and of in->gate_out];
 *   *_layer1: b + W_ih(1) . h0 for all steps at once, layer 1's recurrence [, the latent / compression head
 *             into out]; writes layer 1's slot of state_out [and of in->gate_out].
 * A recurrent launch holds one CU per 32-row tile for its T sequential steps; with the two calls of a time
 * segment on DIFFERENT streams (layer1 ordered after layer0 by an event, same ws) layer 1 of segment i runs
 * beside layer 0 of segment i+1, so the critical path is one layer's steps plus a segment of the other.
 * state_in / state_out: as for vsv_encode_windows_state / vsv_decode (all layers per row; each call touches its
 * own layer's slot).  vsv_decode_prefix_layer0 reads in->ctx / zb (ctx_T = z_T = T, lv = NULL) and
 * in->state_in; _layer1 reads in->state_in and in->gate_out only. */
size_t vsv_encode_windows_split_ws_bytes(const vsv_dims *dims, long n_windows, int T);
int vsv_encode_windows_layer0(const vsv_dims *dims, const float *packed, int which, const float *vol, const float *ex,
                              long n_windows, int T, const float *state_in, float *state_out,
                              void *ws, size_t ws_bytes, void *stream);
int vsv_encode_windows_layer1(const vsv_dims *dims, const float *packed, int which, long n_windows, int T,
                              const float *state_in, float *state_out, float *out,
                              void *ws, size_t ws_bytes, void *stream);
size_t vsv_decode_prefix_split_ws_bytes(const vsv_dims *dims, long n_rows, int T);
struct vsv_decode_in;
int vsv_decode_prefix_layer0(const vsv_dims *dims, const float *packed, const struct vsv_decode_in *in, long n_rows,
                             int T, float *state_out, void *ws, size_t ws_bytes, void *stream);
int vsv_decode_prefix_layer1(const vsv_dims *dims, const float *packed, const struct vsv_decode_in *in, long n_rows,
                             int T, float *state_out, void *ws, size_t ws_bytes, void *stream);

/* The whole date-range prefix in one call: for the windows vol[d : d + ctx_len] (d < n_days) the decoder state
 * after the context steps (state_out, (n_days, vsv_state_floats)) and the first future step's shared gates
 * (gate_out, (n_days, vsv_gate_floats)) -- posterior mean and context embedding from the two encoders, decoder
 * over (ctx_t, z_mean_t): the context half of get_surface_given_conditions (:695-715) for every day of the loops
 * of generate_surfaces.py:60-102.  The ctx_len steps are cut into n_seg time segments and every stack into its
 * two layers (the *_layer0 / *_layer1 stages above); all stage launches are issued here onto `streams`
 * (n_streams = 6: one per stage; 4: the two encoders' same-layer stages share a stream -- HIP's default number of
 * hardware queues) and ordered by HIP events; the work is forked from `stream` and joined back into it.
 * Creates and destroys HIP events (host objects); no device memory is allocated.  Results are bit-equal to the
 * fused launches for any n_seg.  Two-layer memories, at most 8192 days per call.
 * Beyond 2048 days (where the stages run on 16-row tiles) the encoders' heads and the decoder's input assembly are
 * not launches of their own: the heads ride inside the encoders' layer-1 recurrent launches and the decoder's
 * layer-0 stage reads their outputs where those launches leave them (same numbers bit for bit; VSV_PREFIX_FUSE=0
 * keeps the separate launches). */
size_t vsv_context_prefix_ws_bytes(const vsv_dims *dims, long n_days, int ctx_len, int n_seg);
int vsv_context_prefix(const vsv_dims *dims, const float *packed, const float *vol, const float *ex, long n_days,
                       int ctx_len, int n_seg, void *const *streams, int n_streams,
                       float *state_out, float *gate_out, void *ws, size_t ws_bytes, void *stream);

/* The one entry point that allocates nothing and needs no caller workspace.  NO COUNTERPART IN THE REFERENCE:
 * it exists for callers that cannot allocate device memory -- code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  Same computation and results as vsv_encode_windows (bit for bit);
 * the workspace is a device array built into the library, VSV_NOALLOC_SCRATCH_BYTES per GPU, held from the
 * library's first use on that GPU until the process ends.  Requests whose vsv_encode_windows_ws_bytes exceed
 * it are refused with -6 (e.g. 512 windows x 60 steps fit, 4000 x 60 do not).  The call makes no hipMalloc /
 * hipFree and calls no allocating routine; its first call on a device looks the array's address up
 * (hipGetSymbolAddress).  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two
 * streams, two threads) race -- serialise them.
 * UNDER HIP GRAPH CAPTURE: the address lookup loads the library's code object and must not happen inside a
 * capture -- call vsv_init_device() (or make one eager call of this entry point) on the device first; after that
 * the call only enqueues kernels.  vsv_init_device: resolves the current device's copy of the scratch array,
 * 0 or a hipError_t; idempotent, thread-safe, no allocation. */
#define VSV_NOALLOC_SCRATCH_BYTES (16u << 20)
int vsv_init_device(void);
int vsv_encode_windows_noalloc(const vsv_dims *dims, const float *packed, int which, const float *vol, const float *ex,
                               long n_windows, int T, float *out, void *stream);

/* ---- decoder ---------------------------------------------------------------
 * CVAEMemRandDecoder.forward (cvae_with_mem_randomized.py:482-514) over T steps
 * for n_rows independent rows, with the LSTM input assembled on the fly
 * (the torch.cat at :707 / :1066) from:
 *   ctx part  f <  L: ctx[(row/ctx_div)*ctx_rs + t*L + f]        for t < ctx_T, else 0
 *                     (the zero padding of :704-705 / :1061-1062);
 *   z   part  f >= L: base + scale * eps, where for t < z_T
 *                       base  = zb[(row/z_div)*z_rs + t*L + f]
 *                       scale = lv ? exp(0.5*lv[same index]) : 0   (reparameterisation, :199)
 *                     and for t >= z_T   base = mu0, scale = std0 (prior N(mu,std), :693-694);
 *                     eps = eps_in[(row*eps_T + (t-eps_t0))*L + f] when eps_in != NULL,
 *                     otherwise Philox4x32-10 + Box-Muller keyed by
 *                     (seed; sample = row % S, day = day0 + row / S, step = step0 + t - eps_t0, f).
 * The LSTM starts from state_in[(row/state_div)] (layout (n, layers, 2, HP) with
 * HP = vsv_state_floats(dims)/(2*layers): h then c, unit-major) or from zero when NULL,
 * and writes its final state to state_out (same layout, one per row) when not NULL.
 * For steps t >= t_out0 the head (Linear -> ConvT+ReLU x3 -> Conv [, ex Linear x2],
 * :494-510) is evaluated and written to
 *   out_mode 0: surf_out (n_rows, T-t_out0, Q, 5, 5), the reference layout;
 *   out_mode 1: surf_out (n_rows/S, T-t_out0, Q*25, S), MC-sample-minor, the layout
 *               vsv_percentile reads (requires n_rows % S == 0);
 *   out_mode VSV_OUT_CHANNEL(c), 0 <= c < Q: surf_out (n_rows/S, T-t_out0, 25, S), sample-minor,
 *               decoder channel c alone: the rows of the output Conv that produce the other channels
 *               are neither multiplied nor stored (the MC products of generate_surfaces.py:60-102
 *               reduce the p50 channel only);
 * and ex_out (n_rows, T-t_out0, X) when X > 0 and ex_out != NULL.
 * z_out, when not NULL, receives the z part that was fed in: (n_rows, T, L).
 * gate_out, when not NULL, receives per row bias + W_hh . h_final of every layer (the
 * part of the next step's gate pre-activations that depends only on the final state);
 * a later call that continues MANY rows from that state (state_in + state_div, the MC
 * samples of one day) passes it back as gate_in and its first step skips the
 * recurrent matrix products -- same numbers up to float summation order.
 * Pass t_out0 = T to run the LSTM only (context prefix: state_out is the product). */
typedef struct vsv_decode_in {
    const float *ctx;   long ctx_rs;  int ctx_T;  long ctx_div;
    const float *zb;    const float *lv;  long z_rs;  int z_T;  long z_div;
    float mu0, std0;
    const float *eps_in; int eps_T; int eps_t0;
    uint64_t seed; long S; long day0; int step0;
    const float *state_in; long state_div;
    const float *gate_in;   /* optional, with state_in: see below */
    float *gate_out;        /* optional: (n_rows, vsv_gate_floats) */
} vsv_decode_in;

size_t vsv_state_floats(const vsv_dims *dims);   /* floats of decoder LSTM state per row */
size_t vsv_gate_floats(const vsv_dims *dims);    /* floats of gate_in / gate_out per row */
size_t vsv_decode_ws_bytes(const vsv_dims *dims, long n_rows, int T, int t_out0);
#define VSV_OUT_CHANNEL(c) (2 + (c))
int vsv_decode(const vsv_dims *dims, const float *packed, const vsv_decode_in *in,
               long n_rows, int T, int t_out0,
               float *state_out, float *z_out,
               float *surf_out, int out_mode, float *ex_out,
               void *ws, size_t ws_bytes, void *stream);

/* ---- MC reduction ------------------------------------------------------------
 * np.percentile(x, q, axis=samples) with linear interpolation -- the MC->quantile
 * convention of experiments/bootstrap_baseline/generate_bootstrap_autoregressive.py:194-199.
 * in: sample-minor vectors of S floats; vector v = (group g = v / inner, i = v % inner) starts at
 * in + (g*group_stride + i)*S, so group_stride = inner is a dense (n_vec, S) array and
 * group_stride = Q*25 with in offset by channel*25*S reduces ONE decoder channel of the
 * (days, steps, Q*25, S) tensor vsv_decode writes, in place.
 * out[(v / inner)*nq*inner + k*inner + v % inner] is the k-th percentile of vector v
 * (inner = 25 gives (..., nq, 5, 5) per (day, step)).  q_host: nq percentiles in [0,100] (host).
 * S <= 1024: one wavefront sorts a vector in registers (bitonic network, cross-lane exchanges);
 * S <= 16384: one workgroup per vector, sorted in LDS; S > 16384 is refused (-1): the reference's MC
 * products use 1000 samples per day.  A NaN sample is ordered arbitrarily (np.percentile would return NaN). */
int vsv_percentile(const float *in, long n_vec, int S, const double *q_host, int nq,
                   int inner, long group_stride, float *out, void *stream);

/* ---- noise ---------------------------------------------------------------------
 * The eps the decoder draws in Philox mode, materialised: out (n_days*S, n_steps, L).
 * Stands in for torch.randn at cvae_with_mem_randomized.py:693-694. */
int vsv_philox_normal(uint64_t seed, long day0, long n_days, long S, int step0, int n_steps,
                      int L, float *out, void *stream);

/* Latents from the Gaussian mixture fitted to the aggregated posterior -- CVAEMemRand._sample_from_fitted_prior
 * (cvae_with_mem_randomized.py:605-640; parameters as written by analysis_code/fit_aggregated_posterior.py:139-155):
 * component c ~ multinomial(weights), z = means[c] + chol[c] . eps with chol the lower Cholesky factor of the
 * component's covariance.  means (K, L), chol (K, L, L), cdf (K) = cumulative weights, all float32 on the device;
 * z_out (n, L); comp_out (n) int32 or NULL.  comp_in (n, int32) / eps_in (n, L) inject the draws (parity with the
 * reference's torch.multinomial / torch.randn); NULL draws them from Philox4x32-10 keyed by (seed; sample index).
 * L <= 16. */
int vsv_gmm_sample(const float *means, const float *chol, const float *cdf, int K, int L, long n,
                   const int *comp_in, const float *eps_in, uint64_t seed, float *z_out, int *comp_out, void *stream);

/* ---- training ---------------------------------------------------------------------
 * CVAEMemRand.train_step / train_step_multihorizon / test_step
 * (cvae_with_mem_randomized.py:1076-1308) split into: forward passes that keep the
 * activations BPTT needs, the loss terms with their gradients, and backward kernels that
 * turn output gradients into gate / pre-activation gradients.  Weight gradients are the
 * plain matrix products  dW = G^T . X  of tensors these calls leave in memory (the caller
 * runs them as library GEMMs and scatters them onto the parameters through the packing
 * map); the recurrent and element-wise work is here.
 *
 * Tile layouts.  rt = ceil(n_rows/32) row tiles; an instance is (row tile, t),
 * inst = tile*T + t.  A "tile-B" tensor is [inst][slots][64]: slot s, lane (hh = l>>5,
 * m = l&31) holds one feature of row 32*tile + m -- feature 2s+hh in NATURAL order
 * (LSTM h / c, decoder x part, first conv input) or 32*(s>>4) + 8*((s>>2)&3) + 4*hh + (s&3)
 * in SLOT order (outputs of dense layers and gate rows; gate row = 32*tile' + 4*unit%8 + gate).
 * NTm/NTd = ceil(hidden/8) gate tiles.  Buffers of the vsv_train_save struct, all caller-owned:
 *   emb_acts  encoders: [n_conv][inst][64][64]  input of every conv layer
 *   xe        [inst][kx][64]                    LSTM x part (encoders: conv embedding, slot
 *                                               order, kx = 64; decoder: [ctx | z], natural)
 *   gates     [inst][layers][16*NT][64]         activated gates i,f,g,o (slot 16*tile+4*jj+gate)
 *   c, h      [inst][layers][4*NT][64]          cell / hidden state after the step (natural)
 *   hout      [tile*(T-t_out0) + t-t_out0][4*NT][64]  top-layer h of the output steps
 *   head_acts decoder: [n_head][out inst][64][64]     input of every head layer
 *   hd        [inst][4*NT][64]                  layer 0's h after dropout (drop_p > 0, 2 layers)
 * drop_p > 0 turns on nn.LSTM's inter-layer dropout (train mode, :155/:307/:476 with
 * config["mem_dropout"]): every hidden unit of layer 0's output is zeroed with probability
 * drop_p and the survivors scaled by 1/(1-drop_p) before layer 1 reads them (the recurrent
 * path keeps the undropped h).  The mask is a pure function of (drop_seed, drop_id, row, t,
 * unit) -- Philox4x32-10, vsv_dropout_mask materialises it -- where drop_id tells the three
 * LSTMs of a model apart (0 posterior encoder, 1 context encoder, 2 decoder). */
typedef struct vsv_train_save {
    float *emb_acts, *xe, *gates, *c, *h, *hout, *head_acts;
    float *hd;
    float drop_p;
    uint64_t drop_seed;
    int drop_id;
    const uint64_t* drop_seed_dev;  /* non-null: the kernels read the seed from this device word at start instead
                                     * of drop_seed, so a captured HIP graph draws fresh masks on every replay */
} vsv_train_save;

/* vsv_encode / vsv_decode (out_mode 0, zero initial state, head not fused) that also fill `sv`. */
int vsv_train_encode(const vsv_dims *dims, const float *packed, int which,
                     const float *surf, long surf_row_stride, long surf_t_stride,
                     const float *ex, long ex_row_stride, long ex_t_stride,
                     long n_rows, int T, float *out, const vsv_train_save *sv, void *stream);
/* Both encoders' vsv_train_encode in one call -- posterior encoder (which = 0) over T0 steps into out0 / sv0, context
 * encoder (which = 1) over T1 steps into out1 / sv1, same batch: CVAEMemRand.forward's two independent encoder calls
 * (cvae_with_mem_randomized.py:1046-1058).  Few row tiles (<= 16 of 32 rows): every launch of a stack goes out
 * together with its twin of the other stack as ONE launch (half the launches, both stacks side by side without a second
 * stream); otherwise the two stacks run one after the other on `stream`.  Results equal two vsv_train_encode calls,
 * except for the layout of out0: TWO planes (2, n_rows, T0, L) -- z_mean, then z_log_var -- instead of (n_rows, T0, 2L),
 * the form vsv_train_decode's zb / lv arguments take. */
int vsv_train_encode_pair(const vsv_dims *dims, const float *packed,
                          const float *surf, long surf_row_stride, long surf_t_stride,
                          const float *ex, long ex_row_stride, long ex_t_stride,
                          long n_rows, int T0, int T1, float *out0, float *out1,
                          const vsv_train_save *sv0, const vsv_train_save *sv1, void *stream);
int vsv_train_decode(const vsv_dims *dims, const float *packed, const vsv_decode_in *in,
                     long n_rows, int T, int t_out0, float *z_out, float *surf_out, float *ex_out,
                     const vsv_train_save *sv, void *stream);

/* Backward of a chain of dense layers (conv embeddings, linear heads, decoder head), one
 * layer per entry, LAST forward layer first.  Per layer: g = (incoming + add) masked by the
 * forward output's ReLU -> g_out (tile-B, slot order; dW = g^T . layer input); then
 * incoming = W^T . g through the transposed fragments wT (built by the caller from the
 * packed blob; their output order is the previous layer's slot order).  in_mode 0 needs in_f >= 1 and
 * a layer with `add` needs add_n >= 1 (error -1 otherwise: the kernels clamp feature indices into those
 * ranges instead of branching around each load). */
typedef struct vsv_dense_bwd_layer {
    const float *wT; int nqT, ntT;          /* ntT = 0: stop, no input gradient wanted */
    const float *act_out; int n_relu;       /* forward output of the layer (tile-B [inst][64][64]) or NULL */
    float *g_out;                           /* [inst][64][64] or NULL */
    const float *add; int add_f0, add_n;    /* row-major (row, T, add_n) extra gradient on output features */
} vsv_dense_bwd_layer;
typedef struct vsv_dense_bwd {
    int n_layers;
    vsv_dense_bwd_layer layer[VSV_MAX_CHAIN];
    long n_rows, n_rowtiles;
    int T;
    int in_mode;                            /* 0: row-major (row, T, in_f); 1: tile-B [inst][64][64], slot order */
    const float *in; int in_f;
    float *dx_out;                          /* [inst][64][64] gradient w.r.t. the chain input, or NULL */
} vsv_dense_bwd;
int vsv_dense_backward(const vsv_dense_bwd *d, void *stream);

/* BPTT of nn.LSTM (the autograd of cvae_with_mem_randomized.py:193, :342, :491).  wT_hh[l],
 * wT_in[l]: transposed fragments with K = the layer's gate rows (slot order, 16*NT k-steps) and
 * output = "unit tiles" (tile w, register 4*jj'+r' <-> unit 8*(4*jj'+w) + 2*r' + hh), 4 tiles;
 * wT_in[0] maps to the x features instead (nt_x tiles).  dh_top: gradient w.r.t. the top
 * layer's h at the output steps, [out inst][64][64] with slot 16*w + register.  dgates
 * receives the gate pre-activation gradients in the layout of sv->gates; the x gradient is
 * written as tile-B (dx_tile, slot order) and/or row-major (dx_rows). */
typedef struct vsv_lstm_bwd {
    const float *wT_hh[2], *wT_in[2];
    int nt_x;
    long n_rows, n_rowtiles;
    int T, t_out0;
    const float *sv_gates, *sv_c, *dh_top;
    float *dgates, *dx_tile, *dx_rows;
    int dx_f;
    float drop_p;                           /* as in vsv_train_save: masks d(layer 0 output) coming from layer 1 */
    uint64_t drop_seed;
    int drop_id;
    int gru;                                /* 1: the saved gates are nn.GRU's (r, z, n, W_hn h + b_hn); 2: nn.RNN's (h' in slot 0) */
    const uint64_t* drop_seed_dev;          /* as in vsv_train_save */
    int sv_nl, sv_l0;                       /* sv_nl > 0: sv_gates / sv_c / dgates hold sv_nl layers per instance and
                                             * this call handles `layers` of them from sv_l0 (wT_hh[0] is layer sv_l0's):
                                             * one call per layer, with vsv_tileb_gemm_launch between them */
} vsv_lstm_bwd;
int vsv_lstm_backward(const vsv_lstm_bwd *d, int hidden, int layers, void *stream);

/* The parts of BPTT off the recurrence, for all (row tile, t) instances at once:
 *   out(inst) = W^T . g(inst)      d h_{l-1}(t) = W_ih(l)^T . dgates(l, t)   (x dropout factors of layer l-1)
 *                                  dx(t)        = W_ih(0)^T . dgates(0, t)
 * wT: transposed fragments (nqT K-quads, ntT <= 4 output tiles); g: [inst][4*nqT][64] with g_stride floats
 * per instance; out_tile [inst][64][64] (slot 16*tile + register) and/or out_rows (n_rows, T, out_f).
 * With two layers vsv_lstm_backward(layer 1) -> this -> vsv_lstm_backward(layer 0, dh_top = out_tile,
 * t_out0 = 0) -> this computes what one two-layer call does, with half the work on the time loop. */
typedef struct vsv_tileb_gemm {
    const float *wT;  int nqT, ntT;
    const float *g;  long g_stride;
    long n_rows, n_rowtiles;  int T;
    float *out_tile, *out_rows;  int out_f;
    float drop_p;  uint64_t drop_seed;  const uint64_t *drop_seed_dev;  int drop_id, drop_layer;
} vsv_tileb_gemm;
int vsv_tileb_gemm_launch(const vsv_tileb_gemm *d, void *stream);

/* The three backward launches above for TWO independent stacks at once (the two encoders of a training step, whose
 * backward passes do not depend on each other: the autograd of cvae_with_mem_randomized.py:1046-1058).  Each takes the
 * two argument sets of the matching single entry point.  With few instances (a batch of 64) the pair goes out as ONE
 * launch whose blocks pick their set -- half the launches, both stacks side by side without a second stream --,
 * otherwise as two launches one after the other on `stream`.  Same results as two single calls. */
int vsv_dense_backward_pair(const vsv_dense_bwd *d0, const vsv_dense_bwd *d1, void *stream);
int vsv_lstm_backward_pair(const vsv_lstm_bwd *d0, const vsv_lstm_bwd *d1, int hidden, int layers, void *stream);
int vsv_tileb_gemm_launch_pair(const vsv_tileb_gemm *d0, const vsv_tileb_gemm *d1, void *stream);

/* Weight gradients of dense / LSTM layers (autograd's dW = dY^T . X for nn.Linear, the conv layers'
 * expanded matrices and nn.LSTM / nn.GRU, reached from loss.backward() at :1127): for each problem,
 *   gp[position of W[n][k]] = sum over instances and rows of G[inst][a][row] * X[inst][k][row]
 *   gp[position of b[n]]    = sum of G[inst][a][row]                          (bias != 0)
 * with G the gradients the backward kernels wrote in tile-B layout read as [inst][a = 2*slot+half][32 rows]
 * (n = the feature slot order puts at a), X the layer's saved input in its k order: x[0] (n_b[0] columns)
 * followed by x[1] (n_b[1] columns, may be 0); shift[j] = 1 reads x[j] of the previous time step of the
 * same row tile (instances are [tile][t], t < T; zeros at t = 0) -- the recurrent input h_{t-1}.
 * Positions follow the packed layout of a section at (w_off, b_off) with nt output tiles. */
#define VSV_MAX_WGRAD 24
typedef struct vsv_wgrad_prob {
    const float *g;  long g_stride;  int n_a;     /* n_a = 32 * (output tiles used), stride in floats per instance */
    const float *x[2];  long x_stride[2];  int n_b[2];  int shift[2];
    int n_inst, T;
    long w_off, b_off;  int nt;  int bias;
} vsv_wgrad_prob;
typedef struct vsv_wgrad {
    int n;
    float *gp;                                    /* gradient w.r.t. the packed blob (vsv_packed_floats) */
    vsv_wgrad_prob p[VSV_MAX_WGRAD];
} vsv_wgrad;
int vsv_weight_grads(const vsv_wgrad *d, void *stream);

/* Loss terms of train_step (:1113-1125) and their gradients: pinball over the quantile
 * channels (QuantileLoss, :28-45; qw = weights / sum(weights)), ex-feature MSE ("l2") or L1
 * on the first ex_n features (:1115-1118), KL over all T positions (:1123-1124).
 * sums (3 doubles, zeroed by the caller) receive re_surface, re_ex_feats, kl; the gradients
 * are those of  grad_scale * (re_surface + ex_weight*re_ex_feats + kl_weight*kl).
 * Hs > H: pred / ex_pred / d_pred / d_ex hold Hs steps per row and the loss is over the last H of
 * them (gradient zero on the others) -- train_step_multihorizon's horizons (:1204-1249) share one
 * decoder launch that evaluates the head on the longest horizon's steps, and grad_scale is the
 * horizon's weight.  Hs = 0 means H, grad_scale = 0 means 1. */
typedef struct vsv_loss {
    const float *pred, *target;             /* (B,Hs,Q,25); surface (B,T,25), last H steps are the targets */
    int B, H, T, Q;
    float q[8], qw[8];
    const float *ex_pred, *ex_target;       /* (B,H,X) or NULL; (B,T,X) */
    int X, ex_n, ex_l2;
    float ex_weight, kl_weight;
    const float *zm, *zl; int L;            /* (B,T,L) */
    float *d_pred, *d_ex, *d_zm, *d_zl;
    double *sums;
    int Hs;  float grad_scale;
    int surf_l2;                            /* 1: re_surface = F.mse_loss (legacy one-channel CVAEMem, cvae_with_mem.py:681; Q must be 1) */
    const float *weights_dev;               /* optional DEVICE pair {ex_weight, kl_weight} read by the kernel in place of
                                             * the two fields above: a captured graph follows a KL-annealing schedule
                                             * (kl_weight changed between steps) without being captured again */
    /* optional: the five numbers a step reports -- loss, re_surface, re_ex_feats, reconstruction_loss, kl_loss -- as
     * loss_vec (5 doubles) = mix (DEVICE 5 x 3 doubles, row-major) . sums, written by the LAST workgroup to finish,
     * which then puts sums and *ticket back to zero: sums and ticket (one unsigned) must be zero before the first call
     * and are zero again after every call (no fill and no matrix-vector launch per step); one call at a time per
     * (sums, ticket) pair.  All three NULL: sums is only accumulated into, as before. */
    const double *mix;
    double *loss_vec;
    unsigned *ticket;
} vsv_loss;
int vsv_loss_grad(const vsv_loss *d, void *stream);

/* Backward of the reparameterisation z = mu + exp(0.5 logvar) * eps (:199) and of the decoder-input cat (:1066),
 * the autograd steps between the decoder's and the encoders' backward passes: from dx (B, T, 2L) = gradient
 * w.r.t. [ctx_pad | z],
 *   d_enc (B, T, 2L) = [ dz + d_zm | dz * 0.5 * (z - mu) + d_zl ]  with dz = dx[.., Lc:] (+ d_z),
 *   d_ctx (B, Tc, Lc) = dx[:, :Tc, :Lc] (* ctx_mask (B, Tc), the per-row context lengths of multi-horizon training).
 * dx rows are [ctx part (Lc = context-embedding width) | z part (L)].
 * d_zm / d_zl (the KL term's gradients), d_z and ctx_mask may be NULL. */
int vsv_reparam_backward(const float *dx, const float *z, const float *zm, const float *d_zm, const float *d_zl,
                         const float *d_z, const float *ctx_mask, long B, int T, int Tc, int L, int Lc,
                         float *d_enc, float *d_ctx, void *stream);

/* Live nn.Parameters <-> packed blob on the device, one launch each way (train_step re-packs after every
 * optimizer step, reference :1131, and needs the parameters' gradients, :1127 loss.backward()).
 * vsv_pack_params: packed[i] = cst[i] + P(map_a[i]) + P(map_b[i]); a map entry is ((tensor + 1) << 24 | element),
 *   0 = nothing; P reads param_ptrs[tensor][element] (param_ptrs: DEVICE array of the float32 tensors' addresses;
 *   map_b carries the b_hh half of the b_ih + b_hh sums; cst, the parameter-independent part, may be NULL).  The
 *   maps are what vsv_pack_weights does to a state_dict, as indices (vsv/engine.py::build_gather_maps).
 * vsv_param_grads: flat[j] = sum_{k in [row_ptr[j], row_ptr[j+1])} gp[cols[k]]  (j over the n parameter elements in
 *   state_dict order, cols ascending: a fixed summation order) and partials[b] = sum of squares of workgroup b's 256
 *   elements (ceil(n / 256) floats) for the clip.
 * vsv_clip_grad_norm: torch.nn.utils.clip_grad_norm_(max_norm) (reference :1129) on that flat gradient in place:
 *   coef = min(1, max_norm / (sqrt(sum partials) + 1e-6)); norm_out (may be NULL) gets the norm. */
int vsv_pack_params(const void *const *param_ptrs, const unsigned *map_a, const unsigned *map_b, const float *cst,
                    float *packed, long n, void *stream);
int vsv_param_grads(const float *gp, const int *row_ptr, const int *cols, long n, float *flat, float *partials,
                    void *stream);
int vsv_clip_grad_norm(float *flat, long n, const float *partials, int n_partials, float max_norm, float *norm_out,
                       void *stream);

/* torch.optim.AdamW.step() (the reference's optimizer.step(), :1131) for all parameters in one launch.
 * params / exp_avg / exp_avg_sq: DEVICE arrays of n_tensors float32 tensor addresses (the optimizer's own state
 * tensors, updated in place); grad_flat: the n gradient elements in tensor order (vsv_param_grads' output, after
 * the clip); starts: n_tensors + 1 element offsets of the tensors in grad_flat (device).  step = the step count
 * AFTER this step (>= 1).  Same operation order as torch's foreach implementation (decoupled weight decay, lerp,
 * bias corrections as doubles); amsgrad / maximize are not built: the caller keeps torch's step() for those. */
int vsv_adamw_step(void *const *params, void *const *exp_avg, void *const *exp_avg_sq, const float *grad_flat,
                   const int *starts, int n_tensors, long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, long step, void *stream);

/* The dropout factors (0 or 1/(1-p)) the training kernels apply to layer `layer`'s output of
 * LSTM `id`: out (n_rows, T, hidden).  Test / debugging aid: the reference draws its masks from
 * torch's generator (inside nn.LSTM), so parity is checked with this mask injected into the
 * oracle and the keep rate is checked distributionally. */
int vsv_dropout_mask(float p, uint64_t seed, int id, int layer, long n_rows, int T, int hidden,
                     float *out, void *stream);

/* ---- profiling aid -----------------------------------------------------------------
 * Arms the NEXT vsv_decode call made by this thread: it records the two hipEvent_t
 * (created by the caller with timing enabled) on the call's stream immediately before
 * and after its decoder-LSTM kernel launch, so a benchmark can time that kernel live.
 * Pass NULLs to disarm.  No counterpart in the reference. */
void vsv_profile_next_decode(void *ev_start, void *ev_stop);

/* ---- diagnostics ------------------------------------------------------------------
 * Dumps the packed-blob layout (offsets and tile counts per layer) so a host-side
 * test can decode the fragments without a GPU.  No counterpart in the reference. */
int vsv_debug_layout(const vsv_dims *dims, long long *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* VSV_HIP_H */
