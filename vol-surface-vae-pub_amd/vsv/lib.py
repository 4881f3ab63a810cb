"""ctypes binding of libvsv_hip.so (the C ABI declared in include/vsv_hip.h).

Fails loudly: importing works anywhere (so the CPU test-suite can check the
exported symbols), but every compute call requires the library AND a GPU --
there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSV_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libvsv_hip.so")   # VSV_LIB: A/B builds

VSV_MAX_CONV = 4
VSV_MAX_INTERACT = 3


class VsvDims(C.Structure):
    _fields_ = [
        ("latent_dim", C.c_int), ("ex_feats_dim", C.c_int), ("mem_hidden", C.c_int),
        ("mem_layers", C.c_int), ("n_conv", C.c_int), ("surface_hidden", C.c_int * VSV_MAX_CONV),
        ("n_ctx_conv", C.c_int), ("ctx_surface_hidden", C.c_int * VSV_MAX_CONV),
        ("n_quantiles", C.c_int), ("interaction_layers", C.c_int), ("mem_type", C.c_int),
        ("compress_context", C.c_int), ("use_dense_surface", C.c_int),
    ]


_FP = C.POINTER(C.c_float)


class VsvLstmRaw(C.Structure):
    _fields_ = [("w_ih", _FP * 2), ("w_hh", _FP * 2), ("b_ih", _FP * 2), ("b_hh", _FP * 2)]


class VsvRawWeights(C.Structure):
    _fields_ = [
        ("enc_conv_w", _FP * VSV_MAX_CONV), ("enc_conv_b", _FP * VSV_MAX_CONV),
        ("enc_mem", VsvLstmRaw),
        ("z_mean_w", _FP), ("z_mean_b", _FP), ("z_log_var_w", _FP), ("z_log_var_b", _FP),
        ("ctx_conv_w", _FP * VSV_MAX_CONV), ("ctx_conv_b", _FP * VSV_MAX_CONV),
        ("ctx_mem", VsvLstmRaw),
        ("ctx_comp_w", _FP), ("ctx_comp_b", _FP),
        ("dec_mem", VsvLstmRaw),
        ("dec_in_w", _FP), ("dec_in_b", _FP),
        ("dec_deconv_w", _FP * VSV_MAX_CONV), ("dec_deconv_b", _FP * VSV_MAX_CONV),
        ("dec_out_w", _FP), ("dec_out_b", _FP),
        ("ex_in_w", _FP), ("ex_in_b", _FP), ("ex_out_w", _FP), ("ex_out_b", _FP),
        ("enc_int_w", _FP * VSV_MAX_INTERACT), ("enc_int_b", _FP * VSV_MAX_INTERACT),
        ("ctx_int_w", _FP * VSV_MAX_INTERACT), ("ctx_int_b", _FP * VSV_MAX_INTERACT),
        ("dec_int_w", _FP * VSV_MAX_INTERACT), ("dec_int_b", _FP * VSV_MAX_INTERACT),
    ]


class VsvDecodeIn(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p), ("ctx_rs", C.c_long), ("ctx_T", C.c_int), ("ctx_div", C.c_long),
        ("zb", C.c_void_p), ("lv", C.c_void_p), ("z_rs", C.c_long), ("z_T", C.c_int), ("z_div", C.c_long),
        ("mu0", C.c_float), ("std0", C.c_float),
        ("eps_in", C.c_void_p), ("eps_T", C.c_int), ("eps_t0", C.c_int),
        ("seed", C.c_uint64), ("S", C.c_long), ("day0", C.c_long), ("step0", C.c_int),
        ("state_in", C.c_void_p), ("state_div", C.c_long),
        ("gate_in", C.c_void_p), ("gate_out", C.c_void_p),
    ]


VSV_MAX_CHAIN = 8


class VsvTrainSave(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("emb_acts", "xe", "gates", "c", "h", "hout", "head_acts", "hd")] + \
               [("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_id", C.c_int), ("drop_seed_dev", C.c_void_p)]


class VsvDenseBwdLayer(C.Structure):
    _fields_ = [("wT", C.c_void_p), ("nqT", C.c_int), ("ntT", C.c_int),
                ("act_out", C.c_void_p), ("n_relu", C.c_int),
                ("g_out", C.c_void_p),
                ("add", C.c_void_p), ("add_f0", C.c_int), ("add_n", C.c_int)]


class VsvDenseBwd(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("layer", VsvDenseBwdLayer * VSV_MAX_CHAIN),
                ("n_rows", C.c_long), ("n_rowtiles", C.c_long), ("T", C.c_int), ("in_mode", C.c_int),
                ("in_", C.c_void_p), ("in_f", C.c_int), ("dx_out", C.c_void_p)]


class VsvLstmBwd(C.Structure):
    _fields_ = [("wT_hh", C.c_void_p * 2), ("wT_in", C.c_void_p * 2), ("nt_x", C.c_int),
                ("n_rows", C.c_long), ("n_rowtiles", C.c_long), ("T", C.c_int), ("t_out0", C.c_int),
                ("sv_gates", C.c_void_p), ("sv_c", C.c_void_p), ("dh_top", C.c_void_p),
                ("dgates", C.c_void_p), ("dx_tile", C.c_void_p), ("dx_rows", C.c_void_p), ("dx_f", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_id", C.c_int), ("gru", C.c_int),
                ("drop_seed_dev", C.c_void_p), ("sv_nl", C.c_int), ("sv_l0", C.c_int)]


class VsvTilebGemm(C.Structure):
    _fields_ = [("wT", C.c_void_p), ("nqT", C.c_int), ("ntT", C.c_int),
                ("g", C.c_void_p), ("g_stride", C.c_long),
                ("n_rows", C.c_long), ("n_rowtiles", C.c_long), ("T", C.c_int),
                ("out_tile", C.c_void_p), ("out_rows", C.c_void_p), ("out_f", C.c_int),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_seed_dev", C.c_void_p),
                ("drop_id", C.c_int), ("drop_layer", C.c_int)]


VSV_MAX_WGRAD = 24


class VsvWgradProb(C.Structure):
    _fields_ = [("g", C.c_void_p), ("g_stride", C.c_long), ("n_a", C.c_int),
                ("x", C.c_void_p * 2), ("x_stride", C.c_long * 2), ("n_b", C.c_int * 2), ("shift", C.c_int * 2),
                ("n_inst", C.c_int), ("T", C.c_int),
                ("w_off", C.c_long), ("b_off", C.c_long), ("nt", C.c_int), ("bias", C.c_int)]


class VsvWgrad(C.Structure):
    _fields_ = [("n", C.c_int), ("gp", C.c_void_p), ("p", VsvWgradProb * VSV_MAX_WGRAD)]


class VsvLoss(C.Structure):
    _fields_ = [("pred", C.c_void_p), ("target", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("T", C.c_int), ("Q", C.c_int),
                ("q", C.c_float * 8), ("qw", C.c_float * 8),
                ("ex_pred", C.c_void_p), ("ex_target", C.c_void_p),
                ("X", C.c_int), ("ex_n", C.c_int), ("ex_l2", C.c_int),
                ("ex_weight", C.c_float), ("kl_weight", C.c_float),
                ("zm", C.c_void_p), ("zl", C.c_void_p), ("L", C.c_int),
                ("d_pred", C.c_void_p), ("d_ex", C.c_void_p), ("d_zm", C.c_void_p), ("d_zl", C.c_void_p),
                ("sums", C.c_void_p), ("Hs", C.c_int), ("grad_scale", C.c_float), ("surf_l2", C.c_int),
                ("weights_dev", C.c_void_p), ("mix", C.c_void_p), ("loss_vec", C.c_void_p), ("ticket", C.c_void_p)]


# name -> (restype, argtypes); the full list of symbols include/vsv_hip.h declares
SIGNATURES = {
    "vsv_abi_version": (C.c_int, []),
    "vsv_last_error": (C.c_char_p, []),
    "vsv_packed_floats": (C.c_size_t, [C.POINTER(VsvDims)]),
    "vsv_pack_weights": (C.c_int, [C.POINTER(VsvDims), C.POINTER(VsvRawWeights), _FP]),
    "vsv_encode_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int]),
    "vsv_encode": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int,
                             C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                             C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_windows_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int]),
    "vsv_encode_windows": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_windows_noalloc": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_long, C.c_int, C.c_void_p, C.c_void_p]),
    "vsv_init_device": (C.c_int, []),
    "vsv_enc_state_floats": (C.c_size_t, [C.POINTER(VsvDims)]),
    "vsv_encode_state": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                   C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_windows_state": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_windows_split_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int]),
    "vsv_encode_windows_layer0": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_windows_layer1": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_long, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_decode_prefix_split_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int]),
    "vsv_decode_prefix_layer0": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.POINTER(VsvDecodeIn), C.c_long, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_decode_prefix_layer1": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.POINTER(VsvDecodeIn), C.c_long, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_context_prefix_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int, C.c_int]),
    "vsv_context_prefix": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_pre_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long]),
    "vsv_encode_pre": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                                 C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_encode_from_pre_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int]),
    "vsv_encode_from_pre": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_long,
                                      C.c_long, C.c_int, C.c_void_p, C.c_long, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_state_floats": (C.c_size_t, [C.POINTER(VsvDims)]),
    "vsv_gate_floats": (C.c_size_t, [C.POINTER(VsvDims)]),
    "vsv_decode_ws_bytes": (C.c_size_t, [C.POINTER(VsvDims), C.c_long, C.c_int, C.c_int]),
    "vsv_decode": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.POINTER(VsvDecodeIn),
                             C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "vsv_percentile": (C.c_int, [C.c_void_p, C.c_long, C.c_int, C.POINTER(C.c_double), C.c_int,
                                 C.c_int, C.c_long, C.c_void_p, C.c_void_p]),
    "vsv_profile_next_decode": (None, [C.c_void_p, C.c_void_p]),
    "vsv_reparam_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsv_pack_params": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]),
    "vsv_param_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsv_clip_grad_norm": (C.c_int, [C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "vsv_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_long,
                                 C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_long, C.c_void_p]),
    "vsv_dropout_mask": (C.c_int, [C.c_float, C.c_uint64, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "vsv_train_encode": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                   C.c_long, C.c_int, C.c_void_p, C.POINTER(VsvTrainSave), C.c_void_p]),
    "vsv_train_encode_pair": (C.c_int, [C.POINTER(VsvDims), C.c_void_p,
                                        C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                        C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.POINTER(VsvTrainSave), C.POINTER(VsvTrainSave), C.c_void_p]),
    "vsv_train_decode": (C.c_int, [C.POINTER(VsvDims), C.c_void_p, C.POINTER(VsvDecodeIn),
                                   C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(VsvTrainSave), C.c_void_p]),
    "vsv_dense_backward": (C.c_int, [C.POINTER(VsvDenseBwd), C.c_void_p]),
    "vsv_lstm_backward": (C.c_int, [C.POINTER(VsvLstmBwd), C.c_int, C.c_int, C.c_void_p]),
    "vsv_dense_backward_pair": (C.c_int, [C.POINTER(VsvDenseBwd), C.POINTER(VsvDenseBwd), C.c_void_p]),
    "vsv_lstm_backward_pair": (C.c_int, [C.POINTER(VsvLstmBwd), C.POINTER(VsvLstmBwd), C.c_int, C.c_int, C.c_void_p]),
    "vsv_tileb_gemm_launch_pair": (C.c_int, [C.POINTER(VsvTilebGemm), C.POINTER(VsvTilebGemm), C.c_void_p]),
    "vsv_loss_grad": (C.c_int, [C.POINTER(VsvLoss), C.c_void_p]),
    "vsv_tileb_gemm_launch": (C.c_int, [C.POINTER(VsvTilebGemm), C.c_void_p]),
    "vsv_weight_grads": (C.c_int, [C.POINTER(VsvWgrad), C.c_void_p]),
    "vsv_debug_layout": (C.c_int, [C.POINTER(VsvDims), C.POINTER(C.c_longlong), C.c_int]),
    "vsv_gmm_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_void_p,
                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vsv_philox_normal": (C.c_int, [C.c_uint64, C.c_long, C.c_long, C.c_long, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library and attach the signatures.  Raises if the
    library has not been built (python __graft_entry__.py / make -C csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C vol-surface-vae-pub_amd/csrc` "
                "(there is no CPU fallback for the HIP path)")
        # torch ships its own HIP/HSA runtime (torch/lib/libamdhip64.so, same SONAME as
        # /opt/rocm's).  It must be the first one loaded: if this library pulled in
        # /opt/rocm's copy before torch came up, the process would hold two HSA runtimes
        # and every launch would fail with hipErrorNoDevice.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.vsv_abi_version() != 23:
            raise RuntimeError("libvsv_hip.so ABI version mismatch")
        _lib = lib
    return _lib


class VsvError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().vsv_last_error().decode("utf-8", "replace")
        if rc < 0:
            raise ValueError(f"{what}: {msg}")
        raise VsvError(f"{what}: HIP error {rc}: {msg}")


def make_dims(latent_dim, ex_feats_dim, mem_hidden, mem_layers, surface_hidden, ctx_surface_hidden,
              n_quantiles, interaction_layers=0, mem_type="lstm", compress_context=True,
              use_dense_surface=False) -> VsvDims:
    d = VsvDims()
    d.compress_context = int(bool(compress_context))
    d.use_dense_surface = int(bool(use_dense_surface))
    # the reference builds nn.LSTM for "lstm", nn.GRU for "gru" and nn.RNN for ANY other value (:154-159)
    d.mem_type = {"lstm": 0, "gru": 1}.get(mem_type, 2)
    d.interaction_layers = int(interaction_layers or 0)
    if d.interaction_layers + 1 > VSV_MAX_INTERACT:
        raise NotImplementedError("at most 2 interaction layers are supported by the HIP path")
    d.latent_dim, d.ex_feats_dim, d.mem_hidden, d.mem_layers = latent_dim, ex_feats_dim, mem_hidden, mem_layers
    if len(surface_hidden) > VSV_MAX_CONV or len(ctx_surface_hidden) > VSV_MAX_CONV:
        raise NotImplementedError("at most 4 conv layers are supported by the HIP path")
    d.n_conv = len(surface_hidden)
    d.n_ctx_conv = len(ctx_surface_hidden)
    for i, c in enumerate(surface_hidden):
        d.surface_hidden[i] = int(c)
    for i, c in enumerate(ctx_surface_hidden):
        d.ctx_surface_hidden[i] = int(c)
    d.n_quantiles = n_quantiles
    return d
