"""Host-side driver of the HIP path: packs a CVAEMemRand state_dict for the
kernels and launches them on torch-owned device memory (torch is plumbing
here: allocation, streams, dtype casts).  Every compute goes through the C ABI
of libvsv_hip.so; nothing in this module computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib


def dims_from_config(cfg: dict) -> _lib.VsvDims:
    """vsv_dims from a reference model config dict
    (keys as validated by cvae_with_mem_randomized.py:969-1020)."""
    if cfg.get("ex_feats_hidden") is not None or cfg.get("ctx_ex_feats_hidden") is not None:
        raise NotImplementedError("HIP path implements ex_feats_hidden=None only")
    if tuple(cfg["feat_dim"]) != (5, 5):
        raise NotImplementedError("HIP path implements the 5x5 grid only")
    dims = _lib.make_dims(cfg["latent_dim"], cfg["ex_feats_dim"], cfg["mem_hidden"], cfg["mem_layers"],
                          list(cfg["surface_hidden"]), list(cfg["ctx_surface_hidden"]),
                          len(cfg.get("quantiles", [0.05, 0.5, 0.95])), cfg.get("interaction_layers") or 0,
                          cfg.get("mem_type", "lstm"), bool(cfg.get("compress_context", False)),
                          bool(cfg.get("use_dense_surface", False)))
    # shapes the kernels cannot run are refused here, when the model / Engine is built, not at the first launch
    L = _lib.load()
    if L.vsv_packed_floats(C.byref(dims)) == 0:
        raise NotImplementedError("HIP path: unsupported model shape: " + L.vsv_last_error().decode())
    return dims


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def pack_state_dict(dims: _lib.VsvDims, sd: Dict[str, np.ndarray]) -> np.ndarray:
    """state_dict (name -> array, reference key names) -> packed float32 blob."""
    L = _lib.load()
    keep = []          # keep the float32 copies alive until packing is done

    def get(name):
        a = np.ascontiguousarray(np.asarray(sd[name], dtype=np.float32))
        keep.append(a)
        return _fp(a)

    raw = _lib.VsvRawWeights()

    def lstm(dst, prefix):
        for l in range(dims.mem_layers):
            dst.w_ih[l] = get(f"{prefix}.weight_ih_l{l}")
            dst.w_hh[l] = get(f"{prefix}.weight_hh_l{l}")
            dst.b_ih[l] = get(f"{prefix}.bias_ih_l{l}")
            dst.b_hh[l] = get(f"{prefix}.bias_hh_l{l}")

    # use_dense_surface: the Linear stacks enc_dense_i / ctx_enc_dense_i / dec_dense_i ride in the conv slots
    kind, dkind = ("dense", "dense") if dims.use_dense_surface else ("conv", "deconv")
    for i in range(dims.n_conv):
        raw.enc_conv_w[i] = get(f"encoder.surface_embedding.enc_{kind}_{i}.weight")
        raw.enc_conv_b[i] = get(f"encoder.surface_embedding.enc_{kind}_{i}.bias")
    lstm(raw.enc_mem, "encoder.mem")
    raw.z_mean_w, raw.z_mean_b = get("encoder.z_mean_layer.weight"), get("encoder.z_mean_layer.bias")
    raw.z_log_var_w, raw.z_log_var_b = get("encoder.z_log_var_layer.weight"), get("encoder.z_log_var_layer.bias")
    for i in range(dims.n_ctx_conv):
        raw.ctx_conv_w[i] = get(f"ctx_encoder.ctx_surface_embedding.ctx_enc_{kind}_{i}.weight")
        raw.ctx_conv_b[i] = get(f"ctx_encoder.ctx_surface_embedding.ctx_enc_{kind}_{i}.bias")
    lstm(raw.ctx_mem, "ctx_encoder.mem")
    if dims.compress_context:
        raw.ctx_comp_w, raw.ctx_comp_b = get("ctx_encoder.final_compression.weight"), get("ctx_encoder.final_compression.bias")
    lstm(raw.dec_mem, "decoder.mem")
    raw.dec_in_w, raw.dec_in_b = get("decoder.surface_decoder_input.weight"), get("decoder.surface_decoder_input.bias")
    names = [f"dec_{dkind}_{i}" for i in range(dims.n_conv - 1)] + ["dec_final"]
    for i, nm in enumerate(names):
        raw.dec_deconv_w[i] = get(f"decoder.surface_decoder.{nm}.weight")
        raw.dec_deconv_b[i] = get(f"decoder.surface_decoder.{nm}.bias")
    raw.dec_out_w, raw.dec_out_b = get("decoder.surface_decoder.dec_output.weight"), get("decoder.surface_decoder.dec_output.bias")
    if dims.interaction_layers > 0:
        # n Linear+ReLU and a final Linear per module (reference :131-142, :283-294, :452-463)
        for dst_w, dst_b, mod, pf in ((raw.enc_int_w, raw.enc_int_b, "encoder", "enc"),
                                      (raw.ctx_int_w, raw.ctx_int_b, "ctx_encoder", "ctx"),
                                      (raw.dec_int_w, raw.dec_int_b, "decoder", "dec")):
            names = [f"{pf}_interact_linear_{i}" for i in range(dims.interaction_layers)] + [f"{pf}_interact_linear_final"]
            for i, nm in enumerate(names):
                dst_w[i] = get(f"{mod}.interaction.{nm}.weight")
                dst_b[i] = get(f"{mod}.interaction.{nm}.bias")
    if dims.ex_feats_dim > 0:
        raw.ex_in_w, raw.ex_in_b = get("decoder.ex_feats_decoder_input.weight"), get("decoder.ex_feats_decoder_input.bias")
        raw.ex_out_w, raw.ex_out_b = get("decoder.ex_feats_decoder.weight"), get("decoder.ex_feats_decoder.bias")

    n = L.vsv_packed_floats(C.byref(dims))
    if n == 0:
        raise ValueError("vsv_packed_floats: " + L.vsv_last_error().decode())
    out = np.empty(n, dtype=np.float32)
    _lib.check(L.vsv_pack_weights(C.byref(dims), C.byref(raw), _fp(out)), "vsv_pack_weights")
    return out


def _fold_geometry(dims: _lib.VsvDims):
    """(w_off, b_off, nq, nt) of the decoder head's Linear, first ConvT and folded section, and whether the
    Linear reads its input in slot order (after interaction layers) -- from vsv_debug_layout."""
    L = _lib.load()
    n = L.vsv_debug_layout(C.byref(dims), None, 0)
    buf = (C.c_longlong * n)()
    assert L.vsv_debug_layout(C.byref(dims), buf, n) == n
    v = [int(x) for x in buf]
    MC = 8
    nrec = 2 * (MC + 3) + 2 + MC
    fold = v[6 * nrec + 16: 6 * nrec + 22]
    fold_at = v[6 * nrec + 22]
    d0 = 2 * (MC + 3) + 2                                        # first decoder head record
    lin = v[6 * (d0 + fold_at): 6 * (d0 + fold_at) + 6]
    conv = v[6 * (d0 + fold_at + 1): 6 * (d0 + fold_at + 1) + 6]
    return dict(lin=lin[:4], conv=conv[:4], fold=fold[:4], perm_in=fold_at > 0)


def _wpos(w_off, nt, n, k):
    """Blob position of W[n][k] of a packed section (vsv_layout.h): n = 32*tile + row, k = 2*kstep + half."""
    return w_off + (((k >> 3) * nt + (n >> 5)) * 64 + ((k & 1) * 32 + (n & 31))) * 4 + ((k >> 1) & 3)


def _bpos(b_off, n):
    t, i = n >> 5, n & 31
    return b_off + (t * 2 + ((i >> 2) & 1)) * 16 + 4 * (i >> 3) + (i & 3)


def _k_of_feature(f, perm):
    """k index (2*kstep + half) at which a section reads input feature f: natural order, or the slot order
    the previous layer's accumulators leave (feature 32t + 8a + 4h + b sits at slot 16t + 4a + b, half h)."""
    if not perm:
        return f
    t, i = f >> 5, f & 31
    return 2 * (16 * t + 4 * (i >> 3) + (i & 3)) + ((i >> 2) & 1)


def build_fold_maps(dims: _lib.VsvDims):
    """Index maps for folding the decoder head's Linear into its first ConvT on the device:
    dense(A) = blob[a_w] (128 x 128), dense(B) = blob[b_w] (128 x K), F = A @ B -> blob[f_w], and the biases."""
    g = _fold_geometry(dims)
    n = np.arange(128)
    K = 8 * g["lin"][2]
    kin = _k_of_feature(np.arange(K), g["perm_in"])
    kmid = _k_of_feature(np.arange(128), True)                   # the ConvT reads the Linear's slot order
    a_w = _wpos(g["conv"][0], g["conv"][3], n[:, None], kmid[None, :])
    b_w = _wpos(g["lin"][0], g["lin"][3], n[:, None], kin[None, :])
    f_w = _wpos(g["fold"][0], g["fold"][3], n[:, None], kin[None, :])
    sec_floats = g["fold"][2] * g["fold"][3] * 256
    return dict(a_w=a_w, b_w=b_w, f_w=f_w, a_b=_bpos(g["conv"][1], n), b_b=_bpos(g["lin"][1], n),
                f_b=_bpos(g["fold"][1], n), fold_w=(g["fold"][0], g["fold"][0] + sec_floats),
                fold_b=(g["fold"][1], g["fold"][1] + g["fold"][3] * 32))


def build_gather_maps(dims: _lib.VsvDims, spec):
    """Index maps that express the packing as two gathers: with
    flat = cat([0, p.flatten() for p in params in `spec` order]),
    packed = flat[idx_a] + flat[idx_b]  (idx_b is non-zero only where the packer
    sums b_ih + b_hh).  Built by packing state_dicts whose values are their own
    (1-based) flat offsets, so it is exactly vsv_pack_weights' mapping.
    spec: list of (name, shape)."""
    fake_a, fake_b = {}, {}
    off = 1
    for name, shape in spec:
        n = int(np.prod(shape))
        ids = np.arange(off, off + n, dtype=np.float32).reshape(shape)
        is_bhh = ".bias_hh_l" in name
        fake_a[name] = np.zeros(shape, np.float32) if is_bhh else ids
        fake_b[name] = ids if is_bhh else np.zeros(shape, np.float32)
        off += n
    assert off < 2 ** 24, "parameter count exceeds float32-exact index range"
    idx_a = pack_state_dict(dims, fake_a)
    idx_b = pack_state_dict(dims, fake_b)
    # constants the packer writes whatever the parameters are (the identity "head" of an uncompressed context
    # embedding): they are not gathers -- set_weights_from_params adds them from const_blob(dims)
    const = pack_state_dict(dims, {k: np.zeros_like(v) for k, v in fake_a.items()})
    idx_a[const != 0] = 0
    idx_b[const != 0] = 0
    # the folded Linear . ConvT section is a PRODUCT of parameters, not a gather: nothing maps there
    # (set_weights_from_params computes it on the device, build_fold_maps)
    fm = build_fold_maps(dims)
    for lo, hi in (fm["fold_w"], fm["fold_b"]):
        idx_a[lo:hi] = 0
        idx_b[lo:hi] = 0
    return idx_a.astype(np.int64), idx_b.astype(np.int64)


def const_blob(dims: _lib.VsvDims, spec) -> Optional[np.ndarray]:
    """The parameter-independent part of the packed blob (None when there is none)."""
    c = pack_state_dict(dims, {name: np.zeros(shape, np.float32) for name, shape in spec})
    return c if np.any(c != 0) else None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """Packed weights on one GPU + launch helpers.  All tensor arguments are
    float32 CUDA tensors; outputs are freshly allocated float32 CUDA tensors."""

    def __init__(self, cfg: dict, state_dict: Optional[Dict[str, np.ndarray]], device="cuda:0"):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("the HIP path needs a GPU (device must be a cuda/ROCm device); "
                               "there is no CPU fallback")
        self.dims = dims_from_config(cfg)
        self.L = self.dims.latent_dim
        # width of the context embedding: latent_dim with compress_context, else the context memory's hidden size
        self.Lc = self.L if self.dims.compress_context else self.dims.mem_hidden
        self.X = self.dims.ex_feats_dim
        self.Q = self.dims.n_quantiles
        self.state_floats = int(self.lib.vsv_state_floats(C.byref(self.dims)))
        self.gate_floats = int(self.lib.vsv_gate_floats(C.byref(self.dims)))
        self.packed = None
        self._maps = None
        self._const = None
        self._pmaps = None
        self._from_params = False
        self._fold = None
        self._side = None
        if state_dict is not None:
            self.set_weights(state_dict)

    def set_weights(self, state_dict: Dict[str, np.ndarray]) -> None:
        """Host packing through vsv_pack_weights (numpy state_dict)."""
        blob = pack_state_dict(self.dims, state_dict)
        self.packed = torch.from_numpy(blob).to(self.device)
        self._from_params = False
        self._fold_stale = False                               # (the host packer writes the folded section itself)

    def set_weights_from_params(self, named_params) -> torch.Tensor:
        """Device-side packing of live nn.Parameters through the index maps of build_gather_maps: one launch of
        vsv_pack_params reading the float32 tensors in place through a device table of their addresses
        (VSV_NATIVE_PACK=0, other dtypes / devices: torch.cat and two index gathers)."""
        named_params = list(named_params)
        if self._maps is None:
            spec = [(n, tuple(p.shape)) for n, p in named_params]
            ia, ib = build_gather_maps(self.dims, spec)
            self._maps = (torch.from_numpy(ia).to(self.device), torch.from_numpy(ib).to(self.device))
            cb = const_blob(self.dims, spec)
            self._const = None if cb is None else torch.from_numpy(cb).to(self.device)
            # the same maps as (tensor + 1) << 24 | element for the one-launch native gather (vsv_pack_params)
            sizes = np.array([int(np.prod(s)) for _, s in spec], dtype=np.int64)
            self._pmaps = None
            if len(spec) < 255 and int(sizes.max()) < (1 << 24):
                starts = np.concatenate([[1], 1 + np.cumsum(sizes)])
                enc = []
                for m in (ia, ib):
                    t = np.searchsorted(starts, m, side="right") - 1          # tensor of a 1-based flat offset
                    e = ((t + 1) << 24) | (m - starts[np.clip(t, 0, len(sizes) - 1)])
                    enc.append(torch.from_numpy(np.where(m == 0, 0, e).astype(np.uint32).view(np.int32)).to(self.device))
                self._pmaps = tuple(enc)
            self._ptr_key, self._ptr_tab = None, None
        ts = [p for _, p in named_params]
        native = (self._pmaps is not None and os.environ.get("VSV_NATIVE_PACK", "1") != "0" and
                  all(p.device == self.device and p.dtype == torch.float32 and p.is_contiguous() for p in ts))
        if native:
            key = tuple(p.data_ptr() for p in ts)
            if key != self._ptr_key:
                if torch.cuda.is_current_stream_capturing():
                    native = False                                 # (the table upload is a host copy: not inside a capture)
                else:
                    self._ptr_tab = torch.tensor(key, dtype=torch.int64).to(self.device)
                    self._ptr_key = key
        if native:
            n = self._maps[0].numel()
            packed = torch.empty(n, dtype=torch.float32, device=self.device)
            rc = self.lib.vsv_pack_params(self._ptr_tab.data_ptr(), self._pmaps[0].data_ptr(), self._pmaps[1].data_ptr(),
                                          None if self._const is None else self._const.data_ptr(), packed.data_ptr(), n,
                                          self._stream())
            if rc != 0:
                raise RuntimeError("vsv_pack_params: " + self.lib.vsv_last_error().decode())
            # the blob is a pure function of _ptr_tab's tensors; later gathers (gather_from_params) may read them
            # again only if they all outlive this call: nn.Parameters, not expressions packed in a parameter's place
            self._from_params = all(isinstance(p, torch.nn.Parameter) for p in ts)
        else:
            self._from_params = False
            flat = torch.cat([torch.zeros(1, dtype=torch.float32, device=self.device)] +
                             [p.reshape(-1).to(device=self.device, dtype=torch.float32) for _, p in named_params])
            packed = flat[self._maps[0]] + flat[self._maps[1]]
            if self._const is not None:
                packed = packed + self._const
        # generation's folded head layer: (first ConvT) . (Linear), two small matmuls on the gathered blob
        if self._fold is None:
            fm = build_fold_maps(self.dims)
            self._fold = {k: torch.from_numpy(np.ascontiguousarray(v).astype(np.int64)).to(self.device)
                          for k, v in fm.items() if k not in ("fold_w", "fold_b")}
        # (computed when a generation call first needs it -- Engine.decode --, not on every re-pack: a training
        # step re-packs after each optimizer step and never reads the folded section)
        self.packed = packed.detach()
        self._fold_stale = True
        return packed

    def gather_from_params(self, idx1: torch.Tensor, cache: dict) -> Optional[torch.Tensor]:
        """blob_ext[idx1] with blob_ext = [0, packed...] (idx1: int64 positions, 0 = the zero slot), read straight
        from the parameters by one vsv_pack_params launch through maps composed once (kept in `cache`); None
        when the current blob was not packed from live float32 parameters."""
        if not self._from_params or self._pmaps is None:
            return None
        m = cache.get("maps")
        if m is None:
            z = torch.zeros(1, dtype=torch.int32, device=self.device)
            m = [torch.cat([z, pm])[idx1].contiguous() for pm in self._pmaps]
            m.append(None if self._const is None else
                     torch.cat([torch.zeros(1, dtype=torch.float32, device=self.device), self._const])[idx1].contiguous())
            cache["maps"] = m
        out = torch.empty(idx1.numel(), dtype=torch.float32, device=self.device)
        rc = self.lib.vsv_pack_params(self._ptr_tab.data_ptr(), m[0].data_ptr(), m[1].data_ptr(),
                                      None if m[2] is None else m[2].data_ptr(), out.data_ptr(), out.numel(), self._stream())
        if rc != 0:
            raise RuntimeError("vsv_pack_params: " + self.lib.vsv_last_error().decode())
        return out

    def _ensure_fold(self) -> None:
        """The folded (first ConvT) . (Linear) section of a blob packed from live parameters, on demand."""
        if not getattr(self, "_fold_stale", False):
            return
        f = self._fold
        with torch.no_grad():
            packed = self.packed
            A, Bm = packed[f["a_w"]], packed[f["b_w"]]
            packed.index_copy_(0, f["f_w"].reshape(-1), (A @ Bm).reshape(-1))
            packed.index_copy_(0, f["f_b"], A @ packed[f["b_b"]] + packed[f["a_b"]])
        self._fold_stale = False

    # -- helpers ---------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ws(self, nbytes: int) -> torch.Tensor:
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)

    # -- encoders ----------------------------------------------------------
    def encode(self, which: int, surface: torch.Tensor, ex: Optional[torch.Tensor] = None) -> torch.Tensor:
        """surface (B,T,5,5) [, ex (B,T,X)] -> (B,T,2L) (which=0: [z_mean|z_log_var])
        or (B,T,L) (which=1: context embedding)."""
        surface = _f32c(surface, self.device)
        B, T = surface.shape[:2]
        if self.X > 0:
            if ex is None:
                raise ValueError("model has ex_feats_dim > 0 but no ex_feats were given")
            ex = _f32c(ex, self.device)
        return self.encode_strided(which, surface, T * 25, 25, ex, T * self.X, self.X, B, T)

    def encode_strided(self, which, surf, surf_rs, surf_ts, ex, ex_rs, ex_ts, n_rows, T) -> torch.Tensor:
        nh = self.Lc if which else 2 * self.L
        out = torch.empty((n_rows, T, nh), dtype=torch.float32, device=self.device)
        nws = self.lib.vsv_encode_ws_bytes(C.byref(self.dims), n_rows, T)
        ws = self._ws(nws)
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode(C.byref(self.dims), _ptr(self.packed), which,
                                     _ptr(surf), surf_rs, surf_ts, _ptr(ex), ex_rs, ex_ts,
                                     n_rows, T, _ptr(out), _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode")
        return out

    def encode_state(self, which, surf, surf_rs, surf_ts, ex, ex_rs, ex_ts, n_rows, T, state_in=None, state_div=1,
                     want_state=False):
        """vsv_encode_state: encoder `which` over T steps starting from state_in[row // state_div] (or the
        zero state) -> (out (n_rows, T, nh), state (n_rows, enc_state_floats) or None)."""
        nh = self.Lc if which else 2 * self.L
        out = torch.empty((n_rows, T, nh), dtype=torch.float32, device=self.device)
        st_out = None
        if want_state:
            st_out = torch.empty((n_rows, int(self.lib.vsv_enc_state_floats(C.byref(self.dims)))),
                                 dtype=torch.float32, device=self.device)
        ws = self._ws(self.lib.vsv_encode_ws_bytes(C.byref(self.dims), n_rows, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode_state(C.byref(self.dims), _ptr(self.packed), which,
                                           _ptr(surf), surf_rs, surf_ts, _ptr(ex), ex_rs, ex_ts, n_rows, T,
                                           _ptr(state_in), state_div, _ptr(st_out),
                                           _ptr(out), _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode_state")
        return out, st_out

    def pre_table_floats(self, n_rows: int) -> int:
        """floats of one day's pre-activation table for n_rows rows: [32 * NTm gate values][rows padded to 32]."""
        ntm = int(self.lib.vsv_enc_state_floats(C.byref(self.dims))) // (16 * self.dims.mem_layers)   # gate tiles as built
        return 32 * ntm * (((n_rows + 31) // 32) * 32)

    def encode_pre(self, which, surf, surf_rs, ex, ex_rs, n_rows, pre_out: torch.Tensor):
        """vsv_encode_pre: the position-independent part of encoder `which` (embedding + first-layer input
        product) of one day per row -> pre_out (a view of pre_table_floats(n_rows) floats)."""
        assert pre_out.is_contiguous() and pre_out.numel() == self.pre_table_floats(n_rows)
        ws = self._ws(self.lib.vsv_encode_pre_ws_bytes(C.byref(self.dims), n_rows))
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode_pre(C.byref(self.dims), _ptr(self.packed), which, _ptr(surf), surf_rs,
                                         _ptr(ex), ex_rs, n_rows, _ptr(pre_out), ((n_rows + 31) // 32) * 32,
                                         _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode_pre")

    def encode_from_pre(self, which, pre: torch.Tensor, n_rows, T, state_in=None, state_div=1):
        """vsv_encode_from_pre: encoder `which` over the T days whose tables are pre[0:T] (T, table floats)."""
        assert pre.is_contiguous() and pre.shape[0] >= T and pre[0].numel() == self.pre_table_floats(n_rows)
        nh = self.Lc if which else 2 * self.L
        out = torch.empty((n_rows, T, nh), dtype=torch.float32, device=self.device)
        ws = self._ws(self.lib.vsv_encode_from_pre_ws_bytes(C.byref(self.dims), n_rows, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode_from_pre(C.byref(self.dims), _ptr(self.packed), which, _ptr(pre),
                                              ((n_rows + 31) // 32) * 32, pre[0].numel(), n_rows, T,
                                              _ptr(state_in), state_div, None,
                                              _ptr(out), _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode_from_pre")
        return out

    def encode_both_strided(self, surf, surf_rs, surf_ts, ex, ex_rs, ex_ts, n_rows, T, T_ctx=None):
        """Posterior encoder (which=0, T steps) and context encoder (which=1, T_ctx steps) of the
        same windows on two HIP streams: they are independent and, for a date-range prefix,
        each is a latency-bound launch that fills only part of the chip."""
        T_ctx = T if T_ctx is None else T_ctx
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        side = self._side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ctx = self.encode_strided(1, surf, surf_rs, surf_ts, ex, ex_rs, ex_ts, n_rows, T_ctx)
        enc = self.encode_strided(0, surf, surf_rs, surf_ts, ex, ex_rs, ex_ts, n_rows, T)
        main.wait_stream(side)
        ctx.record_stream(main)
        return enc, ctx

    MAX_WINDOWS = 8192          # vsv_encode_windows: one-wave-per-gate-tile schedule only

    def encode_windows(self, which, vol, ex, n_windows, T, state_in=None, want_state=False):
        """Encoder `which` over the overlapping windows vol[r : r + T] (r < n_windows) of a contiguous
        calendar series vol (>= n_windows + T - 1, 5, 5) [, ex (.., X)]: vsv_encode_windows(_state).
        state_in (n_windows, enc_state_floats): continue from it; want_state: -> (out, state)."""
        nh = self.Lc if which else 2 * self.L
        out = torch.empty((n_windows, T, nh), dtype=torch.float32, device=self.device)
        ws = self._ws(self.lib.vsv_encode_windows_ws_bytes(C.byref(self.dims), n_windows, T))
        st_out = None
        if want_state:
            st_out = torch.empty((n_windows, int(self.lib.vsv_enc_state_floats(C.byref(self.dims)))),
                                 dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode_windows_state(C.byref(self.dims), _ptr(self.packed), which, _ptr(vol), _ptr(ex),
                                                   n_windows, T, _ptr(state_in), _ptr(st_out),
                                                   _ptr(out), _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode_windows")
        return (out, st_out) if want_state else out

    # -- the date-range prefix as layer stages (vsv_*_layer0 / _layer1) ------------------------------
    def enc_state_floats(self) -> int:
        return int(self.lib.vsv_enc_state_floats(C.byref(self.dims)))

    def encode_windows_layer0(self, which, vol, ex, n_windows, T, state_in, state_out, ws=None):
        """Layer 0 of encoder `which` over the windows (on the current stream); h0 of every step stays in the
        returned workspace, which the matching encode_windows_layer1 call takes."""
        if ws is None:
            ws = self._ws(self.lib.vsv_encode_windows_split_ws_bytes(C.byref(self.dims), n_windows, T))
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode_windows_layer0(C.byref(self.dims), _ptr(self.packed), which, _ptr(vol), _ptr(ex),
                                                    n_windows, T, _ptr(state_in), _ptr(state_out),
                                                    _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode_windows_layer0")
        return ws

    def encode_windows_layer1(self, which, n_windows, T, state_in, state_out, ws):
        nh = self.Lc if which else 2 * self.L
        out = torch.empty((n_windows, T, nh), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_encode_windows_layer1(C.byref(self.dims), _ptr(self.packed), which, n_windows, T,
                                                    _ptr(state_in), _ptr(state_out), _ptr(out),
                                                    _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_encode_windows_layer1")
        return out

    def decode_prefix_layer(self, layer: int, n_rows: int, T: int, ctx, zb, state_in, state_out, gates, ws=None):
        """Layer `layer` of the decoder's context prefix over (ctx_t, z_mean_t) (ctx, zb: (n_rows, T, L))."""
        din = _lib.VsvDecodeIn()
        if layer == 0:
            din.ctx, din.ctx_rs, din.ctx_T, din.ctx_div = ctx.data_ptr(), T * self.Lc, T, 1
            din.zb, din.z_rs, din.z_T, din.z_div = zb.data_ptr(), T * self.L, T, 1
        din.S, din.state_div = 1, 1
        if state_in is not None:
            din.state_in = state_in.data_ptr()
        if gates is not None:
            din.gate_out = gates.data_ptr()
        if ws is None:
            ws = self._ws(self.lib.vsv_decode_prefix_split_ws_bytes(C.byref(self.dims), n_rows, T))
        fn = self.lib.vsv_decode_prefix_layer0 if layer == 0 else self.lib.vsv_decode_prefix_layer1
        with torch.cuda.device(self.device):
            rc = fn(C.byref(self.dims), _ptr(self.packed), C.byref(din), n_rows, T, _ptr(state_out),
                    _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, f"vsv_decode_prefix_layer{layer}")
        return ws

    def context_prefix(self, vol, ex, n_days: int, ctx_len: int, n_seg: int, n_streams: int = 4):
        """vsv_context_prefix: decoder state after the context steps + first-step shared gates for the windows
        vol[d : d + ctx_len], as a pipeline of layer stages over time segments on the engine's worker streams."""
        if n_streams == 2:
            # the caller's stream and the engine's side stream (the pair encode_both_windows runs on)
            if self._side is None:
                self._side = torch.cuda.Stream(self.device)
            handles = [torch.cuda.current_stream(self.device).cuda_stream, self._side.cuda_stream]
        else:
            pool = getattr(self, "_stage_streams", None)
            if pool is None or len(pool) < n_streams:
                pool = self._stage_streams = [torch.cuda.Stream(self.device) for _ in range(6)]
            handles = [s.cuda_stream for s in pool[:n_streams]]
        arr = (C.c_void_p * n_streams)(*handles)
        state = torch.empty((n_days, self.state_floats), dtype=torch.float32, device=self.device)
        gates = torch.empty((n_days, self.gate_floats), dtype=torch.float32, device=self.device)
        ws = self._ws(self.lib.vsv_context_prefix_ws_bytes(C.byref(self.dims), n_days, ctx_len, n_seg))
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_context_prefix(C.byref(self.dims), _ptr(self.packed), _ptr(vol), _ptr(ex), n_days, ctx_len,
                                             n_seg, arr, n_streams, _ptr(state), _ptr(gates),
                                             _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_context_prefix")
        # (ws is used on the worker streams, which the call joined back into the current stream: freeing it to
        # the current stream's pool after this point is ordered behind that join)
        return state, gates

    def encode_both_windows(self, vol, ex, n_windows, T, T_ctx=None):
        """Both encoders over overlapping windows (posterior encoder T steps, context encoder T_ctx), on two
        HIP streams (see encode_both_strided); falls back to the strided entry point beyond MAX_WINDOWS."""
        T_ctx = T if T_ctx is None else T_ctx
        if n_windows > self.MAX_WINDOWS:
            return self.encode_both_strided(vol, 25, 25, ex, self.X, self.X, n_windows, T, T_ctx)
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        side = self._side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ctx = self.encode_windows(1, vol, ex, n_windows, T_ctx)
        enc = self.encode_windows(0, vol, ex, n_windows, T)
        main.wait_stream(side)
        ctx.record_stream(main)
        return enc, ctx

    # -- decoder -----------------------------------------------------------
    @staticmethod
    def OUT_CHANNEL(c: int) -> int:
        """vsv_decode's out_mode for sample-minor output of decoder channel c alone (VSV_OUT_CHANNEL)."""
        return 2 + int(c)

    def decode(self, n_rows: int, T: int, t_out0: int, *,
               ctx: Optional[torch.Tensor] = None, ctx_div: int = 1,
               zb: Optional[torch.Tensor] = None, lv: Optional[torch.Tensor] = None, z_div: int = 1,
               mu0: float = 0.0, std0: float = 1.0,
               eps: Optional[torch.Tensor] = None, eps_t0: int = 0,
               seed: int = 0, S: int = 1, day0: int = 0, step0: int = 0,
               state_in: Optional[torch.Tensor] = None, state_div: int = 1,
               gate_in: Optional[torch.Tensor] = None, want_gates: bool = False,
               want_state: bool = False, want_z: bool = False, out_mode: int = 0):
        """See vsv_decode in include/vsv_hip.h.  ctx (n,ctx_T,L); zb/lv (n,z_T,L);
        eps (n_rows,eps_T,L).  out_mode 0: surf (n_rows, T', Q, 5, 5); 1: (n_rows/S, T', Q*25, S);
        OUT_CHANNEL(c) = 2 + c: (n_rows/S, T', 25, S), decoder channel c only (the others are not computed).
        Returns dict(surf, ex, state, z, gates)."""
        self._ensure_fold()
        din = _lib.VsvDecodeIn()
        if ctx is not None:
            ctx = _f32c(ctx, self.device)
            assert ctx.shape[-1] == self.Lc
            din.ctx, din.ctx_rs, din.ctx_T, din.ctx_div = ctx.data_ptr(), ctx.shape[1] * self.Lc, ctx.shape[1], ctx_div
        else:
            din.ctx_div = 1
        if zb is not None:
            zb = _f32c(zb, self.device)
            din.zb, din.z_rs, din.z_T, din.z_div = zb.data_ptr(), zb.shape[1] * self.L, zb.shape[1], z_div
            if lv is not None:
                lv = _f32c(lv, self.device)
                assert lv.shape == zb.shape
                din.lv = lv.data_ptr()
        else:
            din.z_div = 1
        din.mu0, din.std0 = float(mu0), float(std0)
        if eps is not None:
            eps = _f32c(eps, self.device)
            assert eps.shape[0] == n_rows and eps.shape[2] == self.L
            din.eps_in, din.eps_T = eps.data_ptr(), eps.shape[1]
        din.eps_t0 = eps_t0
        din.seed, din.S, din.day0, din.step0 = seed & (2 ** 64 - 1), S, day0, step0
        if state_in is not None:
            state_in = _f32c(state_in, self.device)
            din.state_in, din.state_div = state_in.data_ptr(), state_div
        else:
            din.state_div = 1
        dev = self.device
        if gate_in is not None:
            gate_in = _f32c(gate_in, dev)
            assert state_in is not None and gate_in.shape == (state_in.shape[0], self.gate_floats)
            din.gate_in = gate_in.data_ptr()
        gates = torch.empty((n_rows, self.gate_floats), dtype=torch.float32, device=dev) if want_gates else None
        if gates is not None:
            din.gate_out = gates.data_ptr()
        touts = T - t_out0
        state = torch.empty((n_rows, self.state_floats), dtype=torch.float32, device=dev) if want_state else None
        z = torch.empty((n_rows, T, self.L), dtype=torch.float32, device=dev) if want_z else None
        surf = ex = None
        if touts > 0:
            if out_mode == 0:
                surf = torch.empty((n_rows, touts, self.Q, 5, 5), dtype=torch.float32, device=dev)
            elif out_mode == 1:
                surf = torch.empty((n_rows // S, touts, self.Q * 25, S), dtype=torch.float32, device=dev)
            else:
                surf = torch.empty((n_rows // S, touts, 25, S), dtype=torch.float32, device=dev)
            if self.X > 0:
                ex = torch.empty((n_rows, touts, self.X), dtype=torch.float32, device=dev)
        ws = self._ws(self.lib.vsv_decode_ws_bytes(C.byref(self.dims), n_rows, T, t_out0))
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_decode(C.byref(self.dims), _ptr(self.packed), C.byref(din), n_rows, T, t_out0,
                                     _ptr(state), _ptr(z), _ptr(surf), out_mode, _ptr(ex),
                                     _ptr(ws), ws.numel(), self._stream())
        _lib.check(rc, "vsv_decode")
        return {"surf": surf, "ex": ex, "state": state, "z": z, "gates": gates}

    # -- reduction / noise ---------------------------------------------------
    def percentile(self, samples: torch.Tensor, qs: Sequence[float] = (5, 50, 95), inner: int = 25,
                   channel: Optional[int] = None, n_channels: int = 1) -> torch.Tensor:
        """samples (..., inner-multiple, S) sample-minor -> (n_groups, nq, inner).
        With channel/n_channels the tensor is (..., n_channels*inner, S) and only the `inner` vectors of
        that channel are reduced per group, read in place through the group stride (no gather copy)."""
        samples = _f32c(samples, self.device)
        S = samples.shape[-1]
        n_all = samples.numel() // S
        if channel is None:
            n_vec, stride, off = n_all, inner, 0
        else:
            assert 0 <= channel < n_channels and n_all % (n_channels * inner) == 0
            n_vec, stride, off = n_all // n_channels, n_channels * inner, channel * inner * S
        q = (C.c_double * len(qs))(*[float(x) for x in qs])
        out = torch.empty((n_vec // inner, len(qs), inner), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_percentile(samples.data_ptr() + 4 * off, n_vec, S, q, len(qs), inner, stride,
                                         _ptr(out), self._stream())
        _lib.check(rc, "vsv_percentile")
        return out

    def reduce_async(self, samples: torch.Tensor, qs, inner: int, channel: Optional[int], n_channels: int,
                     overlap: bool = True):
        """percentile() on the side stream, ordered after the work queued so far on the current stream:
        the reduction of one chunk of MC samples (LDS / VALU work) overlaps the MFMA-bound decode of the
        next.  Call join_side() before reading the results on the current stream.
        overlap=False (the caller has no next chunk): on the current stream -- a fork and join across hardware
        queues costs ~20 us and there is nothing to run beside."""
        if not overlap:
            return self.percentile(samples, qs, inner, channel, n_channels)
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            out = self.percentile(samples, qs, inner, channel, n_channels)
        samples.record_stream(self._side)
        out.record_stream(main)
        return out

    def join_side(self):
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)

    def dropout_mask(self, p: float, seed: int, lstm_id: int, layer: int, n_rows: int, T: int, hidden: int):
        """The factors (0 or 1/(1-p)) the training kernels apply between LSTM layers: (n_rows, T, hidden)."""
        out = torch.empty((n_rows, T, hidden), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_dropout_mask(p, seed & (2 ** 64 - 1), lstm_id, layer, n_rows, T, hidden,
                                           _ptr(out), self._stream())
        _lib.check(rc, "vsv_dropout_mask")
        return out

    def gmm_sample(self, means: torch.Tensor, chol: torch.Tensor, weights: torch.Tensor, n: int, seed: int = 0,
                   comp: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None, want_comp: bool = False):
        """vsv_gmm_sample: n latents from the fitted mixture (means (K,L), lower Cholesky factors (K,L,L),
        weights (K)).  comp (n,) / eps (n,L) inject the component picks / standard normal draws; otherwise Philox
        keyed by seed.  -> z (n, L) float32 [, comp (n,) int32]."""
        K, L = means.shape
        m = _f32c(means, self.device)
        ch = _f32c(chol, self.device)
        cdf = torch.cumsum(weights.to(self.device, torch.float64) / weights.sum(), 0).to(torch.float32).contiguous()
        ci = None if comp is None else comp.to(self.device, torch.int32).contiguous()
        ei = None if eps is None else _f32c(eps, self.device)
        if ci is not None:
            assert ci.shape == (n,)
        if ei is not None:
            assert ei.shape == (n, L)
        z = torch.empty((n, L), dtype=torch.float32, device=self.device)
        co = torch.empty((n,), dtype=torch.int32, device=self.device) if want_comp else None
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_gmm_sample(_ptr(m), _ptr(ch), _ptr(cdf), K, L, n, _ptr(ci), _ptr(ei), seed & (2 ** 64 - 1),
                                         _ptr(z), _ptr(co), self._stream())
        _lib.check(rc, "vsv_gmm_sample")
        return (z, co) if want_comp else z

    def philox_normal(self, seed: int, day0: int, n_days: int, S: int, step0: int, n_steps: int) -> torch.Tensor:
        out = torch.empty((n_days * S, n_steps, self.L), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.vsv_philox_normal(seed & (2 ** 64 - 1), day0, n_days, S, step0, n_steps, self.L,
                                            _ptr(out), self._stream())
        _lib.check(rc, "vsv_philox_normal")
        return out
