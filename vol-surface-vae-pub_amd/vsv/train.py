"""Training on the HIP path: forward with saved activations, loss terms and
gradients, BPTT, and the assembly of parameter gradients.

Division of labour (DESIGN.md "Training"): the recurrent and element-wise work
-- LSTM forward/backward through time, conv/linear chains forward/backward,
losses -- runs in the hand-written kernels behind the C ABI
(vsv_train_encode / vsv_train_decode / vsv_loss_grad / vsv_dense_backward /
vsv_lstm_backward).  The matrix products dW = G^T . X over the tensors those
kernels leave in HBM run in one more kernel (vsv_weight_grads: every layer of a
step in one or two launches, written at the packed positions), and the packed
gradient is scattered onto the parameters through the same index maps that
define the weight packing, so Toeplitz-expanded conv weights, interleaved LSTM
gates and summed biases all come out as gradients of the original parameters.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Sequence, Dict, List, Optional

import numpy as np
import torch

from . import lib as _lib
from .engine import Engine, build_gather_maps


class _Sec:
    def __init__(self, rec):
        self.w_off, self.b_off, self.nq, self.nt, self.n_relu, self.n_out = [int(x) for x in rec]


def read_layout(dims) -> dict:
    """Blob geometry from vsv_debug_layout."""
    L = _lib.load()
    n = L.vsv_debug_layout(C.byref(dims), None, 0)
    buf = (C.c_longlong * n)()
    assert L.vsv_debug_layout(C.byref(dims), buf, n) == n
    v = np.array(buf[:], dtype=np.int64)
    MC = 8                                                     # kMaxDenseLayers
    nrec = 2 * (MC + 3) + 2 + MC
    recs = v[: 6 * nrec].reshape(nrec, 6)
    lay = {"enc": []}
    for w in range(2):
        base = w * (MC + 3)
        lay["enc"].append(dict(conv=[_Sec(recs[base + i]) for i in range(MC)],
                               lstm=[_Sec(recs[base + MC]), _Sec(recs[base + MC + 1])], head=_Sec(recs[base + MC + 2])))
    d0 = 2 * (MC + 3)
    lay["dec"] = dict(lstm=[_Sec(recs[d0]), _Sec(recs[d0 + 1])], head=[_Sec(recs[d0 + 2 + i]) for i in range(MC)])
    sc = [int(x) for x in v[6 * nrec: 6 * nrec + 16]]          # (the folded head section follows: engine.py)
    (lay["enc"][0]["kx"], lay["enc"][1]["kx"], lay["enc"][0]["nh"], lay["enc"][1]["nh"], lay["E"], lay["ex_f0"],
     lay["dec"]["kx"], lay["dec"]["n_head"], lay["dec"]["ex_layer"], lay["dec"]["ex_f0"],
     lay["NTm"], lay["NTd"], lay["Hd"], lay["enc"][0]["n_chain"], lay["enc"][1]["n_chain"], lay["NI"]) = sc
    return lay


# ---- index arithmetic of the fragment layout (vsv_layout.h) -------------------------------------

def _slot_feature(s, hh):
    """feature (row of a tile-ordered matrix) held at slot s, lane half hh (SLOT order)."""
    return 32 * (s >> 4) + 8 * ((s >> 2) & 3) + 4 * hh + (s & 3)


def _wpos(sec: _Sec, n, k):
    """Position in the forward blob of W[n][k]: n = 32*tile + row, k = 2*kstep + half."""
    return sec.w_off + (((k >> 3) * sec.nt + (n >> 5)) * 64 + ((k & 1) * 32 + (n & 31))) * 4 + ((k >> 1) & 3)


def _bpos(sec: _Sec, n):
    t, i = n >> 5, n & 31
    return sec.b_off + (t * 2 + ((i >> 2) & 1)) * 16 + 4 * (i >> 3) + (i & 3)


def transposed_map(sec: _Sec, ks0: int, nks: int, kind: str, ntT: int) -> np.ndarray:
    """Forward-blob positions (or -1) of the transposed fragments of `sec` restricted to
    its k-steps [ks0, ks0+nks): K' = all 16*nt forward output k-steps (slot order), output
    tile t', row i' -> forward input feature
      'slot': the feature at slot 16t' + 4(i'>>3) + (i'&3), half (i'>>2)&1   (previous layer's slot order)
      'unit': hidden unit 8*(4*(i'>>3) + t') + 2*(i'&3) + ((i'>>2)&1)          (LSTM backward's unit tiles)
      'nat' : feature 32t' + i'."""
    nqT = 4 * sec.nt
    q, t, l, j = np.meshgrid(np.arange(nqT), np.arange(ntT), np.arange(64), np.arange(4), indexing="ij")
    s = 4 * q + j
    hh, i = l >> 5, l & 31
    n = _slot_feature(s, hh)
    if kind == "slot":
        kloc = 2 * (16 * t + 4 * (i >> 3) + (i & 3)) + ((i >> 2) & 1)
    elif kind == "unit":
        kloc = 8 * (4 * (i >> 3) + t) + 2 * (i & 3) + ((i >> 2) & 1)
    else:
        kloc = 32 * t + i
    ok = kloc < 2 * nks
    k = 2 * ks0 + np.where(ok, kloc, 0)
    pos = _wpos(sec, n, k)
    return np.where(ok, pos, -1).reshape(-1).astype(np.int64)


class _BwdBlob:
    """Transposed fragments of every layer, gathered on device from the forward blob."""

    def __init__(self):
        self.maps: List[np.ndarray] = []
        self.off: Dict[str, tuple] = {}
        self.cur = 0

    def add(self, name, sec, ks0, nks, kind, ntT):
        m = transposed_map(sec, ks0, nks, kind, ntT)
        self.off[name] = (self.cur, 4 * sec.nt, ntT)
        self.maps.append(m)
        self.cur += m.size

    def index(self):
        return np.concatenate(self.maps)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class TrainEngine:
    """Forward + backward of the train_step loss for one model on one GPU."""

    _DUMP = 1 << 16

    def __init__(self, eng: Engine, named_shapes):
        self.eng = eng
        self.lib = eng.lib
        self.dev = eng.device
        self.dims = eng.dims
        self.lay = lay = read_layout(eng.dims)
        self.NL = eng.dims.mem_layers
        # dense chain in front of each encoder's LSTM: conv layers [+ interaction Linears]
        self.n_conv = [lay["enc"][0]["n_chain"], lay["enc"][1]["n_chain"]]
        self.L, self.X, self.Q = eng.L, eng.X, eng.Q
        self.Lc = eng.Lc
        self.Hm = eng.dims.mem_hidden
        NTm, NTd = lay["NTm"], lay["NTd"]
        # ---- backward blob
        bb = _BwdBlob()
        for w in range(2):
            e = lay["enc"][w]
            # conv 0 reads the 25 grid points in natural order: its transposed fragments give d(surface)
            bb.add(f"e{w}.conv0", e["conv"][0], 0, 4 * e["conv"][0].nq, "nat", 1)
            for li in range(1, self.n_conv[w]):
                bb.add(f"e{w}.conv{li}", e["conv"][li], 0, 4 * e["conv"][li].nq, "slot", e["conv"][li - 1].nt)
            kx = e["kx"]
            bb.add(f"e{w}.in0", e["lstm"][0], 0, kx, "slot", kx // 16)
            bb.add(f"e{w}.hh0", e["lstm"][0], kx, 4 * NTm, "unit", 4)
            if self.NL == 2:
                bb.add(f"e{w}.in1", e["lstm"][1], 0, 4 * NTm, "unit", 4)
                bb.add(f"e{w}.hh1", e["lstm"][1], 4 * NTm, 4 * NTm, "unit", 4)
            bb.add(f"e{w}.head", e["head"], 0, 4 * NTm, "unit", 4)
        d = lay["dec"]
        bb.add("d.in0", d["lstm"][0], 0, d["kx"], "nat", (eng.Lc + eng.L + 31) // 32)
        bb.add("d.hh0", d["lstm"][0], d["kx"], 4 * NTd, "unit", 4)
        if self.NL == 2:
            bb.add("d.in1", d["lstm"][1], 0, 4 * NTd, "unit", 4)
            bb.add("d.hh1", d["lstm"][1], 4 * NTd, 4 * NTd, "unit", 4)
        bb.add("d.head0", d["head"][0], 0, 4 * NTd, "unit", 4)
        for li in range(1, d["n_head"]):
            bb.add(f"d.head{li}", d["head"][li], 0, 4 * d["head"][li].nq, "slot", 4)
        self.bb = bb
        idx = bb.index()
        self._bwd_idx = torch.from_numpy(np.where(idx < 0, 0, idx + 1)).to(self.dev)      # 0 -> the zero slot
        self._bwd_cache = {}
        # ---- parameter scatter maps (same maps as the device-side packing)
        ia, ib = build_gather_maps(self.dims, list(named_shapes))
        # blob positions that hold no parameter (zero padding) are scattered over a dump region
        # behind the parameters instead of all landing on one word (atomic contention)
        n_par = int(sum(int(np.prod(s)) for _, s in named_shapes))
        dump = n_par + 1 + (np.arange(ia.size, dtype=np.int64) % self._DUMP)
        self._idx_a = torch.from_numpy(np.where(ia == 0, dump, ia)).to(self.dev)
        nz = np.nonzero(ib)[0]                                  # only the bias_hh entries
        self._idx_b_pos = torch.from_numpy(nz.astype(np.int64)).to(self.dev)
        self._idx_b = torch.from_numpy(ib[nz]).to(self.dev)
        self._sizes = [int(np.prod(s)) for _, s in named_shapes]
        self._shapes = [tuple(s) for _, s in named_shapes]
        # the same maps inverted to CSR rows per parameter element (columns = blob positions, ascending) for the
        # one-launch gather vsv_param_grads
        pa, pb = np.nonzero(ia)[0], nz
        par = np.concatenate([ia[pa], ib[pb]]) - 1
        pos = np.concatenate([pa, pb])
        order = np.lexsort((pos, par))
        counts = np.bincount(par, minlength=n_par)
        self._row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(self.dev)
        self._cols = torch.from_numpy(pos[order].astype(np.int32)).to(self.dev)
        self.last_partials = None

    # -- helpers -----------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _z(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.dev)

    def _e(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def _wT(self, blobT, name):
        off, nq, nt = self.bb.off[name]
        return blobT.data_ptr() + 4 * off, nq, nt

    def _acc_layer(self, c, sec, G, X, bias=True, X1=None, T=1):
        """Queue gp[blob positions of sec] = G^T . [X | X1 shifted one step] (and the bias row sums) for
        vsv_weight_grads (flushed by _wg_flush once every G has been written).  G (NI, so, 64) slot order,
        X (NI, ks, 64) in the layer's k order, X1 (NI, ks1, 64) the recurrent input's source (h of the same
        step; the kernel reads the previous step's).  Views are fine: pointer and instance stride are passed."""
        for t in (G, X) + (() if X1 is None else (X1,)):
            assert t.dim() == 3 and t.shape[2] == 64 and t.stride(2) == 1 and t.stride(1) == 64 and t.dtype == torch.float32
        c.wg.append((sec, G, X, X1, bool(bias), int(T)))

    def _wg_flush(self, c, gp):
        """One vsv_weight_grads launch per VSV_MAX_WGRAD queued layers, on the current stream."""
        q, c.wg = c.wg, []
        for o in range(0, len(q), _lib.VSV_MAX_WGRAD):
            a = _lib.VsvWgrad()
            a.gp = gp.data_ptr()
            part = q[o:o + _lib.VSV_MAX_WGRAD]
            a.n = len(part)
            for i, (sec, G, X, X1, bias, T) in enumerate(part):
                p = a.p[i]
                p.g, p.g_stride, p.n_a = G.data_ptr(), G.stride(0), 2 * G.shape[1]
                p.x[0], p.x_stride[0], p.n_b[0], p.shift[0] = X.data_ptr(), X.stride(0), 2 * X.shape[1], 0
                if X1 is not None:
                    p.x[1], p.x_stride[1], p.n_b[1], p.shift[1] = X1.data_ptr(), X1.stride(0), 2 * X1.shape[1], 1
                p.n_inst, p.T = G.shape[0], T
                p.w_off, p.b_off, p.nt, p.bias = sec.w_off, sec.b_off, sec.nt, int(bias)
            _lib.check(self.lib.vsv_weight_grads(C.byref(a), self._stream()), "vsv_weight_grads")
        # operands stay referenced until the step's context goes (a flush may run on a stream of its own)
        c.wg_keep = getattr(c, "wg_keep", []) + q

    # -- building blocks ----------------------------------------------------------------
    class _Ctx:
        """Per-call state shared by the forward / backward helpers."""

    def _begin(self, surf, ex, dropout_p, dropout_seed):
        c = TrainEngine._Ctx()
        c.surf = surf.to(self.dev, torch.float32).contiguous()
        c.B, c.T = c.surf.shape[:2]
        c.ex = ex.to(self.dev, torch.float32).contiguous() if self.X > 0 else None
        c.rt = (c.B + 31) // 32
        c.packed = self.eng.packed
        c.drop_p = float(dropout_p) if self.NL > 1 else 0.0
        # an int, or a 1-element int64 tensor on the device the kernels read at launch (HIP-graph replays)
        c.drop_seed_dev = dropout_seed if torch.is_tensor(dropout_seed) else None
        c.drop_seed = 0 if c.drop_seed_dev is not None else int(dropout_seed) & (2 ** 64 - 1)
        c.blobT = None
        c.wg = []
        return c

    def _set_drop(self, c, s, sv, NI, NT, drop_id):
        if c.drop_p > 0:
            sv["hd"] = self._e(NI, 4 * NT, 64)
            s.hd, s.drop_p, s.drop_seed, s.drop_id = sv["hd"].data_ptr(), c.drop_p, c.drop_seed, drop_id
            s.drop_seed_dev = _ptr(c.drop_seed_dev)

    def _enc_save(self, c, which, Tw, drop_id=None):
        """Buffers of one encoder stack's training forward over Tw days -> (out (B, Tw, nh), sv dict, VsvTrainSave)."""
        lay, NL, NTm = self.lay, self.NL, self.lay["NTm"]
        e = lay["enc"][which]
        NI = c.rt * Tw
        sv = dict(emb_acts=self._e(self.n_conv[which], NI, 64, 64), xe=self._e(NI, e["kx"], 64),
                  gates=self._e(NI, NL, 16 * NTm, 64), c=self._e(NI, NL, 4 * NTm, 64),
                  h=self._e(NI, NL, 4 * NTm, 64), hout=self._e(NI, 4 * NTm, 64))
        out = self._e(c.B, Tw, e["nh"])
        s = _lib.VsvTrainSave()
        for k, v in sv.items():
            setattr(s, k, v.data_ptr())
        self._set_drop(c, s, sv, NI, NTm, which if drop_id is None else drop_id)
        sv["drop_id"] = which if drop_id is None else drop_id
        return out, sv, s

    def _enc_fwd_pair(self, c, T0, T1):
        """vsv_train_encode_pair: posterior encoder over T0 days and context encoder over T1, on the current stream.
        -> (enc_out, sv_e, ctx_out, sv_c)."""
        out0, sv0, s0 = self._enc_save(c, 0, T0)
        out0 = out0.view(2, c.B, T0, self.L)                      # (the pair entry point writes z_mean / z_log_var planes)
        out1, sv1, s1 = self._enc_save(c, 1, T1)
        X = self.X
        rc = self.lib.vsv_train_encode_pair(C.byref(self.dims), c.packed.data_ptr(),
                                            c.surf.data_ptr(), c.T * 25, 25, _ptr(c.ex), c.T * X, X,
                                            c.B, T0, T1, out0.data_ptr(), out1.data_ptr(), C.byref(s0), C.byref(s1),
                                            self._stream())
        _lib.check(rc, "vsv_train_encode_pair")
        return out0, sv0, out1, sv1

    def _enc_fwd(self, c, which, Tw, stream, drop_id=None):
        """vsv_train_encode of the first Tw days: buffers come from the current stream's pool, the three
        launches go to `stream`.  -> (out (B, Tw, nh), saved activations)."""
        lay, NL, NTm = self.lay, self.NL, self.lay["NTm"]
        e = lay["enc"][which]
        NI = c.rt * Tw
        sv = dict(emb_acts=self._e(self.n_conv[which], NI, 64, 64), xe=self._e(NI, e["kx"], 64),
                  gates=self._e(NI, NL, 16 * NTm, 64), c=self._e(NI, NL, 4 * NTm, 64),
                  h=self._e(NI, NL, 4 * NTm, 64), hout=self._e(NI, 4 * NTm, 64))
        out = self._e(c.B, Tw, e["nh"])
        s = _lib.VsvTrainSave()
        for k, v in sv.items():
            setattr(s, k, v.data_ptr())
        self._set_drop(c, s, sv, NI, NTm, which if drop_id is None else drop_id)
        sv["drop_id"] = which if drop_id is None else drop_id
        X = self.X
        with torch.cuda.stream(stream):
            rc = self.lib.vsv_train_encode(C.byref(self.dims), c.packed.data_ptr(), which,
                                           c.surf.data_ptr(), c.T * 25, 25, _ptr(c.ex), c.T * X, X,
                                           c.B, Tw, out.data_ptr(), C.byref(s), self._stream())
        _lib.check(rc, "vsv_train_encode")
        return out, sv

    def _blobT(self, c):
        if c.blobT is None:
            # the backward kernels' transposed sections: a gather of the blob -- straight from the parameters when
            # the blob was packed from them (one launch), else through the blob
            if c.packed is self.eng.packed:
                c.blobT = self.eng.gather_from_params(self._bwd_idx, self._bwd_cache)
            if c.blobT is None:
                ext = torch.cat([self._z(1), c.packed])
                c.blobT = ext[self._bwd_idx]
        return c.blobT

    def _dense_bwd_args(self, c, layers, n_inst_T, in_mode, in_t, in_f, want_dx):
        a = _lib.VsvDenseBwd()
        a.n_layers = len(layers)
        for i, ly in enumerate(layers):
            for k, v in ly.items():
                setattr(a.layer[i], k, v)
        a.n_rows, a.n_rowtiles, a.T = c.B, c.rt, n_inst_T
        a.in_mode, a.in_, a.in_f = in_mode, in_t.data_ptr(), in_f
        dx = self._e(c.rt * n_inst_T, 64, 64) if want_dx else None
        a.dx_out = _ptr(dx)
        return a, dx

    def _dense_bwd(self, c, layers, n_inst_T, in_mode, in_t, in_f, want_dx):
        a, dx = self._dense_bwd_args(c, layers, n_inst_T, in_mode, in_t, in_f, want_dx)
        _lib.check(self.lib.vsv_dense_backward(C.byref(a), self._stream()), "vsv_dense_backward")
        return dx

    def _lstm_bwd(self, c, prefix, hidden, NT, sv, Tw, t_out0, dh_top, nt_x, drop_id, dx_tile=None, dx_rows=None, dx_f=0,
                  stages=None):
        """BPTT of one memory stack -> dgates (NI, NL, 16*NT, 64).  Two layers run as four launches: the
        recurrence of layer 1, d h_0 = W_ih(1)^T . dgates(1) for every step at once, the recurrence of
        layer 0, dx = W_ih(0)^T . dgates(0) for every step at once -- only the W_hh^T products stay on the
        time loop (VSV_BWD_FUSED=1: the single two-layer launch)."""
        NL = self.NL
        blobT = self._blobT(c)
        dg = self._e(c.rt * Tw, NL, 16 * NT, 64)
        st = self._stream()

        def rec_args(l0, layers, dh, t0, ntx):
            a = _lib.VsvLstmBwd()
            a.drop_p, a.drop_seed, a.drop_id = c.drop_p, c.drop_seed, drop_id
            a.drop_seed_dev = _ptr(c.drop_seed_dev)
            a.gru = self.dims.mem_type
            for l in range(layers):
                a.wT_hh[l] = self._wT(blobT, f"{prefix}.hh{l0 + l}")[0]
                a.wT_in[l] = self._wT(blobT, f"{prefix}.in{l0 + l}")[0]
            a.nt_x, a.n_rows, a.n_rowtiles, a.T, a.t_out0 = ntx, c.B, c.rt, Tw, t0
            a.sv_gates, a.sv_c, a.dh_top = sv["gates"].data_ptr(), sv["c"].data_ptr(), _ptr(dh)
            a.dgates = dg.data_ptr()
            if ntx > 0:
                a.dx_tile, a.dx_rows, a.dx_f = _ptr(dx_tile), _ptr(dx_rows), dx_f
            if layers != NL:
                a.sv_nl, a.sv_l0 = NL, l0
            return a

        def rec(l0, layers, dh, t0, ntx):
            # (8 * NT: the hidden width as built -- widths between the built tile counts run zero-padded)
            _lib.check(self.lib.vsv_lstm_backward(C.byref(rec_args(l0, layers, dh, t0, ntx)), 8 * NT, layers, st),
                       "vsv_lstm_backward")

        def off_args(name, l, out_tile, out_rows, out_f, drop_layer):
            w, nq, nt = self._wT(blobT, name)
            g = _lib.VsvTilebGemm()
            g.wT, g.nqT, g.ntT = w, nq, nt
            g.g, g.g_stride = dg.data_ptr() + 4 * l * 16 * NT * 64, NL * 16 * NT * 64
            g.n_rows, g.n_rowtiles, g.T = c.B, c.rt, Tw
            g.out_tile, g.out_rows, g.out_f = _ptr(out_tile), _ptr(out_rows), out_f
            if drop_layer is not None and c.drop_p > 0:
                g.drop_p, g.drop_seed, g.drop_seed_dev = c.drop_p, c.drop_seed, _ptr(c.drop_seed_dev)
                g.drop_id, g.drop_layer = drop_id, drop_layer
            return g

        def off_loop(name, l, out_tile, out_rows, out_f, drop_layer):
            _lib.check(self.lib.vsv_tileb_gemm_launch(C.byref(off_args(name, l, out_tile, out_rows, out_f, drop_layer)), st),
                       "vsv_tileb_gemm_launch")

        if stages is not None:
            # the pair path: hand the four stages' argument builders to the caller, who launches each stage for two
            # stacks at once (_enc_bwd_pair)
            dh0 = self._e(c.rt * Tw, 64, 64)
            # (the launches come later, after BOTH stacks' arguments are built: the buffer must outlive this call, or
            # the allocator hands its memory to the other stack's buffers)
            c.__dict__.setdefault("bwd_hold", []).append(dh0)
            stages.extend([("rec", rec_args(1, 1, dh_top, t_out0, 0)), ("off", off_args(f"{prefix}.in1", 1, dh0, None, 0, 0)),
                           ("rec", rec_args(0, 1, dh0, 0, 0)),
                           ("off", off_args(f"{prefix}.in0", 0, dx_tile, dx_rows, dx_f, None) if nt_x > 0 else None)])
            return dg
        if NL == 1 or os.environ.get("VSV_BWD_FUSED", "0") == "1":
            rec(0, NL, dh_top, t_out0, nt_x)
            return dg
        rec(1, 1, dh_top, t_out0, 0)
        dh0 = self._e(c.rt * Tw, 64, 64)
        off_loop(f"{prefix}.in1", 1, dh0, None, 0, 0)
        rec(0, 1, dh0, 0, 0)
        if nt_x > 0:
            off_loop(f"{prefix}.in0", 0, dx_tile, dx_rows, dx_f, None)
        return dg

    def _lstm_wgrads(self, c, gp, secs, sv, dg, Tw, NT, x_tile):
        """dW of the LSTM layers: X = [x | h_prev] (layer 0), [h0 | h1_prev] (layer 1)."""
        for l in range(self.NL):
            # layer 1 read layer 0's h AFTER dropout when it is on
            xin = x_tile if l == 0 else (sv["hd"] if c.drop_p > 0 else sv["h"][:, l - 1])
            self._acc_layer(c, secs[l], dg[:, l], xin, X1=sv["h"][:, l], T=Tw)

    def _enc_bwd(self, c, gp, which, sv, Tw, d_out, want_dsurf=False):
        """Backward of one encoder stack from d(out) (B, Tw, nh): head, LSTM (BPTT), conv embedding;
        accumulates the packed-blob gradient into gp.  want_dsurf: also return d(surface) (B, Tw, 25)."""
        lay, NTm = self.lay, self.lay["NTm"]
        blobT = self._blobT(c)
        e = lay["enc"][which]
        pf = f"e{which}"
        rt = c.rt
        gh = self._e(rt * Tw, 64, 64)
        w, nq, nt = self._wT(blobT, f"{pf}.head")
        dh = self._dense_bwd(c, [dict(wT=w, nqT=nq, ntT=nt, n_relu=0, g_out=gh.data_ptr())], Tw, 0, d_out,
                             d_out.shape[-1], True)
        self._acc_layer(c, e["head"], gh[:, : 16 * e["head"].nt], sv["hout"])
        kx = e["kx"]
        # (an embedding narrower than four tiles -- fewer than 4 channels in the last conv layer -- leaves the
        # top slots of the 64-slot instance unwritten: they must read as zero)
        dxt = self._e(rt * Tw, 64, 64) if kx == 64 else self._z(rt * Tw, 64, 64)
        dgl = self._lstm_bwd(c, pf, self.Hm, NTm, sv, Tw, 0, dh, kx // 16, sv["drop_id"], dx_tile=dxt)
        self._lstm_wgrads(c, gp, e["lstm"], sv, dgl, Tw, NTm, sv["xe"])
        nc = self.n_conv[which]
        # the chain's last output as the backward kernel reads activations: 64 slots per instance (the LSTM's
        # x operand is stored kx slots per instance)
        xe_act = sv["xe"]
        if kx < 64:
            xe_act = self._z(rt * Tw, 64, 64)
            xe_act[:, :kx] = sv["xe"]
        layers, gs = [], []
        for li in range(nc - 1, -1, -1):
            g = self._e(rt * Tw, 64, 64)
            gs.append((li, g))
            ly = dict(ntT=0, nqT=0, n_relu=e["conv"][li].n_relu, g_out=g.data_ptr(),
                      act_out=(xe_act if li == nc - 1 else sv["emb_acts"][li + 1]).data_ptr())
            if li > 0 or want_dsurf:
                ly["wT"], ly["nqT"], ly["ntT"] = self._wT(blobT, f"{pf}.conv{li}")
            layers.append(ly)
        dxs = self._dense_bwd(c, layers, Tw, 1, dxt, 0, want_dsurf)
        for li, g in gs:
            cs = e["conv"][li]
            self._acc_layer(c, cs, g[:, : 16 * cs.nt], sv["emb_acts"][li][:, : 4 * cs.nq])
        if not want_dsurf:
            return None
        # tile-B [inst][slot][half*32 + m] -> (B, Tw, 25): grid point f sits at slot 4*(f>>3) + (f&3), half (f>>2)&1
        f = torch.arange(25, device=self.dev)
        t = dxs.view(rt, Tw, 64, 2, 32)[:, :, 4 * (f >> 3) + (f & 3), (f >> 2) & 1]      # (rt, Tw, 25, 32)
        return t.permute(0, 3, 1, 2).reshape(rt * 32, Tw, 25)[: c.B]

    def _enc_bwd_pair(self, c, gp, specs):
        """_enc_bwd of BOTH encoder stacks, stage by stage: specs = [(which, sv, Tw, d_out), (which, sv, Tw, d_out)];
        every stage's launch carries both stacks (vsv_*_pair).  Same kernels, arguments and queued weight-gradient
        problems as two _enc_bwd calls."""
        lay, NTm, lib, st = self.lay, self.lay["NTm"], self.lib, self._stream()
        blobT = self._blobT(c)
        rt = c.rt
        heads, stages, convs, keep = [], [], [], []
        for which, sv, Tw, d_out in specs:
            e = lay["enc"][which]
            pf = f"e{which}"
            gh = self._e(rt * Tw, 64, 64)
            w, nq, nt = self._wT(blobT, f"{pf}.head")
            a, dh = self._dense_bwd_args(c, [dict(wT=w, nqT=nq, ntT=nt, n_relu=0, g_out=gh.data_ptr())], Tw, 0, d_out,
                                         d_out.shape[-1], True)
            heads.append(a)
            self._acc_layer(c, e["head"], gh[:, : 16 * e["head"].nt], sv["hout"])
            kx = e["kx"]
            dxt = self._e(rt * Tw, 64, 64) if kx == 64 else self._z(rt * Tw, 64, 64)
            stg = []
            dgl = self._lstm_bwd(c, pf, self.Hm, NTm, sv, Tw, 0, dh, kx // 16, sv["drop_id"], dx_tile=dxt, stages=stg)
            stages.append(stg)
            self._lstm_wgrads(c, gp, e["lstm"], sv, dgl, Tw, NTm, sv["xe"])
            nc = self.n_conv[which]
            xe_act = sv["xe"]
            if kx < 64:
                xe_act = self._z(rt * Tw, 64, 64)
                xe_act[:, :kx] = sv["xe"]
            layers, gs = [], []
            for li in range(nc - 1, -1, -1):
                g = self._e(rt * Tw, 64, 64)
                gs.append((li, g))
                ly = dict(ntT=0, nqT=0, n_relu=e["conv"][li].n_relu, g_out=g.data_ptr(),
                          act_out=(xe_act if li == nc - 1 else sv["emb_acts"][li + 1]).data_ptr())
                if li > 0:
                    ly["wT"], ly["nqT"], ly["ntT"] = self._wT(blobT, f"{pf}.conv{li}")
                layers.append(ly)
            ca, _ = self._dense_bwd_args(c, layers, Tw, 1, dxt, 0, False)
            convs.append(ca)
            for li, g in gs:
                cs = e["conv"][li]
                self._acc_layer(c, cs, g[:, : 16 * cs.nt], sv["emb_acts"][li][:, : 4 * cs.nq])
            keep.append((gh, dh, dxt, dgl, xe_act, gs))
        _lib.check(lib.vsv_dense_backward_pair(C.byref(heads[0]), C.byref(heads[1]), st), "vsv_dense_backward_pair")
        for (k0, a0), (k1, a1) in zip(stages[0], stages[1]):
            assert k0 == k1
            if a0 is None or a1 is None:
                for a in (a0, a1):
                    if a is not None:
                        _lib.check(lib.vsv_tileb_gemm_launch(C.byref(a), st), "vsv_tileb_gemm_launch")
            elif k0 == "rec":
                _lib.check(lib.vsv_lstm_backward_pair(C.byref(a0), C.byref(a1), 8 * NTm, 1, st), "vsv_lstm_backward_pair")
            else:
                _lib.check(lib.vsv_tileb_gemm_launch_pair(C.byref(a0), C.byref(a1), st), "vsv_tileb_gemm_launch_pair")
        _lib.check(lib.vsv_dense_backward_pair(C.byref(convs[0]), C.byref(convs[1]), st), "vsv_dense_backward_pair")
        c.bwd_keep = keep                                         # (buffers the launches above read or write)

    def encoder_pass(self, surf, ex, Tw, d_out_fn, which: int = 0, dropout_p: float = 0.0, dropout_seed: int = 0,
                     drop_id: int = 3, want_grads: bool = True):
        """One encoder stack alone (the conditional prior network runs in the posterior encoder's slot of
        its own weight blob): forward on the first Tw days with saved activations, d_out = d_out_fn(out),
        backward.  -> (out (B, Tw, nh), packed-blob gradient or None)."""
        with torch.cuda.device(self.dev):
            c = self._begin(surf, ex, dropout_p, dropout_seed)
            out, sv = self._enc_fwd(c, which, Tw, torch.cuda.current_stream(self.dev), drop_id=drop_id)
            if not want_grads:
                return out, None
            d_out = d_out_fn(out).to(torch.float32).contiguous()
            gp = torch.zeros(c.packed.numel(), dtype=torch.float32, device=self.dev)
            self._enc_bwd(c, gp, which, sv, Tw, d_out)
            self._wg_flush(c, gp)
        return out, gp

    # -- one forward/backward ----------------------------------------------------------
    def forward_saved(self, surf: torch.Tensor, ex: Optional[torch.Tensor], horizon: int,
                      eps: Optional[torch.Tensor] = None, seed: int = 0,
                      dropout_p: float = 0.0, dropout_seed: int = 0,
                      ctx_len: Optional[int] = None, ctx_mask: Optional[torch.Tensor] = None):
        """CVAEMemRand.forward (:1022-1074) keeping what the backward kernels need.  -> call state `c`
        with c.outputs = {"surf" (B,H,Q,5,5), "ex" (B,H,X) | None, "z_mean", "z_log_var", "z" (B,T,L)}.
        dropout_p > 0: nn.LSTM's inter-layer dropout (train mode) in all three memories, masks keyed
        by dropout_seed (vsv_hip.h, vsv_train_save).
        ctx_len / ctx_mask (B, ctx_len, 1): rows with DIFFERENT context lengths in one call (the horizons
        of train_step_multihorizon as row groups): the context encoder runs ctx_len >= T - horizon steps
        and a row's embedding is zeroed from its own context length on -- the zero padding of :1061-1062;
        the head is evaluated on the last `horizon` steps of every row."""
        eng, lib, lay = self.eng, self.lib, self.lay
        dev = self.dev
        L, X, Q, NL = self.L, self.X, self.Q, self.NL
        NTd = lay["NTd"]
        c = self._begin(surf, ex, dropout_p, dropout_seed)
        B, T, rt = c.B, c.T, c.rt
        c.horizon = horizon
        Cn = c.Cn = T - horizon
        Cc = c.Cc = Cn if ctx_len is None else int(ctx_len)          # context-encoder steps (>= Cn)
        assert Cn <= Cc < T
        c.ctx_mask = ctx_mask
        st = self._stream()
        main = torch.cuda.current_stream(dev)
        if eng._side is None:
            eng._side = torch.cuda.Stream(dev)
        side = eng._side
        with torch.cuda.device(dev):
            # the two encoders are independent.  Few row tiles (a batch of 64): one call whose launches carry both
            # stacks (vsv_train_encode_pair: no second stream, half the launches).  Otherwise: context encoder on
            # the side stream
            if 8 * rt <= 128 and self.NL == 2 and self.dims.mem_type == 0 and os.environ.get("VSV_ENC_PAIR", "1") != "0":
                enc_out, c.sv_e, ctx_out, c.sv_c = self._enc_fwd_pair(c, T, Cc)
                zm, zl = enc_out[0], enc_out[1]
            else:
                side.wait_stream(main)
                ctx_out, c.sv_c = self._enc_fwd(c, 1, Cc, side)
                enc_out, c.sv_e = self._enc_fwd(c, 0, T, main)
                main.wait_stream(side)
                zm, zl = enc_out[..., :L].contiguous(), enc_out[..., L:].contiguous()
            if ctx_mask is not None:
                ctx_out = ctx_out * ctx_mask
            # ---- decoder
            NI, NIo = rt * T, rt * horizon
            d = lay["dec"]
            sv_d = dict(xe=self._e(NI, d["kx"], 64), gates=self._e(NI, NL, 16 * NTd, 64),
                        c=self._e(NI, NL, 4 * NTd, 64), h=self._e(NI, NL, 4 * NTd, 64),
                        hout=self._e(NIo, 4 * NTd, 64), head_acts=self._e(d["n_head"], NIo, 64, 64))
            s = _lib.VsvTrainSave()
            for k, v in sv_d.items():
                setattr(s, k, v.data_ptr())
            self._set_drop(c, s, sv_d, NI, NTd, 2)
            din = _lib.VsvDecodeIn()
            din.ctx, din.ctx_rs, din.ctx_T, din.ctx_div = ctx_out.data_ptr(), Cc * self.Lc, Cc, 1
            din.zb, din.lv, din.z_rs, din.z_T, din.z_div = zm.data_ptr(), zl.data_ptr(), T * L, T, 1
            din.mu0, din.std0 = 0.0, 1.0
            if eps is not None:
                eps = eps.to(dev, torch.float32).contiguous()
                assert eps.shape == (B, T, L)
                din.eps_in, din.eps_T = eps.data_ptr(), T
            din.eps_t0, din.seed, din.S, din.day0, din.step0, din.state_div = 0, seed & (2 ** 64 - 1), 1, 0, 0, 1
            z = self._e(B, T, L)
            s_out = self._e(B, horizon, Q, 5, 5)
            e_out = self._e(B, horizon, X) if X > 0 else None
            rc = lib.vsv_train_decode(C.byref(self.dims), c.packed.data_ptr(), C.byref(din), B, T, Cn,
                                      z.data_ptr(), s_out.data_ptr(), _ptr(e_out), C.byref(s), st)
            _lib.check(rc, "vsv_train_decode")
        c.sv_d = sv_d
        c.ctx_out = ctx_out                                        # (referenced by the launch above)
        c.outputs = {"surf": s_out, "ex": e_out, "z_mean": zm, "z_log_var": zl, "z": z}
        return c

    def backward_saved(self, c, d_pred: torch.Tensor, d_ex: Optional[torch.Tensor], d_zm: Optional[torch.Tensor],
                       d_zl: Optional[torch.Tensor], d_z: Optional[torch.Tensor] = None,
                       want_input_grad: bool = False):
        """Backward of forward_saved from the gradients w.r.t. its five outputs (None = zero): decoder
        head, decoder BPTT, reparameterisation (:199), both encoder stacks.  -> gradient w.r.t. the
        PACKED blob; want_input_grad: c.outputs["d_surface"] (B,T,5,5) as well."""
        lay, dev = self.lay, self.dev
        L, X, Q = self.L, self.X, self.Q
        NTd = lay["NTd"]
        B, T, rt, Cn, horizon = c.B, c.T, c.rt, c.Cc, c.horizon
        sv_d, d = c.sv_d, lay["dec"]
        NIo = rt * horizon
        zm, z = c.outputs["z_mean"], c.outputs["z"]
        main = torch.cuda.current_stream(dev)
        side = self.eng._side
        with torch.cuda.device(dev):
            d_pred = d_pred.to(dev, torch.float32).reshape(B, horizon, Q * 25).contiguous()
            if X > 0:
                d_ex = self._z(B, horizon, X) if d_ex is None else d_ex.to(dev, torch.float32).contiguous()
            blobT = self._blobT(c)
            gp = torch.zeros(c.packed.numel(), dtype=torch.float32, device=dev)

            # decoder head
            hd = d["head"][: d["n_head"]]
            layers, gs = [], []
            for li in range(d["n_head"] - 1, -1, -1):
                g = self._e(NIo, 64, 64)
                gs.append((li, g))
                w, nq, nt = self._wT(blobT, f"d.head{li}")
                ly = dict(wT=w, nqT=nq, ntT=nt, n_relu=hd[li].n_relu, g_out=g.data_ptr())
                if li < d["n_head"] - 1 and hd[li].n_relu > 0:
                    ly["act_out"] = sv_d["head_acts"][li + 1].data_ptr()
                if X > 0 and li == d["ex_layer"]:
                    ly["add"], ly["add_f0"], ly["add_n"] = d_ex.data_ptr(), d["ex_f0"], X
                layers.append(ly)
            dh_top = self._dense_bwd(c, layers, horizon, 0, d_pred, Q * 25, True)
            for li, g in gs:
                self._acc_layer(c, hd[li], g[:, : 16 * hd[li].nt], sv_d["head_acts"][li][:, : 4 * hd[li].nq])
            # decoder LSTM
            Lc = self.Lc
            dx = self._e(B, T, Lc + L)
            dg = self._lstm_bwd(c, "d", lay["Hd"], NTd, sv_d, T, c.Cn, dh_top, (Lc + L + 31) // 32, 2, dx_rows=dx, dx_f=Lc + L)
            self._lstm_wgrads(c, gp, d["lstm"], sv_d, dg, T, NTd, sv_d["xe"])
            # the decoder's weight gradients (head + memory: everything queued so far) on a stream of their own,
            # beside the encoders' backward -- a batch of 64 leaves most CUs idle, and the launch would otherwise
            # sit at the end of the step's critical path
            # (a fork and join cost ~20 us of cross-stream ordering: only where the launch is longer than that --
            # c2's 400 instances: 2.35 -> 2.29 ms per step; c1's 22: 0.60 -> 0.63, so not there)
            wg_stream = None
            if rt * T >= 128 and os.environ.get("VSV_WG_FORK", "1") != "0":
                wg_stream = getattr(self.eng, "_third", None)
                if wg_stream is None:
                    wg_stream = self.eng._third = torch.cuda.Stream(dev)
                wg_stream.wait_stream(main)
                with torch.cuda.stream(wg_stream):
                    self._wg_flush(c, gp)
            # reparameterisation (:199) and the context embedding: one launch (vsv_reparam_backward)
            d_enc, d_ctx = self._e(B, T, 2 * L), self._e(B, Cn, Lc)
            f32 = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()
            d_zm, d_zl, d_z = f32(d_zm), f32(d_zl), f32(d_z)
            mask = None if c.ctx_mask is None else c.ctx_mask.reshape(B, Cn).contiguous()
            _lib.check(self.lib.vsv_reparam_backward(dx.data_ptr(), z.data_ptr(), zm.data_ptr(), _ptr(d_zm), _ptr(d_zl),
                                                     _ptr(d_z), _ptr(mask), B, T, Cn, L, Lc, d_enc.data_ptr(),
                                                     d_ctx.data_ptr(), self._stream()), "vsv_reparam_backward")
            if (not want_input_grad and 8 * rt <= 128 and self.NL == 2 and self.dims.mem_type == 0
                    and os.environ.get("VSV_ENC_PAIR", "1") != "0" and os.environ.get("VSV_BWD_FUSED", "0") != "1"):
                # few row tiles: the two stacks' launches pairwise as one launch each, no second stream
                self._enc_bwd_pair(c, gp, [(0, c.sv_e, T, d_enc), (1, c.sv_c, Cn, d_ctx)])
                ds_e = ds_c = None
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ds_c = self._enc_bwd(c, gp, 1, c.sv_c, Cn, d_ctx, want_input_grad)
                ds_e = self._enc_bwd(c, gp, 0, c.sv_e, T, d_enc, want_input_grad)
                main.wait_stream(side)
            self._wg_flush(c, gp)
            if wg_stream is not None:
                main.wait_stream(wg_stream)
            if want_input_grad:
                ds_e[:, :Cn] += ds_c
                c.outputs["d_surface"] = ds_e.view(B, T, 5, 5)
        return gp

    def loss_and_grads(self, surf: torch.Tensor, ex: Optional[torch.Tensor], horizon: int, loss_cfg: dict,
                       eps: Optional[torch.Tensor] = None, seed: int = 0, want_grads: bool = True,
                       dropout_p: float = 0.0, dropout_seed: int = 0, kl_fn=None,
                       extra_d_pred: Optional[torch.Tensor] = None, want_input_grad: bool = False):
        """surf (B,T,5,5) [, ex (B,T,X)] -> (losses dict of 0-d float64 tensors, gradient of
        `loss` w.r.t. the PACKED blob or None, outputs).  loss_cfg: quantiles, quantile_weights, ex_n,
        ex_l2, ex_weight, kl_weight.  dropout_*: see forward_saved.
        kl_fn(z_mean, z_log_var) -> (kl, d_zm, d_zl) replaces the N(0,1) KL term: kl is a 0-d tensor,
        d_* the gradient of kl_weight * kl w.r.t. the posterior parameters (B,T,L) -- the conditional
        prior's Gaussian-Gaussian KL (cvae_conditional_prior.py:65-115).
        extra_d_pred (B,H,Q,5,5): added to d(loss)/d(decoded surfaces) -- the gradient later steps of an
        autoregressive rollout send back through their context (vae/utils.py:526-527, no detach).
        want_input_grad: outputs["d_surface"] (B,T,5,5) = gradient w.r.t. the input surfaces (through both
        encoders; the targets inside the loss are data, as in the reference)."""
        lib, dev = self.lib, self.dev
        L, X, Q = self.L, self.X, self.Q
        c = self.forward_saved(surf, ex, horizon, eps, seed, dropout_p, dropout_seed)
        surf, ex = c.surf, c.ex
        B, T = c.B, c.T
        outputs = c.outputs
        s_out, e_out, zm, zl = outputs["surf"], outputs["ex"], outputs["z_mean"], outputs["z_log_var"]
        st = self._stream()
        with torch.cuda.device(dev):
            # ---- losses + output gradients
            klw = float(loss_cfg["kl_weight"])
            la = _lib.VsvLoss()
            la.pred, la.target = s_out.data_ptr(), surf.data_ptr()
            la.B, la.H, la.T, la.Q = B, horizon, T, Q
            qw = np.asarray(loss_cfg["quantile_weights"], dtype=np.float64)
            for i in range(Q):
                la.q[i] = float(loss_cfg["quantiles"][i])
                la.qw[i] = float(qw[i] / qw.sum())
            d_pred, d_zm, d_zl = self._e(B, horizon, Q * 25), self._e(B, T, L), self._e(B, T, L)
            d_ex = None
            if X > 0:
                d_ex = self._e(B, horizon, X)
                la.ex_pred, la.ex_target, la.d_ex = e_out.data_ptr(), ex.data_ptr(), d_ex.data_ptr()
                la.X, la.ex_n, la.ex_l2 = X, int(loss_cfg["ex_n"]), int(loss_cfg["ex_l2"])
            la.surf_l2 = int(bool(loss_cfg.get("surf_l2", False)))
            la.ex_weight, la.kl_weight = float(loss_cfg["ex_weight"]), (0.0 if kl_fn is not None else klw)
            wdev = loss_cfg.get("weights_dev")           # (graphed step: the two weights live in device memory)
            if wdev is not None:
                if kl_fn is not None:
                    raise ValueError("weights_dev and kl_fn are exclusive")
                la.weights_dev = wdev.data_ptr()
            la.zm, la.zl, la.L = zm.data_ptr(), zl.data_ptr(), L
            la.d_pred, la.d_zm, la.d_zl = d_pred.data_ptr(), d_zm.data_ptr(), d_zl.data_ptr()
            # loss = re_surface + ex_weight * re_ex_feats + kl_weight * kl etc.: five numbers = a 5 x 3 matrix times the
            # kernel's three sums.  Without a custom KL term the kernel's last workgroup does it (vsv_loss.mix) and puts
            # its accumulators back to zero: no fill and no matrix-vector launch per step.
            exw = float(loss_cfg["ex_weight"]) if X > 0 else 0.0
            M = loss_cfg.get("mix_dev")
            if M is None:
                M = self.__dict__.setdefault("_loss_mix", {}).get((exw, klw))
            if M is None:
                M = self._loss_mix[(exw, klw)] = torch.tensor([[1.0, exw, klw], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0],
                                                               [1.0, exw, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64, device=dev)
            in_kernel = kl_fn is None
            if in_kernel:
                acc = self.__dict__.get("_loss_acc")
                if acc is None:          # (zero before the first call, zero again after every call)
                    acc = self._loss_acc = (torch.zeros(3, dtype=torch.float64, device=dev),
                                            torch.zeros(1, dtype=torch.int32, device=dev))
                sums = acc[0]
                v = torch.empty(5, dtype=torch.float64, device=dev)
                la.mix, la.loss_vec, la.ticket = M.data_ptr(), v.data_ptr(), acc[1].data_ptr()
            else:
                sums = torch.zeros(3, dtype=torch.float64, device=dev)
            la.sums = sums.data_ptr()
            _lib.check(lib.vsv_loss_grad(C.byref(la), st), "vsv_loss_grad")
            kl = None if in_kernel else sums[2]
            if extra_d_pred is not None:
                d_pred = d_pred + extra_d_pred.to(dev, torch.float32).reshape(B, horizon, Q * 25)
            if kl_fn is not None:
                kl, g_zm, g_zl = kl_fn(zm, zl)
                kl = kl.to(torch.float64)
                d_zm = d_zm + g_zm
                d_zl = d_zl + g_zl
            if kl_fn is not None:
                v = torch.mv(M, torch.stack([sums[0], sums[1], kl]))
            losses = {"loss": v[0], "re_surface": v[1], "re_ex_feats": v[2], "reconstruction_loss": v[3], "kl_loss": v[4]}
            outputs["loss_vec"] = v                               # (the five numbers above as one float64 vector)
            if not want_grads:
                return losses, None, outputs
            gp = self.backward_saved(c, d_pred, d_ex, d_zm, d_zl, None, want_input_grad)
        return losses, gp, outputs

    def loss_and_grads_multihorizon(self, surf: torch.Tensor, ex: Optional[torch.Tensor], horizons: Sequence[int],
                                    weights: Sequence[float], loss_cfg: dict,
                                    eps: Optional[Sequence[torch.Tensor]] = None, seed: int = 0,
                                    want_grads: bool = True, dropout_p: float = 0.0, dropout_seed: int = 0):
        """train_step_multihorizon's accumulated loss (:1204-1255) in ONE forward/backward: the G horizons
        are G row groups of a G*B-row batch (row g*B + b = sample b predicting horizons[g] days ahead).  Every
        group is a literal forward of the reference for its horizon -- its own posterior noise (eps[g]
        (B,T,L) or Philox keyed by the row) and, with dropout, its own masks -- so the numbers are those of
        G separate passes, while each recurrent kernel is launched once with G times the row tiles (a step is
        bound by the launches' sequential time steps, not by their width).  The context encoder runs
        T - min(horizons) steps for all rows and a group's embedding is zeroed from its own context length
        on; the decoder head is evaluated on the last max(horizons) steps and the loss kernel reads each
        group's last horizons[g] of them, scaling the gradients by weights[g].
        -> ({"loss", "reconstruction_loss", "kl_loss": weighted sums, "horizon_losses": (G,)} float64 device
        tensors, gradient w.r.t. the packed blob or None)."""
        lib, dev = self.lib, self.dev
        L, X, Q = self.L, self.X, self.Q
        G = len(horizons)
        hs = [int(h) for h in horizons]
        hmax, hmin = max(hs), min(hs)
        with torch.cuda.device(dev):
            surf = surf.to(dev, torch.float32)
            B, T = surf.shape[:2]
            assert T - hmax >= 1
            surf_r = surf.reshape(B, T, 25).repeat(G, 1, 1)
            ex_r = ex.to(dev, torch.float32).repeat(G, 1, 1) if X > 0 else None
            Cc = T - hmin
            key = (tuple(hs), B, T)
            cache = self.__dict__.setdefault("_mh_masks", {})
            mask = cache.get(key)
            if mask is None:
                m = np.zeros((G, B, Cc, 1), dtype=np.float32)
                for g, h in enumerate(hs):
                    m[g, :, : T - h] = 1.0
                mask = cache[key] = torch.from_numpy(m.reshape(G * B, Cc, 1)).to(dev)
            eps_r = None
            if eps is not None:
                eps_r = torch.cat([e.to(dev, torch.float32).reshape(B, T, L) for e in eps], dim=0)
            c = self.forward_saved(surf_r, ex_r, hmax, eps_r, seed, dropout_p, dropout_seed, ctx_len=Cc, ctx_mask=mask)
            o = c.outputs
            s_out, e_out, zm, zl = o["surf"], o["ex"], o["z_mean"], o["z_log_var"]
            st = self._stream()
            BG = G * B
            d_pred, d_zm, d_zl = self._e(BG, hmax, Q * 25), self._e(BG, T, L), self._e(BG, T, L)
            d_ex = self._e(BG, hmax, X) if X > 0 else None
            sums = torch.zeros((G, 3), dtype=torch.float64, device=dev)
            qw = np.asarray(loss_cfg["quantile_weights"], dtype=np.float64)
            klw = float(loss_cfg["kl_weight"])
            for g, h in enumerate(hs):
                r0 = g * B
                la = _lib.VsvLoss()
                la.pred, la.target = s_out[r0:].data_ptr(), c.surf[r0:].data_ptr()
                la.B, la.H, la.T, la.Q, la.Hs = B, h, T, Q, hmax
                la.grad_scale = float(weights[g])
                for i in range(Q):
                    la.q[i] = float(loss_cfg["quantiles"][i])
                    la.qw[i] = float(qw[i] / qw.sum())
                if X > 0:
                    la.ex_pred, la.ex_target, la.d_ex = e_out[r0:].data_ptr(), c.ex[r0:].data_ptr(), d_ex[r0:].data_ptr()
                    la.X, la.ex_n, la.ex_l2 = X, int(loss_cfg["ex_n"]), int(loss_cfg["ex_l2"])
                la.ex_weight, la.kl_weight = float(loss_cfg["ex_weight"]), klw
                la.zm, la.zl, la.L = zm[r0:].data_ptr(), zl[r0:].data_ptr(), L
                la.d_pred, la.d_zm, la.d_zl = d_pred[r0:].data_ptr(), d_zm[r0:].data_ptr(), d_zl[r0:].data_ptr()
                la.sums = sums[g:].data_ptr()
                _lib.check(lib.vsv_loss_grad(C.byref(la), st), "vsv_loss_grad")
            wkey = tuple(float(x) for x in weights)
            wc = self.__dict__.setdefault("_mh_weights", {})
            w = wc.get(wkey)                                      # (cached: no host copy inside a graph capture)
            if w is None:
                w = wc[wkey] = torch.tensor(list(wkey), dtype=torch.float64, device=dev)
            recon = sums[:, 0] + float(loss_cfg["ex_weight"]) * sums[:, 1] if X > 0 else sums[:, 0]
            hl = recon + klw * sums[:, 2]
            losses = {"loss": (w * hl).sum(), "reconstruction_loss": (w * recon).sum(),
                      "kl_loss": (w * sums[:, 2]).sum(), "horizon_losses": hl}
            if not want_grads:
                return losses, None
            gp = self.backward_saved(c, d_pred, d_ex, d_zm, d_zl)
        return losses, gp

    def param_grads(self, gp: torch.Tensor) -> List[torch.Tensor]:
        """gradient w.r.t. the packed blob -> gradients of the parameters (spec order)."""
        n = sum(self._sizes)
        if os.environ.get("VSV_NATIVE_PACK", "1") != "0" and gp.dtype == torch.float32 and gp.is_contiguous():
            flat = torch.empty(n + 1, dtype=torch.float32, device=self.dev)
            partials = torch.empty((n + 255) // 256, dtype=torch.float32, device=self.dev)
            rc = self.lib.vsv_param_grads(gp.data_ptr(), self._row_ptr.data_ptr(), self._cols.data_ptr(), n,
                                          flat.data_ptr() + 4, partials.data_ptr(), self._stream())
            if rc != 0:
                raise RuntimeError("vsv_param_grads: " + self.lib.vsv_last_error().decode())
            self.last_partials = partials                        # sums of squares per 256 elements, for the clip
        else:
            flat = torch.zeros(n + 1 + self._DUMP, dtype=torch.float32, device=self.dev)
            flat.index_add_(0, self._idx_a, gp)
            flat.index_add_(0, self._idx_b, gp[self._idx_b_pos])
            self.last_partials = None
        self.last_flat = flat[1:1 + n]                           # the returned gradients are views of this
        out, o = [], 1
        for sz, sh in zip(self._sizes, self._shapes):
            out.append(flat[o:o + sz].view(sh))
            o += sz
        return out
