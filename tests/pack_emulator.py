"""numpy model of the kernels' dataflow on the PACKED weight blob (test
infrastructure).  It follows vsv_kernels.hip statement by statement for one
wave (16 rows): MFMA 16x16x4 lane maps, the private LDS columns, the tile-B
hand-off between dense_chain and lstm_seq.  It lets the CPU suite validate
vsv_pack.cpp and every index map without a GPU; the GPU tests then only have
to confirm that the hardware's fragment layout is the documented one.
"""
import ctypes as C

import numpy as np

from vsv import lib as vlib
from vsv.engine import dims_from_config, pack_state_dict


class Sec:
    def __init__(self, rec):
        self.w_off, self.b_off, self.nq, self.nt, self.n_relu, self.n_out = [int(x) for x in rec]


class PackedModel:
    def __init__(self, cfg, W):
        self.dims = dims_from_config(cfg)
        self.blob = pack_state_dict(self.dims, W).astype(np.float64)
        L = vlib.load()
        n = L.vsv_debug_layout(C.byref(self.dims), None, 0)
        buf = (C.c_longlong * n)()
        assert L.vsv_debug_layout(C.byref(self.dims), buf, n) == n
        v = np.array(buf[:], dtype=np.int64)
        recs = v[: 6 * 22].reshape(22, 6)
        self.enc = []
        for w in range(2):
            base = w * 7
            self.enc.append(dict(conv=[Sec(recs[base + i]) for i in range(4)],
                                 lstm=[Sec(recs[base + 4]), Sec(recs[base + 5])], head=Sec(recs[base + 6])))
        self.dec = dict(lstm=[Sec(recs[14]), Sec(recs[15])], head=[Sec(recs[16 + i]) for i in range(6)])
        sc = v[6 * 22:]
        (self.enc[0]["kx"], self.enc[1]["kx"], self.enc[0]["nh"], self.enc[1]["nh"], self.E, self.ex_f0,
         self.dec["kx"], self.dec["n_head"], self.dec["ex_layer"], self.dec["ex_f0"],
         self.NTm, self.NTd, self.Hd) = [int(x) for x in sc]
        self.enc[1]["ex_f0"] = self.enc[0]["ex_f0"] = self.ex_f0
        self.L, self.X, self.NL = self.dims.latent_dim, self.dims.ex_feats_dim, self.dims.mem_layers
        self.n_conv = [self.dims.n_conv, self.dims.n_ctx_conv]

    # acc (nt,64,4) = bias + W . priv   (tile_gemm + load_bias)
    def gemm(self, sec, priv, acc=None, t0=0, ntc=None, q0=0, nq=None):
        nt_all = sec.nt
        ntc = nt_all - t0 if ntc is None else ntc
        nq = sec.nq - q0 if nq is None else nq
        Wp = self.blob[sec.w_off: sec.w_off + sec.nq * nt_all * 256].reshape(sec.nq, nt_all, 4, 16, 4)
        Wp = Wp[q0:q0 + nq, t0:t0 + ntc]
        if acc is None:
            acc = self.blob[sec.b_off: sec.b_off + nt_all * 256].reshape(nt_all, 64, 4)[t0:t0 + ntc].copy()
        P = priv[: 4 * nq].reshape(nq, 4, 4, 16)              # (q, j, kg, m)
        D = np.einsum("qtkij,qjkm->tim", Wp, P)               # (t, i, m)
        acc = acc + D.reshape(ntc, 4, 4, 16).transpose(0, 1, 3, 2).reshape(ntc, 64, 4)
        return acc


def _lane_gm():
    lane = np.arange(64)
    return lane >> 4, lane & 15


def emu_dense_chain(pm, secs, x_priv, n_in_ks, inj=None, inj_f0=0, aux_layer=-1, aux_f0=0, aux_n=0):
    """x_priv (ks,64) -> (final acc (nt,64,4), aux values dict).  Mirrors dense_chain_kernel."""
    g, m = _lane_gm()
    buf = np.zeros((32, 64))
    buf[: x_priv.shape[0]] = x_priv
    aux = None
    for li, sec in enumerate(secs):
        acc = pm.gemm(sec, buf)
        f = 16 * np.arange(sec.nt)[:, None, None] + 4 * g[None, :, None] + np.arange(4)[None, None, :]
        v = np.where(f < sec.n_relu, np.maximum(acc, 0.0), acc)
        last = li == len(secs) - 1
        if last and inj is not None:
            for k in range(inj.shape[1]):
                sel = f == inj_f0 + k
                v = np.where(sel, np.broadcast_to(inj[m, k][None, :, None], v.shape), v)
        if li == aux_layer:
            aux = np.zeros((16, aux_n))
            for k in range(aux_n):
                tt, rem = divmod(aux_f0 + k, 16)
                gg, r = divmod(rem, 4)
                lanes = np.where(g == gg)[0]
                aux[m[lanes], k] = v[tt, lanes, r]
        if last:
            return v, aux
        buf = np.zeros((32, 64))
        buf[: 4 * sec.nt] = v.transpose(0, 2, 1).reshape(4 * sec.nt, 64)   # obuf[(4t+r)] = v[t,:,r]
    raise AssertionError


def natural_from_acc(v, n):
    """acc (nt,64,4) with NAT16 out order -> (16 rows, n features)."""
    g, m = _lane_gm()
    out = np.zeros((16, 16 * v.shape[0]))
    for t in range(v.shape[0]):
        for r in range(4):
            out[m, 16 * t + 4 * g + r] = v[t, :, r]
    return out[:, :n]


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def emu_lstm_step(pm, secs, NT, NL, xpart, h, c, x_in_lds):
    """One time step of lstm_seq_kernel for one wave.  h: (NL*NT+4, 64) LDS columns,
    c: (NL, NT, 64); xpart (kx,64).  Updates h, c in place."""
    kx = xpart.shape[0]
    for l in range(NL):
        sec = secs[l]
        hn = np.zeros((NT, 64))
        for t0, ntc in ((0, min(NT, 16)), (16, NT - 16)):
            if ntc <= 0:
                continue
            if l == 0 and not x_in_lds:
                acc = pm.gemm(sec, xpart, None, t0, ntc, 0, kx // 4)
                acc = pm.gemm(sec, h, acc, t0, ntc, kx // 4, sec.nq - kx // 4)
            elif l == 0:
                acc = pm.gemm(sec, np.concatenate([xpart, h]), None, t0, ntc)
            else:
                acc = pm.gemm(sec, h, None, t0, ntc)
            ig, fg, gg, og = _sig(acc[..., 0]), _sig(acc[..., 1]), np.tanh(acc[..., 2]), _sig(acc[..., 3])
            c[l, t0:t0 + ntc] = fg * c[l, t0:t0 + ntc] + ig * gg
            hn[t0:t0 + ntc] = og * np.tanh(c[l, t0:t0 + ntc])
        h[l * NT:(l + 1) * NT] = hn


def emu_encode(pm, which, surface, ex):
    """surface (16,T,25), ex (16,T,X)|None -> head outputs (16,T,nh)."""
    e = pm.enc[which]
    g, m = _lane_gm()
    T = surface.shape[1]
    NT, NL = pm.NTm, pm.NL
    h = np.zeros((NL * NT + 4, 64))
    c = np.zeros((NL, NT, 64))
    convs = e["conv"][: pm.n_conv[which]]
    out = np.zeros((16, T, e["nh"]))
    for t in range(T):
        ks = 4 * convs[0].nq
        xp = np.zeros((ks, 64))
        for s in range(ks):
            f = 4 * s + g
            xp[s] = np.where(f < 25, surface[m, t, np.minimum(f, 24)], 0.0)
        v, _ = emu_dense_chain(pm, convs, xp, ks, inj=None if ex is None else ex[:, t], inj_f0=pm.ex_f0)
        xe = v.transpose(0, 2, 1).reshape(4 * convs[-1].nt, 64)          # tile-B tensor
        assert xe.shape[0] == e["kx"]
        emu_lstm_step(pm, e["lstm"], NT, NL, xe, h, c, x_in_lds=False)
        acc = pm.gemm(e["head"], h[(NL - 1) * NT:])
        out[:, t] = natural_from_acc(acc, e["nh"])
    return out


def emu_decode(pm, dec_in, state=None, t_out0=0):
    """dec_in (16,T,2L) -> surf (16,T-t_out0,Q*25), ex (16,T-t_out0,X)|None, final state."""
    g, m = _lane_gm()
    T = dec_in.shape[1]
    NT, NL = pm.NTd, pm.NL
    d = pm.dec
    h = np.zeros((NL * NT + 4, 64))
    c = np.zeros((NL, NT, 64))
    if state is not None:          # state (16, NL, 2, 4*NT)
        for l in range(NL):
            for s in range(NT):
                h[l * NT + s] = state[m, l, 0, 4 * s + g]
                c[l, s] = state[m, l, 1, 4 * s + g]
    surf, exo = [], []
    kx = d["kx"]
    heads = d["head"][: d["n_head"]]
    for t in range(T):
        xp = np.zeros((kx, 64))
        for s in range(kx):
            f = 4 * s + g
            xp[s] = np.where(f < dec_in.shape[2], dec_in[m, t, np.minimum(f, dec_in.shape[2] - 1)], 0.0)
        emu_lstm_step(pm, d["lstm"], NT, NL, xp, h, c, x_in_lds=True)
        if t >= t_out0:
            htop = h[(NL - 1) * NT: NL * NT].copy()
            v, aux = emu_dense_chain(pm, heads, htop, NT, aux_layer=d["ex_layer"] if pm.X else -1,
                                     aux_f0=d["ex_f0"], aux_n=pm.X)
            surf.append(natural_from_acc(v, heads[-1].n_out))
            exo.append(aux)
    st = np.zeros((16, NL, 2, 4 * NT))
    for l in range(NL):
        for s in range(NT):
            st[m, l, 0, 4 * s + g] = h[l * NT + s]
            st[m, l, 1, 4 * s + g] = c[l, s]
    return (np.stack(surf, 1) if surf else None), (np.stack(exo, 1) if surf and pm.X else None), st
