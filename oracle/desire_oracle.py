"""CPU ORACLE for the DESIRE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (desire_amd/) never does; it fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference's model path cannot be imported or run (no tensorflow /
prettytensor in this image and the graph does not construct -- SURVEY.md section 0, 8(c)), and
the reference holds no tests, golden vectors or fixtures for it.  This file is therefore a
restatement of (i) the lines of /root/reference/model/model.py that do define arithmetic, with
TF-1.3 / prettytensor operator semantics stated inline, and (ii) the paper-defined IOC block
frozen in DESIGN.md section 2.  It is pinned against independent formulas instead
(tests/test_oracle_*.py: torch.nn.functional conv/conv_transpose, scipy multivariate_normal,
hand-computed GRU step, closed-form KLD) and the only importable reference module
(utils/data_loader.py -> tests/golden/loader_*.npz).

All arithmetic is numpy float32 unless `dtype=np.float64` is passed (used to judge which of
two fp32 implementations is closer to exact).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

BN_EPS = 1e-3  # model/model.py:460,479


# ------------------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------------------
def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def elu(x):
    # tf.nn.elu (model/model.py:453,471): x if x > 0 else exp(x) - 1
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(x.dtype)


def relu(x):
    return np.maximum(x, 0).astype(x.dtype)


def softmax(x):
    # tf.nn.softmax over the last axis (model/model.py:276)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(x.dtype)


# ------------------------------------------------------------------------------------------
# TF-1.x GRUCell (tensorflow/contrib/rnn GRUCell as used at model/model.py:137,144):
#   [r,u] = sigmoid([x,h] @ Wg + bg)          (bias_start = 1.0)
#   c     = tanh([x, r*h] @ Wc + bc)          (reset applied BEFORE the candidate matmul)
#   h'    = u*h + (1-u)*c                     (u gates the OLD state)
# ------------------------------------------------------------------------------------------
def bf16_round(x):
    """Round-to-nearest-even to bfloat16, returned as float32 (the operand quantiser of the bf16 kernels)."""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).astype(np.uint32)
    return r.view(np.float32)


def gru_cell(x, h, Wg, bg, Wc, bc, q=None):
    """q: optional operand quantiser (bf16 kernels round matrix OPERANDS only; state and gate math stay fp32)."""
    H = h.shape[-1]
    q = q or (lambda v: v)
    g = sigmoid(np.concatenate([x, q(h)], -1) @ Wg + bg)
    r, u = g[..., :H], g[..., H:]
    c = np.tanh(np.concatenate([x, q(r * h)], -1) @ Wc + bc)
    return (u * h + (1 - u) * c).astype(h.dtype)


def _gru_w(w, prefix, dt):
    return (w[prefix + "/gates/kernel"].astype(dt), w[prefix + "/gates/bias"].astype(dt),
            w[prefix + "/candidate/kernel"].astype(dt), w[prefix + "/candidate/bias"].astype(dt))


def gru_encode(seq, w, prefix, dt=np.float32, q=None):
    """static_rnn over the sequence from a zero state -> final state
    (model/model.py:152-167,233-241).  seq [T, A, n_in].
    q = bf16_round restates k_encoder_bf16: the recurrent operands (h, r*h, the h-rows of the kernels) are rounded, the
    2-wide input contribution stays fp32."""
    Wg, bg, Wc, bc = _gru_w(w, prefix, dt)
    if q is not None:
        n_in = seq.shape[-1]
        Wg = np.concatenate([Wg[:n_in], q(Wg[n_in:])], 0)
        Wc = np.concatenate([Wc[:n_in], q(Wc[n_in:])], 0)
    h = np.zeros((seq.shape[1], Wc.shape[1]), dt)
    for t in range(seq.shape[0]):
        h = gru_cell(seq[t].astype(dt), h, Wg, bg, Wc, bc, q)
    return h


# ------------------------------------------------------------------------------------------
# conv / transposed conv, NHWC, TF padding rules
# ------------------------------------------------------------------------------------------
def _same_pad_before(n_in_fwd, k, s):
    """TF 'SAME' padding before the first element for a forward conv over n_in_fwd inputs."""
    n_out = -(-n_in_fwd // s)
    total = max((n_out - 1) * s + k - n_in_fwd, 0)
    return total // 2


def conv2d(x, w, stride, padding):
    """tf.nn.conv2d / prettytensor conv2d (model/model.py:484-486).  x [N,H,W,Ci], w [kh,kw,Ci,Co].
    out[o] = sum_k x[s*o + k - pad] * w[k]."""
    N, Hi, Wi, Ci = x.shape
    kh, kw, _, Co = w.shape
    if padding == "SAME":
        Ho, Wo = -(-Hi // stride), -(-Wi // stride)
        pt, pl = _same_pad_before(Hi, kh, stride), _same_pad_before(Wi, kw, stride)
    else:
        Ho, Wo = (Hi - kh) // stride + 1, (Wi - kw) // stride + 1
        pt = pl = 0
    xp = np.zeros((N, (Ho - 1) * stride + kh, (Wo - 1) * stride + kw, Ci), x.dtype)
    hh, ww = min(Hi, xp.shape[1] - pt), min(Wi, xp.shape[2] - pl)
    xp[:, pt:pt + hh, pl:pl + ww] = x[:, :hh, :ww]
    out = np.zeros((N, Ho, Wo, Co), x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride]
            out += patch @ w[ky, kx]
    return out


def deconv_out_size(n_in, k, s, padding):
    # utils/convolutional_vae_util.py:154-157
    return (n_in - 1) * s + k if padding == "VALID" else n_in * s


def conv2d_transpose(x, w, stride, padding):
    """tf.nn.conv2d_transpose as called by the deconv2d layer
    (utils/convolutional_vae_util.py:109-114).  x [N,Hi,Wi,Ci], w [kh,kw,Co,Ci].
    It is the gradient of conv2d w.r.t. its input: out[s*i + k - pad] += x[i] * w[k]
    (no kernel flip), pad = the forward conv's SAME padding over the OUTPUT extent."""
    N, Hi, Wi, Ci = x.shape
    kh, kw, Co, _ = w.shape
    Ho, Wo = deconv_out_size(Hi, kh, stride, padding), deconv_out_size(Wi, kw, stride, padding)
    pt = _same_pad_before(Ho, kh, stride) if padding == "SAME" else 0
    pl = _same_pad_before(Wo, kw, stride) if padding == "SAME" else 0
    full = np.zeros((N, (Hi - 1) * stride + kh, (Wi - 1) * stride + kw, Co), x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            full[:, ky:ky + (Hi - 1) * stride + 1:stride, kx:kx + (Wi - 1) * stride + 1:stride] += \
                x @ w[ky, kx].T
    return np.ascontiguousarray(full[:, pt:pt + Ho, pl:pl + Wo])


def batch_norm(x, w, prefix, bn_mode, dt):
    """prettytensor batch_normalize, scale_after_normalization=True, eps=1e-3
    (model/model.py:457-462,476-481).  'frozen' = inference phase (moving moments);
    'batch' = the reference's literal default phase=train (batch moments over N,H,W)."""
    gamma, beta = w[prefix + "/bn/gamma"].astype(dt), w[prefix + "/bn/beta"].astype(dt)
    if bn_mode == "per_object":                          # the reference graph: one object per conv call, batch of 1
        ax = tuple(range(1, x.ndim - 1))
        mean = x.mean(axis=ax, keepdims=True, dtype=dt)
        var = ((x - mean) ** 2).mean(axis=ax, keepdims=True, dtype=dt)
    elif bn_mode == "batch":
        ax = tuple(range(x.ndim - 1))
        mean = x.mean(axis=ax, dtype=dt)
        var = ((x - mean) ** 2).mean(axis=ax, dtype=dt)  # tf.nn.moments: biased
    else:
        mean, var = w[prefix + "/bn/moving_mean"].astype(dt), w[prefix + "/bn/moving_var"].astype(dt)
    return ((x - mean) * (gamma / np.sqrt(var + dt(BN_EPS))) + beta).astype(dt)


def vae_encoder(vae_in, w, L, bn_mode="frozen", dt=np.float32, q=None, return_layers=False):
    """model/model.py:471-492.  vae_in [A, 1024] -> (z_mean, z_log_sigma_sq) each [A, L].
    q = bf16_round restates k_conv_gather_bf16: conv2 / conv3 round their input activations and weights."""
    x = vae_in.astype(dt).reshape(-1, 32, 32, 1)
    layers = []
    for name, stride, pad in (("conv1", 2, "SAME"), ("conv2", 2, "SAME"), ("conv3", 1, "VALID")):
        p = "vae_enc/" + name
        if q is not None and name != "conv1":
            x = conv2d(q(x), q(w[p + "/w"].astype(dt)), stride, pad) + w[p + "/b"].astype(dt)
        else:
            x = conv2d(x, w[p + "/w"].astype(dt), stride, pad) + w[p + "/b"].astype(dt)
        x = elu(batch_norm(x, w, p, bn_mode, dt))
        layers.append(x)
    flat = x.reshape(x.shape[0], -1)                      # [A, 4*4*128] NHWC flatten
    params = flat @ w["vae_enc/fc/w"].astype(dt) + w["vae_enc/fc/b"].astype(dt)
    if return_layers:
        return params[:, :L], params[:, L:], layers
    return params[:, :L], params[:, L:]


def vae_decoder(z, w, bn_mode="frozen", dt=np.float32, return_layers=False, q=None, q_layers=("deconv2", "deconv3", "deconv4")):
    """model/model.py:453-469 (+ utils/convolutional_vae_util.py:27-135).  z [R, L] -> [R, 1024].
    All four deconvs sit inside defaults_scope(batch_normalize=True), so the last one is
    BN -> sigmoid.  q = bf16_round restates the bf16-operand kernels (k_deconv2_bf16, k_deconv34_bf16): for the layers
    in q_layers the input activations and the weights are rounded where they enter the contraction; accumulation and
    the BN/ELU/sigmoid epilogues are fp32."""
    x = z.astype(dt).reshape(-1, 1, 1, z.shape[-1])
    layers = []
    for name, stride, pad, act in (("deconv1", 1, "VALID", elu), ("deconv2", 1, "VALID", elu),
                                   ("deconv3", 2, "SAME", elu), ("deconv4", 2, "SAME", sigmoid)):
        p = "vae_dec/" + name
        if q is not None and name in q_layers:
            x = conv2d_transpose(q(x), q(w[p + "/w"].astype(dt)), stride, pad) + w[p + "/b"].astype(dt)
        else:
            x = conv2d_transpose(x, w[p + "/w"].astype(dt), stride, pad) + w[p + "/b"].astype(dt)
        x = act(batch_norm(x, w, p, bn_mode, dt))
        layers.append(x)
    out = x.reshape(x.shape[0], -1)
    return (out, layers) if return_layers else out


# ------------------------------------------------------------------------------------------
# integer paths (bit-exact contract).  Every float op below is a single IEEE fp32 operation
# in the order written; the HIP side uses __fmul_rn/__fsub_rn/__fdiv_rn to forbid contraction.
# ------------------------------------------------------------------------------------------
def scene_cell(pos, Gh, Gw):
    """pos [...,2] normalised fp32 -> (cy, cx) int32.  cy = clamp(floor(y*Gh), 0, Gh-1)."""
    p = pos.astype(np.float32)
    fy = np.floor(p[..., 1] * np.float32(Gh))
    fx = np.floor(p[..., 0] * np.float32(Gw))
    cy = np.clip(fy, 0, Gh - 1).astype(np.int32)
    cx = np.clip(fx, 0, Gw - 1).astype(np.int32)
    return cy, cx


def logpolar_table(r_min, r_max, G):
    """The constants of the log-polar layout: [0..G-1] squared ring radii (geometric between r_min and r_max), [8+2k],
    [9+2k] = (cos, sin) of sector boundary k.  The HIP library builds the same table with the C library's pow/cos/sin;
    parity tests feed the oracle the library's table (desire_get_bin_table) so that a last-bit difference between two
    libm's cannot move a bin edge."""
    tab = np.zeros(20, np.float32)
    for k in range(G):
        t = np.float32(float(r_min) * (float(r_max) / float(r_min)) ** ((k + 1) / G))
        tab[k] = t * t
        tab[8 + 2 * k] = np.float32(np.cos(2.0 * np.pi * k / G))
        tab[9 + 2 * k] = np.float32(np.sin(2.0 * np.pi * k / G))
    return tab


def neighbor_bins_logpolar(pos, valid, G, tab):
    """Log-polar social bins (the paper's layout): for centre i and other j, v = p_j - p_i, d2 = vx*vx + vy*vy (each a
    single fp32 operation); ring = #{k < G : d2 >= tab[k]}, dropped when ring == G (beyond r_max), when j == i or
    !valid[j]; sector = first k with cross(dir_k, v) >= 0 and cross(dir_{k+1}, v) < 0, cross(d, v) = d.x*v.y - d.y*v.x,
    else 0; bin = ring*G + sector.  Comparisons and single IEEE operations only."""
    p = pos.astype(np.float32)
    tab = np.asarray(tab, np.float32)
    xi, yi = p[..., :, None, 0], p[..., :, None, 1]
    xj, yj = p[..., None, :, 0], p[..., None, :, 1]
    dx, dy = (xj - xi).astype(np.float32), (yj - yi).astype(np.float32)
    d2 = ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32)
    ring = np.zeros(d2.shape, np.int32)
    for k in range(G):
        ring += (d2 >= tab[k]).astype(np.int32)
    sector = np.zeros(d2.shape, np.int32)
    found = np.zeros(d2.shape, bool)
    for k in range(G):
        k1 = 0 if k + 1 == G else k + 1
        c0 = ((tab[8 + 2 * k] * dy).astype(np.float32) - (tab[9 + 2 * k] * dx).astype(np.float32)).astype(np.float32)
        c1 = ((tab[8 + 2 * k1] * dy).astype(np.float32) - (tab[9 + 2 * k1] * dx).astype(np.float32)).astype(np.float32)
        hit = (~found) & (c0 >= 0) & (c1 < 0)
        sector = np.where(hit, k, sector)
        found |= hit
    M = p.shape[-2]
    inside = (ring < G) & ~np.eye(M, dtype=bool) & valid[..., None, :].astype(bool)
    return np.where(inside, ring * G + sector, -1).astype(np.int32)


def neighbor_bins(pos, valid, nb_w, nb_h, G, tab=None):
    """tab given: log-polar layout (neighbor_bins_logpolar).  Otherwise:
    Social-LSTM style rectangular neighbourhood grid (lineage of the reference's missing
    grid.getSequenceGridMask, train.py:21,156-157; flags train.py:68-72).
    pos [..., M, 2] fp32, valid [..., M] bool -> bins [..., M, M] int32 (-1 = not pooled).
    For centre i and other j:  low = p_i - nb/2 ; high = p_i + nb/2 ;
    j is dropped if j==i, !valid[j], x_j >= high_x, x_j < low_x, y_j >= high_y, y_j < low_y;
    cell_x = min(int(floor(((x_j - low_x)/nb_w) * G)), G-1), same for y; bin = cell_x + cell_y*G."""
    if tab is not None:
        return neighbor_bins_logpolar(pos, valid, G, tab)
    p = pos.astype(np.float32)
    f = np.float32
    hw, hh = f(nb_w) / f(2), f(nb_h) / f(2)
    xi, yi = p[..., :, None, 0], p[..., :, None, 1]
    xj, yj = p[..., None, :, 0], p[..., None, :, 1]
    lx, hx = xi - hw, xi + hw
    ly, hy = yi - hh, yi + hh
    inside = (xj < hx) & (xj >= lx) & (yj < hy) & (yj >= ly)
    M = p.shape[-2]
    inside &= ~np.eye(M, dtype=bool)
    inside &= valid[..., None, :].astype(bool)
    cx = np.floor(((xj - lx) / f(nb_w)) * f(G))
    cy = np.floor(((yj - ly) / f(nb_h)) * f(G))
    cx = np.clip(np.where(inside, cx, 0), 0, G - 1).astype(np.int32)
    cy = np.clip(np.where(inside, cy, 0), 0, G - 1).astype(np.int32)
    return np.where(inside, cx + cy * G, -1).astype(np.int32)


def bin_margin(pos, nb_w, nb_h, G, valid=None):
    """Smallest distance (normalised units) of any (valid centre, valid other) pair to a bin/window
    boundary -- tests use it to know when a 1e-6 coordinate difference cannot flip an index."""
    p = pos.astype(np.float64)
    dx = (p[..., None, :, 0] - p[..., :, None, 0] + nb_w / 2) / (nb_w / G)
    dy = (p[..., None, :, 1] - p[..., :, None, 1] + nb_h / 2) / (nb_h / G)
    M = p.shape[-2]
    use = np.broadcast_to(~np.eye(M, dtype=bool), dx.shape).copy()
    if valid is not None:
        v = np.asarray(valid, bool)
        use &= v[..., :, None] & v[..., None, :]
    near = (np.abs(dx - G / 2) <= G / 2 + 1) & (np.abs(dy - G / 2) <= G / 2 + 1)     # only pairs near the window
    use &= near
    if not use.any():
        return float("inf")
    mx = np.abs(dx - np.round(dx))[use].min() * (nb_w / G)
    my = np.abs(dy - np.round(dy))[use].min() * (nb_h / G)
    return float(min(mx, my))


# ------------------------------------------------------------------------------------------
# stages
# ------------------------------------------------------------------------------------------
def normalise(frames, d, dt=np.float32):
    """frames [T, A, 3] = (id, x_px, y_px) loader layout (utils/data_loader.py:212-229) ->
    positions [T, A, 2] normalised: one fp32 multiply per coordinate."""
    f = frames.astype(np.float32)
    return np.stack([f[..., 1] * np.float32(d.sx), f[..., 2] * np.float32(d.sy)], -1).astype(dt)


def rows_from_agents(x, d):
    """[A, ...] per-agent -> [R, ...] per-row with r = (scene*K + k)*mno + slot."""
    x = x.reshape((d.n_scenes, 1, d.mno) + x.shape[1:])
    x = np.broadcast_to(x, (d.n_scenes, d.K, d.mno) + x.shape[3:])
    return np.ascontiguousarray(x).reshape((d.R,) + x.shape[3:])


def decode(xz, h0, p_last, w, d, dt=np.float32, return_hidden=False, q=None):
    """GRU decoder (model/model.py:279-285): the SAME input x_z at every step, initial state
    Hx, own weights (scope hidden_states).  Output head = the reference's commented-out linear
    layer (:315-321) per step, added to the last observed position.  -> Yhat [R, T_pred, 2].
    q = bf16_round restates k_decoder_bf16: only the RECURRENT operands (h, r*h and the h-rows of the kernels) are
    rounded; the constant-input half and the head stay fp32."""
    Wg, bg, Wc, bc = _gru_w(w, "dec", dt)
    if q is not None:
        n_in = xz.shape[-1]
        Wg = np.concatenate([Wg[:n_in], q(Wg[n_in:])], 0)
        Wc = np.concatenate([Wc[:n_in], q(Wc[n_in:])], 0)
    Wo, bo = w["head/w"].astype(dt), w["head/b"].astype(dt)
    h = h0.astype(dt)
    ys, hs = [], []
    for _ in range(d.T_pred):
        h = gru_cell(xz.astype(dt), h, Wg, bg, Wc, bc, q)
        ys.append(p_last.astype(dt) + (h @ Wo + bo))
        hs.append(h)
    Y = np.stack(ys, 1).astype(dt)
    return (Y, np.stack(hs, 1)) if return_hidden else Y


def social_pool(pos_t, hprev, valid_rows, d, dt, bin_tab=None):
    """pos_t [R,2], hprev [R,H], valid_rows [R] -> pooled [R, B*H] (bin-major)."""
    G = d.grid_size
    P = pos_t.reshape(d.n_scenes * d.K, d.mno, 2)
    Hh = hprev.reshape(d.n_scenes * d.K, d.mno, d.H)
    V = valid_rows.reshape(d.n_scenes * d.K, d.mno)
    bins = neighbor_bins(P, V, d.nb_w, d.nb_h, G, bin_tab)             # [SK, M, M]
    onehot = (bins[..., None] == np.arange(d.B)).astype(dt)             # [SK, M, M, B]
    pooled = np.einsum("gijb,gjh->gibh", onehot, Hh.astype(dt))
    return pooled.reshape(d.R, d.B * d.H).astype(dt), bins


def ioc_pass(Y, Hx_rows, p_last, valid_rows, grids, grid_of_scene, w, d, dt=np.float32, q=None, bin_tab=None):
    """One IOC scoring + regression pass (paper section 3.3; absent in the reference,
    model/model.py:312-313).  Returns (score [R], dY [R,T_pred,2]).
    q = bf16_round restates the bf16-operand kernel (kernels_bf16.hip): weights, x_t, h, r*h and the pooled sums are
    rounded where they enter a matrix product; accumulation, state and gate math stay fp32."""
    qq = q or (lambda v: v)
    Wg, bg, Wc, bc = _gru_w(w, "ioc", dt)
    Wg, Wc = qq(Wg), qq(Wc)
    Wv, bv = w["ioc/vel_fc/w"].astype(dt), w["ioc/vel_fc/b"].astype(dt)
    Ws, bs = qq(w["ioc/social_fc/w"].astype(dt)), w["ioc/social_fc/b"].astype(dt)
    wsc, bsc = w["ioc/score/w"].astype(dt), w["ioc/score/b"].astype(dt)
    Wr, br = qq(w["ioc/reg/w"].astype(dt)), w["ioc/reg/b"].astype(dt)
    scene_of_row = np.repeat(np.arange(d.n_scenes), d.K * d.mno)
    gidx = np.asarray(grid_of_scene)[scene_of_row]
    h = Hx_rows.astype(dt)
    score = np.zeros(d.R, dt)
    prev = p_last.astype(dt)
    for t in range(d.T_pred):
        cur = Y[:, t].astype(dt)
        e_v = relu((cur - prev) @ Wv + bv)
        cy, cx = scene_cell(cur, d.Gh, d.Gw)
        e_s = grids[gidx, cy, cx].astype(dt)
        pooled, _ = social_pool(cur, qq(h), valid_rows, d, dt, bin_tab)
        e_r = relu(qq(pooled) @ Ws + bs)
        h = gru_cell(qq(np.concatenate([e_v, e_s, e_r], -1)), h, Wg, bg, Wc, bc, q)
        score = score + (h @ wsc[:, 0] + bsc[0])
        prev = cur
    dY = (qq(h) @ Wr + br).reshape(d.R, d.T_pred, 2)
    return score.astype(dt), dY.astype(dt)


def forward(past, fut, eps, grids, grid_of_scene, w, d, bn_mode="frozen", dt=np.float32,
            Y_override: Optional[np.ndarray] = None, ioc_q=None, bin_tab=None) -> Dict[str, np.ndarray]:
    """Whole hot path.  past [T_obs, A, 3], fut [T_pred, A, 3] (or None when d.posterior==0),
    eps [R, L] (row order r=(scene*K+k)*mno+slot), grids [n_grids, Gh, Gw, C],
    grid_of_scene [n_scenes] int.  Returns every intermediate the parity tests compare."""
    out: Dict[str, np.ndarray] = {}
    pn = normalise(past, d, dt)
    valid = past[d.T_obs - 1, :, 0] != 0                               # present at last obs frame
    out["Hx"] = Hx = gru_encode(pn, w, "enc_x", dt)
    p_last = pn[d.T_obs - 1]
    if d.posterior:
        fn = normalise(fut, d, dt)
        out["Hy"] = Hy = gru_encode(fn, w, "enc_y", dt)
        vae_in = relu(np.concatenate([Hx, Hy], -1) @ w["fc_c/w"].astype(dt) + w["fc_c/b"].astype(dt))
        out["vae_in"] = vae_in
        mu, logsig = vae_encoder(vae_in, w, d.L, bn_mode, dt)
        out["z_mean"], out["z_log_sigma_sq"] = mu, logsig
        # model/model.py:264: z = mu + sqrt(exp(log_sigma_sq)) * eps   (K draws per agent here)
        z = rows_from_agents(mu, d) + np.sqrt(np.exp(rows_from_agents(logsig, d))) * eps.astype(dt)
    else:
        z = eps.astype(dt)
    out["z"] = z = z.astype(dt)
    xhat, layers = vae_decoder(z, w, bn_mode, dt, return_layers=True)
    out["d1"], out["d2"], out["d3"] = (l.reshape(d.R, -1) for l in layers[:3])
    out["xhat"] = xhat
    Hx_rows = rows_from_agents(Hx, d)
    # model/model.py:271-280: beta = softmax(relu(xhat W + b)); x_z = beta * Hx
    beta = softmax(relu(xhat @ w["mask_fc/w"].astype(dt) + w["mask_fc/b"].astype(dt)))
    out["xz"] = xz = (beta * Hx_rows).astype(dt)
    p_last_rows = rows_from_agents(p_last, d)
    valid_rows = rows_from_agents(valid, d)
    out["Y0"] = Y = decode(xz, Hx_rows, p_last_rows, w, d, dt)
    if Y_override is not None:
        Y = Y_override.astype(dt)
    score = np.zeros(d.R, dt)
    for _ in range(d.iters):
        if bin_tab is None and getattr(d, "bin_mode", 0) == 1:
            bin_tab = logpolar_table(d.nb_h, d.nb_w, d.grid_size)
        score, dY = ioc_pass(Y, Hx_rows, p_last_rows, valid_rows, grids, grid_of_scene, w, d, dt, q=ioc_q, bin_tab=bin_tab)
        Y = (Y + dY).astype(dt)
    out["Y"] = Y
    out["score"] = score
    return out


def forward_ref_compat(input_data, target_data, eps, w, H=16, L=128, n_dec=7, dt=np.float32):
    """The reference graph AS WRITTEN (model/model.py:116-311), for the parts that define arithmetic, at its own
    dims: H = d_dim = 16, T = seq_length = 8 (H == 2T is REQUIRED by :286-289), 7 decoder steps (:280), ONE eps
    draw per object (:262-263), raw-pixel inputs (:216-231), target = input shifted one frame, batch-norm in
    train phase on a batch of one object (= moments over H,W of that object, :453,471), no output layer: each
    decoder output [H] is re-read as T points (x,y) (:286-289).  CPU plumbing only (BASELINE configs[0]);
    the HIP path implements the frozen spec of DESIGN.md section 2 instead.
    input_data / target_data [MNO, T, 3] object-major as the placeholders (:91-105); eps [MNO, L].
    Needs weights "temporal/w|b" and GRU/fc/CVAE weights sized for H (spec.init_weights(Dims-like)).
    Returns rho [MNO,200], Hx, Hy, output_states [MNO, 7, T, 2], feature_pooling [MNO, 7, T, 200]."""
    MNO, T, _ = input_data.shape
    if H != 2 * T:
        raise ValueError("the reference reinterprets each decoder output [H] as T (x,y) pairs: needs H == 2*T (model/model.py:286-289)")
    rho = temporal_conv(input_data[None].astype(np.float32), w["temporal/w"], w["temporal/b"])[0, :, 0, :]   # O1
    xin = input_data[:, :, 1:3].transpose(1, 0, 2).astype(dt)                 # [T, MNO, 2] raw pixels
    yin = target_data[:, :, 1:3].transpose(1, 0, 2).astype(dt)
    Hx = gru_encode(xin, w, "enc_x", dt)                                       # O2
    Hy = gru_encode(yin, w, "enc_y", dt)                                       # O3
    vae_in = relu(np.concatenate([Hx, Hy], -1) @ w["fc_c/w"].astype(dt) + w["fc_c/b"].astype(dt))   # O4
    mu, lss = vae_encoder(vae_in, w, L, "per_object", dt)                      # O5
    z = mu + np.sqrt(np.exp(lss)) * eps.astype(dt)                             # O6
    xhat = vae_decoder(z, w, "per_object", dt)                                 # O7
    beta = softmax(relu(xhat @ w["mask_fc/w"].astype(dt) + w["mask_fc/b"].astype(dt)))   # O8
    Wg, bg, Wc, bc = _gru_w(w, "dec", dt)
    x_in, h = beta * Hx, Hx
    outs = []
    for _ in range(n_dec):                                                     # O9: same input every step
        h = gru_cell(x_in, h, Wg, bg, Wc, bc)
        outs.append(h)
    states = np.stack(outs, 1).reshape(MNO, n_dec, T, 2)                       # O10: [H] -> T x (x,y)
    fp = np.concatenate([states[..., 0:1] * rho[:, None, None, :100],          # O11
                         states[..., 1:2] * rho[:, None, None, 100:]], -1)
    return {"rho": rho, "Hx": Hx, "Hy": Hy, "z": z, "xhat": xhat, "output_states": states, "feature_pooling": fp}


# ------------------------------------------------------------------------------------------
# losses / utilities the reference defines (kept as utilities)
# ------------------------------------------------------------------------------------------
def kld_loss(z_mean, z_log_sigma_sq):
    """model/model.py:587-591."""
    lat = -0.5 * np.sum(1.0 + z_log_sigma_sq - np.square(z_mean) - np.exp(z_log_sigma_sq), axis=1)
    return lat.mean()


def normal_2d_pdf(x, y, mux, muy, sx, sy, rho):
    """model/model.py:494-523 (Graves 2013 eq. 24-25)."""
    nx, ny = x - mux, y - muy
    sxsy = sx * sy
    z = np.square(nx / sx) + np.square(ny / sy) - 2 * (rho * nx * ny) / sxsy
    neg = 1 - np.square(rho)
    return np.exp(-z / (2 * neg)) / (2 * np.pi * sxsy * np.sqrt(neg))


def reconstr_loss(mux, muy, sx, sy, rho, x, y):
    """model/model.py:525-550."""
    return np.sum(-np.log(np.maximum(normal_2d_pdf(x, y, mux, muy, sx, sy, rho), 1e-20)))


def head_nll(past, fut, w, d, dt=np.float64):
    """The reference's loss for its 5-wide output layer (model/model.py:315-366), teacher-forced over the observed frames of this
    spec's X encoder: for every observed frame t the state h_t -> output layer (`gauss_head/w|b` = output_w / output_b, :315-321) ->
    get_coef (:552-565) -> -log(max(N(next position), 1e-20)) (:494-550) against the position in frame t + 1 (the loader's target =
    the input shifted one frame, utils/data_loader.py:206-207; frame T_obs is the first future frame); an (object, frame) pair
    counts when the object exists in the frame AND in the next one (:351-366); mean over the counted pairs (:374-376).
    past [T_obs, A, 3], fut [T_pred, A, 3] loader layout -> (mean nll, number of counted pairs)."""
    pn = normalise(past, d, dt)
    nxt = np.concatenate([pn[1:], normalise(fut[:1], d, dt)], 0)                      # [T_obs, A, 2] targets
    ids = np.concatenate([past[:, :, 0], fut[:1, :, 0]], 0)                            # [T_obs + 1, A]
    counted = (ids[:-1] != 0) & (ids[1:] != 0)
    Wg, bg, Wc, bc = _gru_w(w, "enc_x", dt)
    h = np.zeros((pn.shape[1], Wc.shape[1]), dt)
    W5, b5 = w["gauss_head/w"].astype(dt), w["gauss_head/b"].astype(dt)
    tot = 0.0
    for t in range(pn.shape[0]):
        h = gru_cell(pn[t], h, Wg, bg, Wc, bc)
        mux, muy, sx, sy, rho = get_coef(h @ W5 + b5)
        m = counted[t]
        if m.any():
            tot += reconstr_loss(mux[m, 0], muy[m, 0], sx[m, 0], sy[m, 0], rho[m, 0], nxt[t, m, 0], nxt[t, m, 1])
    n = int(counted.sum())
    return tot / max(n, 1), n


def get_coef(out5):
    """model/model.py:552-565: split 5, exp on the std devs, tanh on the correlation."""
    mux, muy, sx, sy, corr = np.split(out5, 5, axis=-1)
    return mux, muy, np.exp(sx), np.exp(sy), np.tanh(corr)


def temporal_conv(temporal_data, w_t, b_t, stride=1):
    """O1, model/model.py:116-133: relu(depthwise_conv2d(X[...,0:2], W[1,T,2,100], VALID) + b).
    temporal_data [1, MNO, T, 3]; the slice starts at channel 0, so the two channels are
    (id, x) -- the reference's quirk, kept literally.  Output channel = c*100 + q."""
    x = temporal_data[..., :2].astype(np.float32)              # [1, MNO, T, 2]
    _, M, T, _ = x.shape
    kw = w_t.shape[1]
    n_out = (T - kw) // stride + 1
    out = np.zeros((1, M, n_out, 2 * w_t.shape[3]), np.float32)
    for o in range(n_out):
        win = x[0, :, o * stride:o * stride + kw, :]          # [M, kw, 2]
        out[0, :, o, :] = np.einsum("mkc,kcq->mcq", win, w_t[0]).reshape(M, -1)
    return relu(out + b_t)


def scene_cnn(image, w, dt=np.float32):
    """Scene-context CNN rho(I) (paper section 3.3; the reference has no image input at all,
    model/model.py:312-313).  image [n, 4Gh, 4Gw, 3] -> [n, Gh, Gw, C]:
    conv5x5/16/s2 SAME + ReLU -> conv5x5/32/s2 SAME + ReLU -> conv5x5/C/s1 SAME (linear)."""
    x = image.astype(dt)
    x = relu(conv2d(x, w["scene_cnn/conv1/w"].astype(dt), 2, "SAME") + w["scene_cnn/conv1/b"].astype(dt))
    x = relu(conv2d(x, w["scene_cnn/conv2/w"].astype(dt), 2, "SAME") + w["scene_cnn/conv2/b"].astype(dt))
    return (conv2d(x, w["scene_cnn/conv3/w"].astype(dt), 1, "SAME") + w["scene_cnn/conv3/b"].astype(dt)).astype(dt)


def feature_pooling(Y, rho, d):
    """O11, model/model.py:291-311: f[r,t] = concat(y_x * rho[agent,:100], y_y * rho[agent,100:]).
    Y [R,T,2], rho [A,200] -> [R,T,200]."""
    rr = rows_from_agents(rho.astype(np.float32), d)                     # [R,200]
    return np.concatenate([Y[..., 0:1] * rr[:, None, :100], Y[..., 1:2] * rr[:, None, 100:]], -1).astype(np.float32)


def losses(z_mean, z_log_sigma_sq, Y, fut_n, valid, d, dt=np.float32, present=None):
    """Train-path scalars.  kld[a]: model/model.py:587-589 per agent (the reference then means over
    its batch of 1).  recon[a]: the paper's sample-generation loss, mean over k and the PRESENT target frames of
    ||Y_gt - Yhat_k||_2 (the reference's Gaussian NLL has undefined inputs, :342).  cost: mean of
    (recon + kld) over the agents that count -- the reference's masking rule :351-366,374-376: the object exists
    (id != 0 at the last observed frame) and exists in the target; per target frame here: `present` [T, A] bool
    (id != 0 in that future frame; default all), a frame without the object is skipped, an object in no target
    frame does not count.  Y [R,T,2], fut_n [T,A,2] normalised, valid [A]."""
    kld = (-0.5 * np.sum(1.0 + z_log_sigma_sq - np.square(z_mean) - np.exp(z_log_sigma_sq), axis=1)).astype(dt)
    Yk = Y.reshape(d.n_scenes, d.K, d.mno, d.T_pred, 2).astype(dt)
    gt = fut_n.transpose(1, 0, 2).reshape(d.n_scenes, 1, d.mno, d.T_pred, 2).astype(dt)
    dist = np.sqrt(np.square(Yk - gt).sum(-1))                           # [n,K,mno,T]
    pm = np.ones((d.T_pred, d.A), bool) if present is None else np.asarray(present, bool)
    pm_ = pm.T.reshape(d.n_scenes, 1, d.mno, d.T_pred)
    nf = pm.sum(0).astype(dt)                                            # [A] present target frames
    recon = ((dist * pm_).sum(axis=(1, 3)).reshape(d.A) / (d.K * np.maximum(nf, 1))).astype(dt)
    v = np.asarray(valid, bool) & (nf > 0)
    n = int(v.sum())
    cost = float(((recon + kld) * v).sum() / max(n, 1))
    return kld, recon, cost, n


def gaussian_sample(params, normals):
    """sample() head (model/model.py:661-669): get_coef, then a draw from the bivariate normal at :608 in
    Cholesky form with caller-supplied N(0,1) normals, clipped to <= 1.0."""
    p = params.astype(np.float32)
    mux, muy = p[:, 0], p[:, 1]
    sx, sy, rho = np.exp(p[:, 2]), np.exp(p[:, 3]), np.tanh(p[:, 4])
    n0, n1 = normals[:, 0].astype(np.float32), normals[:, 1].astype(np.float32)
    x = mux + sx * n0
    y = muy + sy * (rho * n0 + np.sqrt(np.maximum(1 - rho * rho, 0)) * n1)
    return np.stack([np.minimum(x, 1.0), np.minimum(y, 1.0)], -1).astype(np.float32)


def rollout(past, w, d, normals, dt=np.float32):
    """sample()'s autoregressive path, model/model.py:623-688, restated on this spec's cells: warm-up = the X-encoder GRU over
    the observed frames carrying its state (:623-632); then for each of the `num` steps (:643): the 5-wide output layer on the
    state -> get_coef-style exp / exp / tanh (:661-663) -> one draw from the bivariate Gaussian (:665, Cholesky form with the
    caller's normals) -> clip to <= 1.0 (:666-669) -> the drawn position is the next input (:680-681).  Objects with id 0 are
    stepped like the others (the reference loops over every object, :660).  past [T_obs, A, 3] loader layout, normals
    [num, A, 2] -> positions [num, A, 2] in normalised units."""
    pn = normalise(past, d, dt)
    Wg, bg, Wc, bc = _gru_w(w, "enc_x", dt)
    h = np.zeros((pn.shape[1], Wc.shape[1]), dt)
    for t in range(pn.shape[0]):
        h = gru_cell(pn[t], h, Wg, bg, Wc, bc)
    W5, b5 = w["gauss_head/w"].astype(dt), w["gauss_head/b"].astype(dt)
    out = []
    for s in range(normals.shape[0]):
        pos = gaussian_sample(h @ W5 + b5, normals[s]).astype(dt)
        out.append(pos)
        h = gru_cell(pos, h, Wg, bg, Wc, bc)
    return np.stack(out, 0)


def ade_fde_k(Y, fut_n, d, present=None):
    """Y [R,T,2], fut_n [T,A,2] -> [A,4] = (ADE mean-of-K, FDE mean-of-K, ADE best-of-K, FDE best-of-K) over the target
    frames the object is present in (`present` [T, A] bool, default all; FDE at the LAST present frame); zeros for an
    object that is in no target frame."""
    Yk = Y.reshape(d.n_scenes, d.K, d.mno, d.T_pred, 2).astype(np.float32)
    gt = fut_n.transpose(1, 0, 2).reshape(d.n_scenes, 1, d.mno, d.T_pred, 2).astype(np.float32)
    e = np.sqrt(np.square(Yk - gt).sum(-1))                    # [n,K,mno,T]
    pm = np.ones((d.T_pred, d.A), bool) if present is None else np.asarray(present, bool)
    pm_ = pm.T.reshape(d.n_scenes, 1, d.mno, d.T_pred)
    nf = pm_.sum(-1)
    ade = (e * pm_).sum(-1) / np.maximum(nf, 1)
    last = np.where(pm_.any(-1), d.T_pred - 1 - np.argmax(pm_[..., ::-1], -1), 0)
    fde = np.take_along_axis(e, np.broadcast_to(last[..., None], e.shape[:3] + (1,)), -1)[..., 0] * (nf > 0)
    out = np.stack([ade.mean(1), fde.mean(1), ade.min(1), fde.min(1)], -1)
    return out.reshape(d.A, 4).astype(np.float32)


def ade_fde(Y, gt):
    """Y [..., T, 2], gt broadcastable -> (ADE, FDE) in the units of Y."""
    e = np.linalg.norm(Y - gt, axis=-1)
    return float(e.mean()), float(e[..., -1].mean())
