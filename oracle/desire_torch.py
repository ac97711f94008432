"""TRAINING ORACLE (test infrastructure, CPU only): the frozen spec of DESIGN.md sections 2 and 8 restated in
PyTorch so that autograd provides the reference gradients the HIP backward is checked against.

Only tests/ (incl. the tests/fuzz_*.py sweeps) and bench.py's cpu_baseline leg (which times its float32, no-grad forward as
the batched CPU restatement) may import this module.  Forward values are pinned against oracle/desire_oracle.py (numpy) in
tests/test_train_oracle.py; gradients are then whatever autograd derives from that same graph.

Loss (the reference's `cost` is recon + kld with the id==0 masking rule, model/model.py:339-376; its recon term has
undefined inputs, so the paper's losses are used):

    L_sgm = mean_counted_a [ mean_k mean_{t present} ||Y_gt - Y0_k||  +  kld_a ]
    L_ioc = mean_counted_a [ CE(softmax_k(-max_{t present} ||Y_gt - Y0_k||) , softmax_k(score_k))  +  mean_k mean_{t present} ||Y_gt - (Y0_k + dY_k)|| ]
    L     = L_sgm + L_ioc        present(a,t): id != 0 in target frame t; counted: id != 0 at the last observed frame and some t present

with Y0 DETACHED inside the IOC module (sampled trajectories are inputs of the ranking/refinement module: positions
enter it only through non-differentiable cell/bin indices and the velocity embedding) -- the two modules share
gradients only through Hx (the IOC GRU's initial state).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import desire_oracle as O

DT = torch.float64


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=DT)


def leaf_weights(w: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: _t(v).clone().requires_grad_(True) for k, v in w.items()}


def gru_cell(x, h, Wg, bg, Wc, bc):
    H = h.shape[-1]
    g = torch.sigmoid(torch.cat([x, h], -1) @ Wg + bg)
    r, u = g[..., :H], g[..., H:]
    c = torch.tanh(torch.cat([x, r * h], -1) @ Wc + bc)
    return u * h + (1 - u) * c


def _gw(w, p):
    return w[p + "/gates/kernel"], w[p + "/gates/bias"], w[p + "/candidate/kernel"], w[p + "/candidate/bias"]


def gru_encode(seq, w, p):
    h = torch.zeros((seq.shape[1], w[p + "/candidate/kernel"].shape[1]), dtype=DT)
    for t in range(seq.shape[0]):
        h = gru_cell(seq[t], h, *_gw(w, p))
    return h


def conv2d_tf(x, wt, stride, padding):
    """NHWC in/out, HWIO weights, TF padding (oracle.conv2d)."""
    N, Hi, Wi, _ = x.shape
    kh, kw = wt.shape[0], wt.shape[1]
    xx = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        Ho, Wo = -(-Hi // stride), -(-Wi // stride)
        ph, pw = max((Ho - 1) * stride + kh - Hi, 0), max((Wo - 1) * stride + kw - Wi, 0)
        xx = F.pad(xx, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return F.conv2d(xx, wt.permute(3, 2, 0, 1), stride=stride).permute(0, 2, 3, 1)


def conv2d_transpose_tf(x, wt, stride, padding):
    """NHWC, weights [kh,kw,out,in] (oracle.conv2d_transpose)."""
    N, Hi, Wi, _ = x.shape
    kh, kw = wt.shape[0], wt.shape[1]
    full = F.conv_transpose2d(x.permute(0, 3, 1, 2), wt.permute(3, 2, 0, 1), stride=stride)
    Ho, Wo = O.deconv_out_size(Hi, kh, stride, padding), O.deconv_out_size(Wi, kw, stride, padding)
    pt = O._same_pad_before(Ho, kh, stride) if padding == "SAME" else 0
    pl = O._same_pad_before(Wo, kw, stride) if padding == "SAME" else 0
    return full[:, :, pt:pt + Ho, pl:pl + Wo].permute(0, 2, 3, 1)


def bn_frozen(x, w, p):
    return (x - w[p + "/bn/moving_mean"]) * (w[p + "/bn/gamma"] / torch.sqrt(w[p + "/bn/moving_var"] + O.BN_EPS)) + w[p + "/bn/beta"]


def bn(x, w, p, d):
    """Batch-norm of the CVAE stacks: frozen moving statistics (default) or, dims.bn_mode = 1, the reference graph's phase=train on a
    batch of ONE object = per-sample, per-channel moments over the layer's pixels (oracle.batch_norm "per_object"); gamma / beta
    are constants of the training spec either way."""
    if getattr(d, "bn_mode", 0) == 1:
        mean = x.mean(dim=(1, 2), keepdim=True)
        var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
        return (x - mean) * (w[p + "/bn/gamma"] / torch.sqrt(var + O.BN_EPS)) + w[p + "/bn/beta"]
    if getattr(d, "bn_mode", 0) == 2:      # whole-batch phase=train statistics (model/model.py:453,459-461,471 with the objects batched):
        mean = x.mean(dim=(0, 1, 2), keepdim=True)                     # per channel over every sample and pixel of the call
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
        return (x - mean) * (w[p + "/bn/gamma"] / torch.sqrt(var + O.BN_EPS)) + w[p + "/bn/beta"]
    return bn_frozen(x, w, p)


def rows_from_agents(x, d):
    x = x.reshape((d.n_scenes, 1, d.mno) + tuple(x.shape[1:]))
    return x.expand((d.n_scenes, d.K, d.mno) + tuple(x.shape[3:])).reshape((d.R,) + tuple(x.shape[3:]))


def head_nll(past, fut, w: Dict[str, torch.Tensor], d):
    """oracle.head_nll (model/model.py:315-366,494-565 on the X encoder's observed steps) as a differentiable graph, in the log form
    the kernels use: -log N = z / (2 (1 - rho^2)) + log(2 pi sx sy sqrt(1 - rho^2)), clamped at -log(1e-20) (no gradient beyond, like
    the reference's max(pdf, 1e-20)).  Returns (mean over counted (object, frame) pairs, their number)."""
    pn = _t(O.normalise(past, d, np.float64))
    nxt = torch.cat([pn[1:], _t(O.normalise(fut[:1], d, np.float64))], 0)
    ids = np.concatenate([np.asarray(past)[:, :, 0], np.asarray(fut)[:1, :, 0]], 0)
    counted = torch.as_tensor((ids[:-1] != 0) & (ids[1:] != 0))
    h = torch.zeros((pn.shape[1], w["enc_x/candidate/kernel"].shape[1]), dtype=DT)
    tot = torch.zeros((), dtype=DT)
    for t in range(pn.shape[0]):
        h = gru_cell(pn[t], h, *_gw(w, "enc_x"))
        o = h @ w["gauss_head/w"] + w["gauss_head/b"]
        sx, sy, rho = torch.exp(o[:, 2]), torch.exp(o[:, 3]), torch.tanh(o[:, 4])
        nx, ny = (nxt[t, :, 0] - o[:, 0]) / sx, (nxt[t, :, 1] - o[:, 1]) / sy
        neg = 1 - rho * rho
        nll = (nx * nx + ny * ny - 2 * rho * nx * ny) / (2 * neg) + torch.log(2 * np.pi * sx * sy * torch.sqrt(neg))
        nll = torch.clamp(nll, max=-np.log(1e-20))
        tot = tot + (nll * counted[t].to(DT)).sum()
    n = int(counted.sum())
    return tot / max(n, 1), n


def forward_loss(past, fut, eps, grids, grid_of_scene, w: Dict[str, torch.Tensor], d, fixed=None, bin_tab=None, head_weight=0.0):
    """past/fut in the oracle layout [T, A, 3]; returns dict with every forward tensor + the loss terms.
    `fixed` = {"Yd": ..., "dmax": ...} pins the stop-gradient quantities (for finite-difference checks).  head_weight > 0 adds
    head_weight x head_nll (the reference's own loss of its Gaussian output layer) to the loss."""
    pn = _t(O.normalise(past, d, np.float64))
    fn = _t(O.normalise(fut, d, np.float64))
    valid = torch.as_tensor(past[d.T_obs - 1, :, 0] != 0)
    out = {}
    Hx = gru_encode(pn, w, "enc_x")
    Hy = gru_encode(fn, w, "enc_y")
    vae_in = torch.relu(torch.cat([Hx, Hy], -1) @ w["fc_c/w"] + w["fc_c/b"])
    x = vae_in.reshape(-1, 32, 32, 1)
    for name, stride, pad in (("conv1", 2, "SAME"), ("conv2", 2, "SAME"), ("conv3", 1, "VALID")):
        p = "vae_enc/" + name
        x = F.elu(bn(conv2d_tf(x, w[p + "/w"], stride, pad) + w[p + "/b"], w, p, d))
    params = x.reshape(x.shape[0], -1) @ w["vae_enc/fc/w"] + w["vae_enc/fc/b"]
    mu, ls = params[:, :d.L], params[:, d.L:]
    z = rows_from_agents(mu, d) + torch.sqrt(torch.exp(rows_from_agents(ls, d))) * _t(eps)
    x = z.reshape(-1, 1, 1, d.L)
    for name, stride, pad, act in (("deconv1", 1, "VALID", F.elu), ("deconv2", 1, "VALID", F.elu),
                                   ("deconv3", 2, "SAME", F.elu), ("deconv4", 2, "SAME", torch.sigmoid)):
        p = "vae_dec/" + name
        x = act(bn(conv2d_transpose_tf(x, w[p + "/w"], stride, pad) + w[p + "/b"], w, p, d))
    xhat = x.reshape(x.shape[0], -1)
    Hx_rows = rows_from_agents(Hx, d)
    beta = torch.softmax(torch.relu(xhat @ w["mask_fc/w"] + w["mask_fc/b"]), -1)
    xz = beta * Hx_rows
    p_last = rows_from_agents(pn[d.T_obs - 1], d)
    h = Hx_rows
    ys = []
    for _ in range(d.T_pred):
        h = gru_cell(xz, h, *_gw(w, "dec"))
        ys.append(p_last + h @ w["head/w"] + w["head/b"])
    Y0 = torch.stack(ys, 1)                                                    # [R, T, 2]
    out.update(Hx=Hx, Hy=Hy, vae_in=vae_in, z_mean=mu, z_log_sigma_sq=ls, z=z, xhat=xhat, xz=xz, Y0=Y0)

    # ---- IOC on detached trajectories ----
    Yd = Y0.detach() if fixed is None else _t(fixed["Yd"])
    valid_rows = rows_from_agents(valid, d).numpy()
    gidx = np.asarray(grid_of_scene)[np.repeat(np.arange(d.n_scenes), d.K * d.mno)]
    gr = _t(grids)
    pre_min = float("inf")                                   # smallest |relu input| met in the IOC module (kink distance)
    Ycur = Yd
    for _it in range(d.iters):
        # refinement pass: the positions are DETACHED where they enter the features (cells and bins are indices, the velocity
        # embedding follows the same rule as pass 1), the additive path Y_p = Y_{p-1} + dY_p keeps the gradient
        pos = Ycur.detach()
        h = Hx_rows
        score = torch.zeros(d.R, dtype=DT)
        prev = p_last
        x_steps = []                                         # [e_v | e_s | e_r] per step (what the kernels keep as ioc_sv_x)
        for t in range(d.T_pred):
            cur = pos[:, t]
            pre_v = (cur - prev) @ w["ioc/vel_fc/w"] + w["ioc/vel_fc/b"]
            e_v = torch.relu(pre_v)
            cy, cx = O.scene_cell(cur.numpy().astype(np.float32), d.Gh, d.Gw)
            e_s = gr[gidx, cy, cx]
            P = cur.numpy().astype(np.float32).reshape(d.n_scenes * d.K, d.mno, 2)
            bins = O.neighbor_bins(P, valid_rows.reshape(d.n_scenes * d.K, d.mno), d.nb_w, d.nb_h, d.grid_size, bin_tab)
            onehot = _t((bins[..., None] == np.arange(d.B)).astype(np.float64))     # [g, i, j, b]
            pooled = torch.einsum("gijb,gjh->gibh", onehot, h.reshape(d.n_scenes * d.K, d.mno, d.H)).reshape(d.R, d.B * d.H)
            pre_r = pooled @ w["ioc/social_fc/w"] + w["ioc/social_fc/b"]
            e_r = torch.relu(pre_r)
            vr = torch.as_tensor(valid_rows.reshape(-1))
            pre_min = min(pre_min, float(pre_r.detach().abs()[vr].min()), float(pre_v.detach().abs()[vr].min()))
            x_steps.append(torch.cat([e_v, e_s, e_r], -1).detach())
            h = gru_cell(torch.cat([e_v, e_s, e_r], -1), h, *_gw(w, "ioc"))
            score = score + (h @ w["ioc/score/w"][:, 0] + w["ioc/score/b"][0])
            prev = cur
        dY = (h @ w["ioc/reg/w"] + w["ioc/reg/b"]).reshape(d.R, d.T_pred, 2)
        Ycur = Ycur + dY
    Y = Ycur
    out.update(score=score, dY=dY, Y=Y, ioc_relu_margin=pre_min, ioc_x=torch.stack(x_steps, 1))      # ioc_x [R, T, E]

    # ---- losses ----
    # masking rule of model/model.py:351-366 per target frame: present[t, a] = id != 0 in future frame t; an object counts when
    # it exists at the last observed frame and in at least one target frame; absent frames carry no ground truth
    pres = torch.as_tensor(np.asarray(fut)[:, :, 0] != 0)                                   # [T, A]
    pm = pres.T.reshape(d.n_scenes, 1, d.mno, d.T_pred).to(DT)
    nf = pres.sum(0).to(DT)                                                                 # [A]
    nfc = torch.clamp(nf, min=1.0)
    v = (valid & (nf > 0)).to(DT)
    n_valid = torch.clamp(v.sum(), min=1.0)
    gt = fn.permute(1, 0, 2).reshape(d.n_scenes, 1, d.mno, d.T_pred, 2)
    e0 = torch.sqrt(((Y0.reshape(d.n_scenes, d.K, d.mno, d.T_pred, 2) - gt) ** 2).sum(-1) + 1e-300)   # [n,K,mno,T]
    recon = (e0 * pm).sum(dim=(1, 3)).reshape(d.A) / (d.K * nfc)
    kld = -0.5 * (1.0 + ls - mu ** 2 - torch.exp(ls)).sum(1)
    L_sgm = ((recon + kld) * v).sum() / n_valid
    dmax = (e0.detach() * pm).max(dim=3).values if fixed is None else _t(fixed["dmax"])   # [n,K,mno] over present frames
    Pt = torch.softmax(-dmax, dim=1)
    logQ = torch.log_softmax(score.reshape(d.n_scenes, d.K, d.mno), dim=1)
    ce = -(Pt * logQ).sum(1).reshape(d.A)
    e1 = torch.sqrt(((Y.reshape(d.n_scenes, d.K, d.mno, d.T_pred, 2) - gt) ** 2).sum(-1) + 1e-300)
    reg = (e1 * pm).sum(dim=(1, 3)).reshape(d.A) / (d.K * nfc)
    L_ioc = ((ce + reg) * v).sum() / n_valid
    out.update(recon=recon, kld=kld, ce=ce, reg=reg, L_sgm=L_sgm, L_ioc=L_ioc, loss=L_sgm + L_ioc, Yd=Yd, dmax=dmax)
    if head_weight:
        L_head, n_head = head_nll(past, fut, w, d)
        out.update(L_head=L_head, n_head=n_head, loss=out["loss"] + float(head_weight) * L_head)
    return out


def loss_and_grads(past, fut, eps, grids, grid_of_scene, w_np: Dict[str, np.ndarray], d, bin_tab=None, head_weight=0.0):
    w = leaf_weights(w_np)
    out = forward_loss(past, fut, eps, grids, grid_of_scene, w, d, bin_tab=bin_tab, head_weight=head_weight)
    out["loss"].backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in w.items()}
    vals = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
    return vals, grads


def adam_step(w, g, m, v, step, lr=0.005, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer defaults (model/model.py:394): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t * m / (sqrt(v) + eps)."""
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return w - lr_t * m / (np.sqrt(v) + eps), m, v
