"""Present-row compaction (dims.flags = DESIRE_FLAG_COMPACT_ROWS, include/desire_hip.h; csrc/kernels_compact.hip).

The loader pads every window to max_num_obj slots (utils/data_loader.py:209-229) and the reference masks id-0 objects in the cost only
(model/model.py:351-366).  With the flag the per-row sample-generation stages run on the rows of present agents only.  Contract tested here:
  * rows of present agents: BIT-IDENTICAL to the uncompacted HIP path (every operand mode, both frozen and per-object batch-norm) and
    within 1e-3 of the CPU oracle (normalised coordinates) on the committed real-SDD goldens;
  * rows of absent agents: zeros in "Y0" (the sample-generation output);
  * edge cases: nothing present, everything present, a single agent, prior sampling (no posterior);
  * the flag is refused where it would change results (bn_mode = 2); the read-back path (training; "compact_host_counts") is refused under hipGraph
    capture, the default inference path -- device-side counts, round 6 -- is captured and replayed on changing data (end of this file)."""
import numpy as np
import pytest

from desire_amd.spec import FLAG_COMPACT_IOC, FLAG_COMPACT_ROWS, init_weights
from tests.helpers import make_case, small_dims
from tests.test_golden_e2e import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch


def ragged_case(d, seed, keep=0.35):
    """make_case windows with most slots absent, SDD-like: each slot is kept with probability `keep` (at least one window keeps none of
    its slots when n_scenes >= 3, one keeps all); a kept slot may still be absent from some TARGET frames."""
    past, fut, eps, grids, gos = make_case(d, seed=seed, n_absent=0)
    rng = np.random.default_rng(seed + 100)
    keep_m = rng.uniform(size=(d.n_scenes, d.mno)) < keep
    if d.n_scenes >= 3:
        keep_m[1] = False
        keep_m[2] = True
    past[~keep_m[:, None, :].repeat(d.T_obs, 1)] = 0
    fut[~keep_m[:, None, :].repeat(d.T_pred, 1)] = 0
    gone = rng.uniform(size=fut.shape[:3]) < 0.1                 # objects leaving the scene mid-target
    fut[gone] = 0
    # objects that were observed but have LEFT by the last observed frame: not `valid`, i.e. padding for every stage (and outside the compact maps)
    left = keep_m & (rng.uniform(size=keep_m.shape) < 0.15)
    if d.n_scenes >= 3:
        left[2] = False
    for sc, sl in zip(*np.nonzero(left)):
        past[sc, -2:, sl, :] = 0
        fut[sc, :, sl, :] = 0
    keep_m = keep_m & ~left
    return past, fut, eps, grids, gos, keep_m


def run(torch, d, w, past, fut, eps, grids, gos):
    from desire_amd import _lib
    h = _lib.Handle(d)
    h.set_weights(w)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.full((d.R, d.T_pred, 2), 7.0, device=dev)
    score = torch.zeros((d.R,), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    h.forward(past_t.data_ptr(), fut_t.data_ptr() if d.posterior else 0, eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), s)
    torch.cuda.synchronize()
    Y0 = h.read_buffer("Y0", (d.R, d.T_pred, 2))
    h.close()
    return Y0, Y.cpu().numpy(), score.cpu().numpy()


def row_mask(d, present):
    return np.repeat(present[:, None, :], d.K, axis=1).reshape(-1)


@pytest.mark.parametrize("kw", [
    dict(),                                          # fp32 operands
    dict(bf16=2),                                    # split operands: six-product sample generation, three-product IOC
    dict(bf16=3),
    dict(bf16=1),                                    # plain bf16 operands (fused deconv3+4)
    dict(bn_mode=1),                                 # per-object batch statistics (the reference graph's batch of one)
    dict(H=64, K=3, mno=16),
    dict(mno=8, K=5, n_scenes=6),
    dict(posterior=0),                               # z ~ N(0, I): no CVAE encoder behind the scan
    dict(mno=64, n_scenes=3, K=2),                   # groups larger than a tile (cluster-form IOC)
], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()) or "fp32")
def test_present_rows_are_bit_identical_and_absent_rows_zero(torch_cuda, kw):
    d = small_dims(**{**dict(n_scenes=4, K=4, T_pred=12), **kw})
    w = init_weights(d, 3)
    past, fut, eps, grids, gos, keep = ragged_case(d, seed=11)
    Y0a, Ya, sa = run(torch_cuda, d, w, past, fut, eps, grids, gos)
    Y0b, Yb, sb = run(torch_cuda, d.replace(flags=FLAG_COMPACT_ROWS), w, past, fut, eps, grids, gos)
    m = row_mask(d, keep)
    assert m.any() and (~m).any()
    np.testing.assert_array_equal(Y0b[m], Y0a[m])
    assert not Y0b[~m].any()
    assert np.isfinite(Yb).all() and np.isfinite(sb).all()
    # the IOC pass pools present agents only, so their refined rows do not depend on what the absent rows hold either
    np.testing.assert_array_equal(Yb[m], Ya[m])
    np.testing.assert_array_equal(sb[m], sa[m])


@pytest.mark.parametrize("which", ["none", "all", "one"])
def test_edge_counts(torch_cuda, which):
    d = small_dims(n_scenes=2, K=3, T_pred=8, mno=16)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=2, n_absent=0)
    keep = np.ones((d.n_scenes, d.mno), bool)
    if which == "none":
        keep[:] = False
    elif which == "one":
        keep[:] = False
        keep[1, 5] = True
    past[~keep[:, None, :].repeat(d.T_obs, 1)] = 0
    fut[~keep[:, None, :].repeat(d.T_pred, 1)] = 0
    Y0a, Ya, sa = run(torch_cuda, d, w, past, fut, eps, grids, gos)
    Y0b, Yb, sb = run(torch_cuda, d.replace(flags=FLAG_COMPACT_ROWS), w, past, fut, eps, grids, gos)
    m = row_mask(d, keep)
    np.testing.assert_array_equal(Y0b[m], Y0a[m])
    np.testing.assert_array_equal(Yb[m], Ya[m])
    assert not Y0b[~m].any()
    assert np.isfinite(Yb).all() and np.isfinite(sb).all()


@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_compacted_path_reproduces_the_sdd_goldens(torch_cuda, tag):
    """Real SDD bookstore windows (9 of 32 slots present at cfg1): the compacted path against the ORACLE's goldens, present rows."""
    d, g, eps, grids, gos, w = load_case(tag)
    Y0, Y, score = run(torch_cuda, d.replace(flags=FLAG_COMPACT_ROWS), w, g["past"], g["fut"], eps, grids, gos)
    present = g["past"][:, -1, :, 0] != 0
    m = row_mask(d, present)
    assert np.abs(Y0 - g["Y0"])[m].max() < 1e-3
    assert not Y0[~m].any()
    err = np.abs(Y - g["Y"]).reshape(d.R, -1).max(1)[m]
    flipped = err > 1e-3                                  # (the un-anchored statistic of test_golden_e2e.py: a 1e-7 move may cross a bin edge)
    assert flipped.mean() < 0.01
    assert err[~flipped].max() < 1e-3
    assert np.abs(score - g["score"])[m][~flipped].max() < 5e-3


def test_flag_is_refused_where_it_would_change_results(torch_cuda):
    from desire_amd import _lib
    d = small_dims(n_scenes=1, K=2, T_pred=8)
    with pytest.raises(_lib.DesireError, match="COMPACT_ROWS"):
        _lib.Handle(d.replace(flags=FLAG_COMPACT_ROWS, bn_mode=2))
    h = _lib.Handle(d)
    with pytest.raises(_lib.DesireError):
        h.set_option("flags", 64)
    h.set_option("flags", FLAG_COMPACT_ROWS)               # on a live handle
    h.set_option("compact_host_counts", 1)                 # the read-back path (what training always runs): it is the one that cannot be captured
    w = init_weights(d, 0)
    h.set_weights(w)
    torch = torch_cuda
    dev = torch.device("cuda")
    past, fut, eps, grids, gos = make_case(d, seed=1)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    with pytest.raises(_lib.DesireError, match="desire_encode comes first"):
        h.sample(eps_t.data_ptr(), Y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        h.encode(past_t.data_ptr(), fut_t.data_ptr(), side.cuda_stream)
        side.synchronize()
        h.graph_begin(side.cuda_stream)
        try:
            with pytest.raises(_lib.DesireError, match="not capturable"):
                h.sample(eps_t.data_ptr(), Y.data_ptr(), side.cuda_stream)
        finally:
            try:
                h.graph_end(side.cuda_stream)
            except _lib.DesireError:
                pass
    h.close()


# ---- training: the per-row stages' saves and their whole backward on the compact rows --------------------------------------------------
def _train_step(torch, d, w, past, fut, eps, grids, gos, min_rows=None):
    from desire_amd import _lib
    h = _lib.Handle(d)
    h.set_weights(w)
    if min_rows is not None:
        h.set_option("compact_min_rows", min_rows)
    h.set_training(True)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
    torch.cuda.synchronize()
    loss = h.train_loss(fut_t.data_ptr())
    grads = {k: h.get_grad(k, w[k].shape) for k in w if not k.startswith("gauss_head/") and "/moving_" not in k and "/gamma" not in k and "/beta" not in k}
    h.close()
    return loss, grads


def _spread(w):
    for k in w:                                  # as tests/test_gpu_train.py: spread the K samples so that the ranking gradients do not vanish
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    return w


@pytest.mark.parametrize("kw", [dict(), dict(bf16=2), dict(bn_mode=1), dict(mno=16, H=64, K=2)], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()) or "fp32")
def test_training_step_on_compact_rows_matches_the_uncompacted_one(torch_cuda, kw):
    """Loss terms and every weight gradient of the compacted step against the uncompacted step: the same sums over the same present rows in a
    different order (absent rows contribute exact zeros to the uncompacted sums), so they agree to fp32 reduction noise."""
    d = small_dims(**{**dict(n_scenes=4, K=3, T_obs=6, T_pred=7, n_grids=1), **kw})
    w = _spread(init_weights(d, 41))
    past, fut, eps, grids, gos, keep = ragged_case(d, seed=42)
    la, ga = _train_step(torch_cuda, d, w, past, fut, eps, grids, gos)
    lb, gb = _train_step(torch_cuda, d.replace(flags=FLAG_COMPACT_ROWS), w, past, fut, eps, grids, gos)
    for key in ("recon", "kld", "ce", "reg", "loss"):
        assert abs(la[key] - lb[key]) <= 1e-6 * max(1.0, abs(la[key])), (key, la[key], lb[key])
    assert la["n_present"] == lb["n_present"] > 0
    worst = ("", 0.0)
    for k in ga:
        if k == "ioc/score/b":                       # softmax over K is shift invariant: the exact gradient is 0, both are rounding noise
            assert np.abs(gb[k]).max() < 1e-6
            continue
        ref = np.abs(ga[k]).max()
        err = float(np.abs(ga[k] - gb[k]).max() / (ref + 1e-12)) if ref > 1e-9 else float(np.abs(gb[k]).max())
        if err > worst[1]:
            worst = (k, err)
        assert np.isfinite(gb[k]).all()
    print("compact vs uncompacted training step: worst relative gradient difference %.2e (%s)" % (worst[1], worst[0]))
    assert worst[1] < (5e-5 if kw.get("bf16") else 2e-5), worst


def test_training_gradients_on_compact_rows_match_float64_autograd(torch_cuda):
    """The compacted training step against the float64 autograd oracle (oracle/desire_torch.py), the bar of tests/test_gpu_train.py."""
    from oracle import desire_torch as OT
    from tests.helpers import to_oracle_layout
    d = small_dims(n_scenes=2, mno=32, K=3, T_obs=6, T_pred=7, n_grids=1)
    w = _spread(init_weights(d, 41))
    past, fut, eps, grids, gos, keep = ragged_case(d, seed=43, keep=0.4)
    vals, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    loss, g = _train_step(torch_cuda, d.replace(flags=FLAG_COMPACT_ROWS), w, past, fut, eps, grids, gos)
    assert abs(loss["loss"] - float(vals["loss"])) < 1e-4 * max(1.0, abs(float(vals["loss"])))
    for k in g:
        if k not in ref or k == "ioc/score/b":
            continue
        rel = float(np.abs(g[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-12))
        assert rel < 2e-4, (k, rel)


# ---- IOC slot classes (DESIRE_FLAG_COMPACT_IOC) --------------------------------------------------------------------------------------------
def run_opts(torch, d, w, past, fut, eps, grids, gos, min_rows=None, Y_in=None):
    from desire_amd import _lib
    h = _lib.Handle(d)
    h.set_weights(w)
    if min_rows is not None:
        h.set_option("compact_min_rows", min_rows)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.full((d.R, d.T_pred, 2), 7.0, device=dev)
    score = torch.full((d.R,), 3.0, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    if Y_in is None:
        h.forward(past_t.data_ptr(), fut_t.data_ptr() if d.posterior else 0, eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), s)
    else:
        h.encode(past_t.data_ptr(), fut_t.data_ptr() if d.posterior else 0, s)
        Y.copy_(t(Y_in))
        h.ioc_refine(Y.data_ptr(), score.data_ptr(), s)
    torch.cuda.synchronize()
    prof_names = None
    h.close()
    return Y.cpu().numpy(), score.cpu().numpy()


def ragged_counts(d, seed, counts):
    """make_case windows where window i keeps its first... no: a random subset of counts[i] slots (so every slot class gets windows)."""
    past, fut, eps, grids, gos = make_case(d, seed=seed, n_absent=0)
    rng = np.random.default_rng(seed + 7)
    keep = np.zeros((d.n_scenes, d.mno), bool)
    for i in range(d.n_scenes):
        keep[i, rng.permutation(d.mno)[: counts[i % len(counts)]]] = True
    past[~keep[:, None, :].repeat(d.T_obs, 1)] = 0
    fut[~keep[:, None, :].repeat(d.T_pred, 1)] = 0
    return past, fut, eps, grids, gos, keep


@pytest.mark.parametrize("kw", [
    dict(),
    dict(bf16=2),
    dict(bf16=3),
    dict(bf16=1),
    dict(mno=64, n_scenes=6, K=2),                   # top class = the cluster / 64-row forms, lower classes the 32-row tile
    dict(mno=128, n_scenes=8, K=2, counts=[3, 40, 70, 100, 128, 0, 33, 64]),      # classes 32 / 64 / 96 / 128 (deathCircle-sized scenes)
    dict(mno=16, H=64, K=3),
    dict(bin_mode=1, grid_size=4, nb_h=0.02, nb_w=0.3),
    dict(iters=2),
], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items() if kv[0] != "counts") or "fp32")
@pytest.mark.parametrize("rows_too", [False, True], ids=["ioc_only", "rows+ioc"])
def test_slot_classes_reproduce_the_scene_shaped_ioc(torch_cuda, kw, rows_too):
    """Every window re-seated in its slot class (fold threshold 0: every class that has windows runs on its own) against the scene-shaped
    pass on the SAME decoded positions: identical neighbour sets and cells, sums regrouped -> 1e-5 on trajectories (normalised units)."""
    from desire_amd.spec import FLAG_COMPACT_IOC
    kw = dict(kw)
    counts = kw.pop("counts", None)
    d = small_dims(**{**dict(n_scenes=8, K=4, T_pred=12), **kw})
    w = init_weights(d, 3)
    past, fut, eps, grids, gos, keep = ragged_counts(d, seed=21, counts=counts or [3, 9, 0, 14, 8, d.mno, 1, 20])
    Ya, sa = run_opts(torch_cuda, d, w, past, fut, eps, grids, gos)
    flags = FLAG_COMPACT_IOC | (FLAG_COMPACT_ROWS if rows_too else 0)
    Yb, sb = run_opts(torch_cuda, d.replace(flags=flags), w, past, fut, eps, grids, gos, min_rows=0)
    m = row_mask(d, keep)
    tolY, tolS = (2e-3, 2e-2) if kw.get("bf16") == 1 else (1e-5, 1e-4)
    assert np.isfinite(Yb).all() and np.isfinite(sb).all()
    assert np.abs(Yb - Ya)[m].max() < tolY, np.abs(Yb - Ya)[m].max()
    assert np.abs(sb - sa)[m].max() < tolS, np.abs(sb - sa)[m].max()
    assert not sb[~m].any()                                # rows that were not run: score 0
    # default fold threshold (everything here is far below it): one class, the handle's own -> the scene-shaped result up to row order
    Yc, sc_ = run_opts(torch_cuda, d.replace(flags=flags), w, past, fut, eps, grids, gos)
    assert np.abs(Yc - Ya)[m].max() < tolY and np.abs(sc_ - sa)[m].max() < tolS


@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_slot_classes_on_the_sdd_goldens(torch_cuda, tag):
    """IOC on the ORACLE's decoder output (bins identical by construction), windows seated in their slot classes: every present row within 1e-3."""
    from desire_amd.spec import FLAG_COMPACT_IOC
    d, g, eps, grids, gos, w = load_case(tag)
    present = g["past"][:, -1, :, 0] != 0
    m = row_mask(d, present)
    Y, score = run_opts(torch_cuda, d.replace(flags=FLAG_COMPACT_IOC | FLAG_COMPACT_ROWS), w, g["past"], g["fut"], eps, grids, gos, min_rows=0, Y_in=g["Y0"])
    assert np.abs(Y - g["Y"])[m].max() < 1e-3
    assert np.abs(score - g["score"])[m].max() < 5e-3


@pytest.mark.parametrize("kw", [dict(), dict(bf16=2), dict(iters=2), dict(mno=64, n_scenes=6, K=2, grid_size=4)], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()) or "fp32")
def test_training_step_on_slot_classes_matches_the_uncompacted_one(torch_cuda, kw):
    """DESIRE_FLAG_COMPACT_ROWS | DESIRE_FLAG_COMPACT_IOC in training: the IOC forward with saves, its BPTT and its weight-gradient reductions run
    per slot class (fold threshold 0), everything else as in the compact-rows step; loss terms and gradients against the uncompacted step."""
    from desire_amd.spec import FLAG_COMPACT_IOC
    d = small_dims(**{**dict(n_scenes=8, K=3, T_obs=6, T_pred=7, n_grids=1), **kw})
    w = _spread(init_weights(d, 41))
    past, fut, eps, grids, gos, keep = ragged_counts(d, seed=44, counts=[3, 9, 0, 14, 8, d.mno, 1, 20])
    la, ga = _train_step(torch_cuda, d, w, past, fut, eps, grids, gos)
    lb, gb = _train_step(torch_cuda, d.replace(flags=FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC), w, past, fut, eps, grids, gos, min_rows=0)
    for key in ("recon", "kld", "ce", "reg", "loss"):
        assert abs(la[key] - lb[key]) <= 2e-6 * max(1.0, abs(la[key])), (key, la[key], lb[key])
    worst = ("", 0.0)
    for k in ga:
        if k == "ioc/score/b":                       # softmax over K is shift invariant: the exact gradient is 0, both are rounding noise
            assert np.abs(gb[k]).max() < 1e-6
            continue
        ref = np.abs(ga[k]).max()
        err = float(np.abs(ga[k] - gb[k]).max() / (ref + 1e-12)) if ref > 1e-9 else float(np.abs(gb[k]).max())
        if err > worst[1]:
            worst = (k, err)
        assert np.isfinite(gb[k]).all()
    print("slot classes vs uncompacted training step: worst relative gradient difference %.2e (%s)" % (worst[1], worst[0]))
    assert worst[1] < (1e-4 if kw.get("bf16") else 3e-5), worst


def test_model_api_with_skip_padding(torch_cuda):
    """The Python surface: DESIREModel(args with dims_flags = COMPACT_ROWS | COMPACT_IOC).forward on loader-shaped windows equals the padded model on
    present agents, and a compacted train_step moves the same loss."""
    import argparse
    from desire_amd.model import DESIREModel
    from desire_amd.spec import FLAG_COMPACT_IOC
    from desire_amd.train import build_parser
    def mk(flags):
        a = build_parser().parse_args(["--d_dim", "64", "--seq_length", "8", "--pred_length", "12", "--max_num_obj", "32", "--batch_size", "4"])
        a.dims_flags = flags
        a.num_samples = 4
        return a
    d = small_dims(n_scenes=4, K=4, T_obs=8, T_pred=12, H=64)
    past, fut, eps, grids, gos, keep = ragged_counts(d, seed=5, counts=[9, 3, 0, 14])
    x = [p.astype(np.float64) for p in past]; y = [f.astype(np.float64) for f in fut]
    m0 = DESIREModel(mk(0), seed=3)
    m1 = DESIREModel(mk(FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC), seed=3)
    Y0, s0 = m0.forward(x, y, seed=1)
    Y1, s1 = m1.forward(x, y, seed=1)
    torch_cuda.cuda.synchronize()
    k = torch_cuda.as_tensor(keep, device=Y0.device)[:, None, :].expand(-1, Y0.shape[1], -1)
    assert float((Y0 - Y1).abs()[k].max()) < 1e-5 and float((s0 - s1).abs()[k].max()) < 1e-4
    assert abs(float(m0.cost) - float(m1.cost)) < 1e-6 * max(1.0, abs(float(m0.cost)))
    l0 = m0.train_step(x, y, seed=2); l1 = m1.train_step(x, y, seed=2)
    assert abs(float(l0["loss"]) - float(l1["loss"])) < 1e-5 * max(1.0, abs(float(l0["loss"])))


def test_training_with_the_gaussian_head_loss_keeps_every_observed_object(torch_cuda):
    """The Gaussian-head term counts (object, observed frame) pairs of objects that may have left by the last observed frame -- they are outside the
    present-agent map -- so with desire_set_head_loss on the encoder stack stays on all agents (the per-row stages still compact): loss terms and
    gradients, gauss_head included, equal the padded step's."""
    from desire_amd import _lib
    from desire_amd.spec import FLAG_COMPACT_IOC
    torch = torch_cuda
    d = small_dims(n_scenes=4, K=3, T_obs=6, T_pred=7, n_grids=1)
    w = _spread(init_weights(d, 41))
    past, fut, eps, grids, gos, keep = ragged_case(d, seed=46)
    res = []
    for flags in (0, FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC):
        h = _lib.Handle(d.replace(flags=flags)); h.set_weights(w); h.set_option("compact_min_rows", 0); h.set_training(True); h.set_head_loss(0.5)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
        p_t, f_t, e_t, g_t = t(past), t(fut), t(eps), t(grids)
        h.set_scene_grids(g_t.data_ptr(), gos)
        Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
        h.forward(p_t.data_ptr(), f_t.data_ptr(), e_t.data_ptr(), Y.data_ptr(), sc.data_ptr())
        h.backward(p_t.data_ptr(), f_t.data_ptr(), e_t.data_ptr())
        torch.cuda.synchronize()
        res.append({k: h.get_grad(k, w[k].shape) for k in ("gauss_head/w", "gauss_head/b", "enc_x/gates/kernel", "enc_y/candidate/kernel", "fc_c/w", "vae_enc/conv2/w")})
        h.close()
    for k in res[0]:
        ref = np.abs(res[0][k]).max()
        assert ref > 0 and np.abs(res[0][k] - res[1][k]).max() / ref < 3e-5, k


# ---- round 6: device-side counts (kernels.h: DynCount) -- inference sizes every compacted launch for the worst case and reads P / the class counts on the device
@pytest.mark.parametrize("kw", [dict(), dict(bf16=2), dict(bf16=3), dict(bf16=1), dict(H=64, K=3, mno=16), dict(posterior=0), dict(mno=64, n_scenes=5, K=2, n_grids=1),
                                dict(mno=96, n_scenes=3, K=2, n_grids=1), dict(mno=128, n_scenes=3, K=2, n_grids=1, bf16=1),
                                dict(mno=192, n_scenes=2, K=2, T_pred=8, n_grids=1),                       # step-wise IOC: the slot classes are ignored, the rows compact
                                dict(mno=64, H=256, n_scenes=3, K=2, n_grids=1, bf16=2),                   # H = 256 with split operands: step-wise as well
                                dict(ioc_form=8), dict(ioc_form=2)],                                       # row-compacted pooling / 64-row tiles as the class kernels
                         ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()) or "fp32")
def test_device_side_counts_equal_the_read_back_counts(torch_cuda, kw):
    """The default inference path (no host wait) against desire_set_option("compact_host_counts", 1) (the round-5 path: counts read back, launches sized
    exactly): per-row stages bit-identical; the IOC equal up to which tiling the launcher picks for the worst-case row count (fp32 summation order)."""
    torch = torch_cuda
    from desire_amd import _lib
    d = small_dims(**{**dict(n_scenes=8, K=3, T_obs=6, T_pred=7, n_grids=1), **kw}).replace(flags=FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC)
    w = init_weights(d, 41)
    counts = [3, 9, 0, 14, 8, d.mno, 1, 20][: d.n_scenes]
    past, fut, eps, grids, gos, keep = ragged_counts(d, seed=44, counts=[min(c, d.mno) for c in counts])
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    p_t, f_t, e_t, g_t = t(past), t(fut), t(eps), t(grids)
    res = []
    for host_counts in (0, 1):
        h = _lib.Handle(d); h.set_weights(w); h.set_option("compact_min_rows", 0); h.set_option("compact_host_counts", host_counts)
        h.set_scene_grids(g_t.data_ptr(), gos)
        Y = torch.full((d.R, d.T_pred, 2), 7.0, device="cuda"); sc = torch.full((d.R,), 3.0, device="cuda")
        h.forward(p_t.data_ptr(), f_t.data_ptr() if d.posterior else 0, e_t.data_ptr(), Y.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        res.append((h.read_buffer("Y0", (d.R, d.T_pred, 2)), Y.cpu().numpy(), sc.cpu().numpy()))
        h.close()
    np.testing.assert_array_equal(res[0][0], res[1][0])                                  # sample generation: the same kernels on the same rows
    tol = 2e-2 if d.bf16 == 1 else 2e-6
    assert np.abs(res[0][1] - res[1][1]).max() <= tol and np.abs(res[0][2] - res[1][2]).max() <= 50 * tol
    m = row_mask(d, keep)
    assert np.abs(res[0][0][m]).max() > 0 and not np.abs(res[0][0][~m]).any()


@pytest.mark.parametrize("which", ["none", "all", "one"])
def test_device_side_counts_edge_cases_and_changing_batches(torch_cuda, which):
    """Nothing / everything / one agent present, then a DIFFERENT presence pattern through the same handle (the counts are per call, nothing is cached on
    the host): equal to a fresh uncompacted handle on present rows."""
    torch = torch_cuda
    from desire_amd import _lib
    d = small_dims(n_scenes=4, K=3, T_obs=6, T_pred=7, n_grids=1)
    w = init_weights(d, 9)
    cnts = {"none": [0, 0, 0, 0], "all": [d.mno] * 4, "one": [0, 1, 0, 0]}[which]
    cases = [ragged_counts(d, seed=5, counts=cnts), ragged_counts(d, seed=6, counts=[5, 0, d.mno, 11])]
    hc = _lib.Handle(d.replace(flags=FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC)); hc.set_weights(w); hc.set_option("compact_min_rows", 0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    for past, fut, eps, grids, gos, keep in cases:
        _, Yr, sr = run(torch, d, w, past, fut, eps, grids, gos)
        p_t, f_t, e_t, g_t = t(past), t(fut), t(eps), t(grids)
        hc.set_scene_grids(g_t.data_ptr(), gos)
        Y = torch.full((d.R, d.T_pred, 2), 7.0, device="cuda"); sc = torch.full((d.R,), 3.0, device="cuda")
        hc.forward(p_t.data_ptr(), f_t.data_ptr(), e_t.data_ptr(), Y.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        m = row_mask(d, keep)
        Yc, scc = Y.cpu().numpy(), sc.cpu().numpy()
        assert np.isfinite(Yc).all() and np.isfinite(scc).all()
        if m.any():
            assert np.abs(Yc[m] - Yr[m]).max() < 2e-6 and np.abs(scc[m] - sr[m]).max() < 1e-4
        assert not np.abs(Yc[~m]).any() and not np.abs(scc[~m]).any()
    hc.close()


def test_a_compacted_forward_replays_from_a_hipgraph(torch_cuda):
    """no hipEventSynchronize in the compacted inference call any more, so it can be captured -- and the replayed graph follows the
    DATA: the same graph run on windows with another presence pattern gives that batch's results (the counts are read on the device at replay time)."""
    torch = torch_cuda
    from desire_amd import _lib
    d = small_dims(n_scenes=6, K=4, T_obs=8, T_pred=12, n_grids=1).replace(flags=FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC)
    w = init_weights(d, 5)
    a = ragged_counts(d, seed=6, counts=[9, 3, 0, 14, d.mno, 1])
    b = ragged_counts(d, seed=7, counts=[0, 20, 7, 2, 0, 11])
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    p, f, e, g = t(a[0]), t(a[1]), t(a[2]), t(a[3])
    h = _lib.Handle(d); h.set_weights(w); h.set_option("compact_min_rows", 0); h.set_scene_grids(g.data_ptr(), a[4])
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
    side = torch.cuda.Stream(); sp = side.cuda_stream
    torch.cuda.synchronize()
    ref = {}
    for tag, case in (("b", b), ("a", a)):                    # direct calls (the first also warms the lazy allocations up outside capture)
        p.copy_(t(case[0])); f.copy_(t(case[1])); e.copy_(t(case[2]))
        h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr(), sp)
        side.synchronize()
        ref[tag] = (Y.clone(), sc.clone())
    h.graph_begin(sp)
    h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr(), sp)
    gid = h.graph_end(sp)
    for rep in range(6):
        tag, case = ("a", a) if rep % 2 == 0 else ("b", b)
        p.copy_(t(case[0])); f.copy_(t(case[1])); e.copy_(t(case[2]))
        Y.zero_(); sc.zero_()
        torch.cuda.synchronize()
        h.graph_launch(gid, sp)
        side.synchronize()
        assert torch.equal(Y, ref[tag][0]) and torch.equal(sc, ref[tag][1]), (rep, tag)
    assert float(ref["a"][0].abs().max()) > 0 and not torch.equal(ref["a"][0], ref["b"][0])
    # the read-back path still refuses capture
    h.set_option("compact_host_counts", 1)
    h.graph_begin(sp)
    with pytest.raises(_lib.DesireError):
        h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr(), sp)
    try:
        h.graph_end(sp)
    except _lib.DesireError:
        pass
    h.close()
