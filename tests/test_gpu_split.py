"""Split-bf16 operands (dims.bf16 = 2, kernels_x3.hip): the IOC kernel's fp32 products as three bf16 MFMAs (hi.hi + lo.hi +
hi.lo, fp32 accumulate) against the PLAIN fp32 oracle -- no rounding oracle is needed, the form claims fp32-equivalence.

The IOC pass starts from the oracle's own Y0 (positions decide scene cells and social bins; see bench.py's accuracy gate),
so what is compared is the arithmetic of the contractions.  Tolerance: 1e-4 on trajectories in normalised frame
coordinates (measured 2e-6 .. 6e-6), a tenth of north_star's 1e-3 gate and two orders below what plain bf16 operands give."""
import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims
from tests.test_golden_e2e import load_case
from tests.test_gpu_parity import oracle_forward, run_gpu, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu

TOL_Y = 1e-4


@pytest.mark.parametrize("kw", [
    dict(),                                              # 32-agent groups, H=128
    dict(mno=16, n_scenes=3, K=5),                       # two groups per tile, ragged last tile
    dict(H=64, T_pred=7, K=3),
    dict(H=32, T_pred=9, K=2, mno=8),                    # logical width 32 zero-padded to the 64-wide tile
    dict(H=16, T_pred=8, T_obs=8, K=1, mno=4, n_scenes=3),
    dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2),          # ~8 neighbours per bin
    dict(mno=8, n_scenes=5, K=3),
    dict(mno=1, n_scenes=3, K=2, n_absent=0),            # nobody to pool from
    dict(T_pred=40, K=2),                                # the headline's horizon
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2),          # 36 bins (more masks than two workgroups per CU leave room for)
    dict(nb_w=0.04, nb_h=0.04, K=2),                     # sparse windows: empty bins skipped per tile
    dict(bin_mode=1, grid_size=4, nb_w=0.45, nb_h=0.04, K=2),       # log-polar bins
    dict(posterior=0, K=3),
    dict(mno=64, n_scenes=1, K=2, n_grids=1),            # one 64-agent group per 64-row tile (kernels_x6r2.hip with two pieces)
    dict(mno=64, n_scenes=2, K=3, n_grids=1, H=64),
])
def test_ioc_split_operands_match_fp32_oracle(torch_cuda, kw):
    kw = dict(kw)
    n_absent = kw.pop("n_absent", 3)
    d = small_dims(**kw)
    w = init_weights(d, 3)
    past, fut, eps, grids, gos = make_case(d, seed=4, n_absent=min(n_absent, d.mno - 1))
    tab = None
    if d.bin_mode == 1:
        from desire_amd import _lib
        hb = _lib.Handle(d); hb.set_weights(w); tab = hb.bin_table(); hb.close()
    ref = oracle_forward(d, w, past, fut, eps, grids, gos, bin_tab=tab)
    _, Y, score = run_gpu(torch_cuda, d.replace(bf16=2), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    _, Yf, scoref = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    err, errf = np.abs(Y - ref["Y"]).max(), np.abs(Yf - ref["Y"]).max()
    print("split operands vs oracle %.2e (fp32 kernel: %.2e)" % (err, errf))
    assert err < TOL_Y, (err, errf)
    assert np.abs(score - ref["score"]).max() < 1e-4 * max(1.0, np.abs(ref["score"]).max())


def test_split_operands_two_refinement_passes(torch_cuda):
    """Pass 2 re-bins from pass 1's output, which differs from the oracle's by ~1e-6: a neighbour within that distance of a bin
    edge may change bins, so the bulk is checked tightly and the whole in the mean."""
    d = small_dims(iters=2, K=3)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6, n_absent=2)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    _, Y, _ = run_gpu(torch_cuda, d.replace(bf16=2), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    e = np.abs(Y - ref["Y"]).reshape(d.R, -1).max(1)
    assert np.median(e) < 2e-5 and e.mean() < 1e-3, (np.median(e), e.mean(), e.max())
    assert (e < TOL_Y).mean() > 0.97


@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_split_operands_reproduce_goldens(tag):
    """The committed real-SDD goldens (BASELINE configs[0] and [1]) through the split form: IOC on the golden decoder output."""
    import torch
    from desire_amd import _lib
    d, g, eps, grids, gos, w = load_case(tag)
    h = _lib.Handle(d.replace(bf16=2))
    h.set_weights(w)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past, fut, grids_t = t(g["past"]), t(g["fut"]), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    h.encode(past.data_ptr(), fut.data_ptr())
    Y = t(g["Y0"]).clone(); score = torch.zeros((d.R,), device="cuda")
    h.ioc_refine(Y.data_ptr(), score.data_ptr())
    torch.cuda.synchronize()
    assert np.abs(Y.cpu().numpy() - g["Y"]).max() < TOL_Y
    assert np.abs(score.cpu().numpy() - g["score"]).max() < 1e-3
    h.close()


def test_shapes_without_a_split_kernel_run_the_fp32_kernels(torch_cuda):
    """dims.bf16 = 2 promises AT LEAST split accuracy: groups of 96 agents (the cluster form has no split IOC kernel) run the fp32 IOC
    kernel, bit-identically to dims.bf16 = 0 from the same decoder output; sample generation runs the six-product kernels (fp32 class:
    Y0 within 1e-6)."""
    d = small_dims(mno=96, n_scenes=1, K=2, n_grids=1)
    w = init_weights(d, 7)
    past, fut, eps, grids, gos = make_case(d, seed=8, n_absent=5)
    ha, _, _ = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    hb, _, _ = run_gpu(torch_cuda, d.replace(bf16=2), w, past, fut, eps, grids, gos)
    Y0a, Y0b = ha.read_buffer("Y0", (d.R, d.T_pred, 2)), hb.read_buffer("Y0", (d.R, d.T_pred, 2))
    assert np.abs(Y0a - Y0b).max() < 1e-6
    _, Ya, sa = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=Y0a)
    _, Yb, sb = run_gpu(torch_cuda, d.replace(bf16=2), w, past, fut, eps, grids, gos, Y_in=Y0a)
    assert np.array_equal(Ya, Yb) and np.array_equal(sa, sb)


@pytest.mark.parametrize("kw", [
    dict(n_scenes=2, mno=32, K=3, T_obs=6, T_pred=7, n_grids=1),                       # the fixture of tests/test_gpu_train.py
    dict(n_scenes=3, mno=16, K=5, T_obs=6, T_pred=7, n_grids=1),                       # two groups per tile, ragged last tile
    dict(n_scenes=2, mno=32, K=2, T_obs=6, T_pred=6, n_grids=1, H=64, case_seed=44),   # the 64-wide instantiations
    dict(n_scenes=2, mno=32, K=2, T_obs=6, T_pred=6, n_grids=1, grid_size=6, nb_w=0.5, nb_h=0.5),    # 36 bins
    dict(n_scenes=2, mno=32, K=2, T_obs=6, T_pred=6, n_grids=1, iters=2),              # two refinement passes accumulate
])
def test_every_weight_gradient_under_split_operands(torch_cuda, kw):
    _weight_gradients_under_split_operands(torch_cuda, kw, flags=0, tol=2e-4)


@pytest.mark.parametrize("kw", [
    dict(n_scenes=2, mno=32, K=3, T_obs=6, T_pred=7, n_grids=1),
    dict(n_scenes=3, mno=16, K=5, T_obs=6, T_pred=7, n_grids=1),
])
def test_weight_gradients_with_the_two_piece_training_forward(torch_cuda, kw):
    """dims.flags = DESIRE_FLAG_TRAIN_FWD_3P: the forward pass's sample generation with two-piece operands (three products) -- the faster
    training step the header documents, with the accuracy it documents: every weight gradient within 1e-3 of float64 autograd (observed
    <= 5e-4; the default six-product forward keeps the 2e-4 of the test above).  (Cases away from the ReLU kinks the docstring below
    describes: a forward that moves Y0 by 1e-5 takes the other branch of an e_r pre-activation that float64 has within that distance of
    zero, and the gradient then differs by a whole term -- the 64-wide case of the list above is such a case under this flag.)"""
    from desire_amd.spec import FLAG_TRAIN_FWD_3P
    _weight_gradients_under_split_operands(torch_cuda, kw, flags=FLAG_TRAIN_FWD_3P, tol=1e-3)


def _weight_gradients_under_split_operands(torch_cuda, kw, flags, tol):
    """the training step under dims.bf16 = 2 -- IOC forward, IOC BPTT (k_ioc_bwd_x3), every large weight-gradient
    reduction (k_gemm_tn2_xp) and the two large data-gradient convolutions (k_conv_gather_x3) with split-bf16 operands -- against
    float64 autograd of the oracle, every one of the 46 weight gradients inside the fp32 training tests' own 2e-4.
    (The loss is only piecewise smooth: an e_r pre-activation within the split forward's ~3e-6 of the ReLU kink takes the other branch
    than float64 does and changes dpre_r by a whole term -- seed 42 of the 64-wide case has one such element at (row 17, t 3, column 42),
    with the backward pass entirely fp32 as well; its case uses another seed.  tests/fuzz_train.py moves such points off the kink.)"""
    kw = dict(kw)
    case_seed = kw.pop("case_seed", 42)
    from desire_amd import _lib
    from oracle import desire_torch as OT
    from tests.helpers import to_oracle_layout
    from tests.test_gpu_train import DONE, rel_err
    torch = torch_cuda
    d = small_dims(**kw)
    w = init_weights(d, 41)
    for k in w:
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    past, fut, eps, grids, gos = make_case(d, seed=case_seed, n_absent=min(4, d.mno - 1))
    _, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    h = _lib.Handle(d.replace(bf16=2, flags=flags)); h.set_weights(w)
    h.set_scene_grids(g.data_ptr(), gos)
    h.set_training(True)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
    h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
    h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
    torch.cuda.synchronize()
    bad = {}
    for name in DONE:
        got = h.get_grad(name, w[name].shape)
        assert np.isfinite(got).all(), name
        if name == "ioc/score/b":
            assert np.abs(got).max() < 1e-6
            continue
        r = rel_err(got, ref[name])
        if not r < tol:
            bad[name] = r
    h.close()
    assert not bad, bad


def test_split_mode_training(torch_cuda):
    """dims.bf16 = 2 while training: the IOC forward, the IOC BPTT, the weight-gradient reductions and the large data-gradient
    convolutions run with split operands; activations and gradients stay fp32 in HBM.  Gradients: against float64 autograd inside the training tests' own 2e-4, and within 1e-4 (relative, whole
    gradient vector) of the all-fp32 step; an optimiser step refreshes the split packs on the device."""
    from desire_amd import _lib
    from desire_amd.spec import weight_shapes
    from oracle import desire_torch as OT
    from tests.helpers import to_oracle_layout
    torch = torch_cuda
    d = small_dims(n_scenes=2, mno=32, K=3, T_obs=6, T_pred=7, n_grids=1)
    w = init_weights(d, 41)
    for k in w:                                   # (as tests/test_gpu_train.py: spread the K samples so ranking gradients exist)
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    past, fut, eps, grids, gos = make_case(d, seed=42, n_absent=4)
    _, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    grads = []
    for mode in (0, 2):
        h = _lib.Handle(d.replace(bf16=mode)); h.set_weights(w)
        h.set_scene_grids(g.data_ptr(), gos)
        h.set_training(True)
        Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
        h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
        h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
        torch.cuda.synchronize()
        grads.append(h.grad_tensor().clone())
        if mode == 2:
            worst = 0.0
            for name in ("ioc/gates/kernel", "ioc/candidate/kernel", "ioc/social_fc/w", "ioc/reg/w", "ioc/vel_fc/w", "ioc/score/w",
                         "dec/gates/kernel", "head/w", "vae_dec/deconv2/w", "enc_x/gates/kernel"):
                got = h.get_grad(name, w[name].shape)
                worst = max(worst, float(np.abs(got - ref[name]).max() / (np.abs(ref[name]).max() + 1e-12)))
            assert worst < 2e-4, worst
            # an optimiser step refreshes the split packs on the device (train.hip: k_repack_split): inference after training
            # must equal a fresh handle built from the trained weights
            h.adam_step(lr=1e-3)
            h.set_training(False)                     # (also hands the trained weights back to get_weight)
            Y0 = torch.zeros_like(Y)
            h.encode(p.data_ptr(), f.data_ptr())
            h.sample(e.data_ptr(), Y0.data_ptr())
            Ya = Y0.clone(); h.ioc_refine(Ya.data_ptr(), sc.data_ptr())
            torch.cuda.synchronize()
            w2 = {k: h.get_weight(k, tuple(shp)) for k, shp in weight_shapes(d).items()}
            h2 = _lib.Handle(d.replace(bf16=2)); h2.set_weights(w2)
            h2.set_scene_grids(g.data_ptr(), gos)
            h2.encode(p.data_ptr(), f.data_ptr())
            Yb = Y0.clone(); h2.ioc_refine(Yb.data_ptr(), sc.data_ptr())
            torch.cuda.synchronize()
            assert not np.allclose(w2["ioc/gates/kernel"], w["ioc/gates/kernel"])
            assert torch.equal(Ya, Yb)
            h2.close()
        h.close()
    rel = float((grads[0] - grads[1]).double().norm() / grads[0].double().norm())
    assert 0 < rel < 1e-4, rel


def test_model_surface_selects_split_operands(torch_cuda):
    """DESIREModel(args.bf16 = "x3") -> dims.bf16 = 2; forward (six-product sample generation, split IOC) agrees with the fp32 model
    far inside the 1e-3 gate."""
    import argparse
    from desire_amd.model import DESIREModel
    base = dict(rnn_size=512, num_layers=1, seq_length=8, pred_length=12, d_dim=128, e_dim=256, latent_size=128, max_num_obj=32,
                num_samples=3, batch_size=1, stride=1, grid_size=4, neighborhood_size=300, img_width=1400.0, img_height=1100.0,
                learning_rate=0.001, grad_clip=10.0)
    rng = np.random.default_rng(0)
    n = 12
    win = np.zeros((20, 32, 3), np.float32)
    win[:, :n, 0] = np.arange(1, n + 1)
    start = rng.uniform(300, 900, size=(1, n, 2)); vel = rng.uniform(-6, 6, size=(1, n, 2))
    win[:, :n, 1:] = start + vel * np.arange(20)[:, None, None]
    outs = []
    for mode in (False, "x3"):
        m = DESIREModel(argparse.Namespace(bf16=mode, **base), seed=1)
        Y, score = m.forward([win[:8]], [win[8:]], seed=3)
        assert m._handle(1, True).dims.bf16 == (2 if mode else 0)
        outs.append(Y.cpu().numpy())
    err = np.abs(outs[0] - outs[1]).max()
    assert 0 < err < TOL_Y, err


def test_split_path_is_window_independent_and_deterministic(torch_cuda):
    """BASELINE configs[1] shapes, 16 windows through the split-operand IOC kernel: together or in two halves, and run twice, the
    refined trajectories and scores are bit-identical (what scene-sharding over GPUs relies on).  Against the fp32 kernels' END-TO-END
    output: sample generation runs the six-product kernels here (Y0 within an ulp or two of the fp32 kernels'), so all but a fraction
    of a percent of the rows keep their cells and bins and agree to 1e-4; a row whose neighbour sat within 1e-7 of a bin edge may not."""
    from desire_amd.spec import Dims
    from desire_amd.synth import make_case as mk
    n = 16
    d = Dims(n_scenes=n, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4, nb_w=0.15, nb_h=0.15,
             sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1, bf16=2)
    w = init_weights(d, 0)
    past, fut, eps, grids, gos = mk(d, seed=1, n_absent=0)
    _, Y, s = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    _, Yr, sr = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    np.testing.assert_array_equal(Y, Yr)
    np.testing.assert_array_equal(s, sr)
    rows = d.K * d.mno
    for lo, hi in ((0, n // 2), (n // 2, n)):
        dd = d.replace(n_scenes=hi - lo)
        _, Yh, sh = run_gpu(torch_cuda, dd, w, past[lo:hi], fut[lo:hi], eps[lo * rows:hi * rows], grids, gos[lo:hi])
        np.testing.assert_array_equal(Yh, Y[lo * rows:hi * rows])
        np.testing.assert_array_equal(sh, s[lo * rows:hi * rows])
    _, Yf, sf = run_gpu(torch_cuda, d.replace(bf16=0), w, past, fut, eps, grids, gos)
    e = np.abs(Y - Yf).reshape(d.R, -1).max(1)
    assert 0 < np.median(e) < 2e-5 and (e > TOL_Y).mean() < 0.01, (np.median(e), (e > TOL_Y).mean(), e.max())
    assert np.median(np.abs(s - sf)) < 1e-4 * max(1.0, np.abs(sf).max())


# ---- three bf16 pieces per operand, six products per fp32 product (dims.bf16 = 3) ------------------------------
@pytest.mark.parametrize("kw", [
    dict(),
    dict(mno=16, n_scenes=3, K=5),
    dict(H=64, T_pred=7, K=3),
    dict(H=16, T_pred=8, T_obs=8, K=1, mno=4, n_scenes=3),
    dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2),
    dict(T_pred=40, K=2),
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2),
    dict(nb_w=0.04, nb_h=0.04, K=2),
    dict(bin_mode=1, grid_size=4, nb_w=0.45, nb_h=0.04, K=2),
    dict(iters=2, K=2),
    dict(mno=64, n_scenes=1, K=2, n_grids=1),            # one 64-agent group per 64-row tile
])
def test_six_product_form_is_as_close_to_the_oracle_as_the_fp32_kernel(torch_cuda, kw):
    """dims.bf16 = 3: x = hi + mid + lo EXACTLY (8 + 8 + 8 bits), products (hi,hi) (hi,mid) (mid,hi) (mid,mid) (hi,lo) (lo,hi): what is
    dropped is <= 2^-23 |a b| per product -- the class of the fp32 fmaf chain's own rounding.  Claim under test:
    its distance from the fp32 ORACLE is the fp32 kernel's own, not the ~1e-5 of the three-product form."""
    kw = dict(kw)
    d = small_dims(**kw)
    w = init_weights(d, 3)
    past, fut, eps, grids, gos = make_case(d, seed=4, n_absent=min(3, d.mno - 1))
    tab = None
    if d.bin_mode == 1:
        from desire_amd import _lib
        hb = _lib.Handle(d); hb.set_weights(w); tab = hb.bin_table(); hb.close()
    ref = oracle_forward(d, w, past, fut, eps, grids, gos, bin_tab=tab)
    if d.iters > 1:                       # pass 2 re-bins from pass 1's output: compare one pass at a time from the oracle's input
        d = d.replace(iters=1)
        ref = oracle_forward(d, w, past, fut, eps, grids, gos, bin_tab=tab)
    _, Y6, s6 = run_gpu(torch_cuda, d.replace(bf16=3), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    _, Y3, _ = run_gpu(torch_cuda, d.replace(bf16=2), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    _, Yf, sf = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    e6, e3, ef = np.abs(Y6 - ref["Y"]).max(), np.abs(Y3 - ref["Y"]).max(), np.abs(Yf - ref["Y"]).max()
    print("vs oracle: six products %.2e | three products %.2e | fp32 kernel %.2e;  six vs fp32 kernel %.2e" % (e6, e3, ef, np.abs(Y6 - Yf).max()))
    assert e6 < max(2.0 * ef, 1e-6), (e6, ef)              # the fp32 kernel's own class (both sit at a few 1e-7)
    assert np.abs(Y6 - Yf).max() < max(2e-6, 1.01 * (e6 + ef))      # (triangle inequality where a crowded bin makes both a few 1e-6)
    assert np.abs(s6 - ref["score"]).max() < max(2.0 * np.abs(sf - ref["score"]).max(), 2e-5 * max(1.0, np.abs(ref["score"]).max()))


@pytest.mark.parametrize("kw", [dict(T_pred=40, K=4, n_scenes=3), dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2), dict(H=64, T_pred=12, K=3)])
def test_six_product_form_against_float64(torch_cuda, kw):
    """Which fp32 implementation is closer to EXACT arithmetic?  The oracle evaluated in float64 on the same inputs (the IOC pass from
    the same fp32 Y0, so cells and bins are shared) is the yardstick; the fp32 numpy oracle, the fp32 MFMA kernel and the six-product
    kernel are three roundings of it.  Claim: the six-product form is not further from exact than the fp32 implementations are
    (what it drops per product, <= 2^-26 |a b| with round-to-nearest pieces, is below one fp32 accumulation rounding)."""
    d = small_dims(**kw)
    w = init_weights(d, 3)
    past, fut, eps, grids, gos = make_case(d, seed=4, n_absent=min(3, d.mno - 1))
    ref32 = oracle_forward(d, w, past, fut, eps, grids, gos)
    ref64 = oracle_forward(d, w, past, fut, eps, grids, gos, dt=np.float64, Y_override=ref32["Y0"])
    _, Y6, _ = run_gpu(torch_cuda, d.replace(bf16=3), w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    _, Yf, _ = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    err = lambda Y: np.abs(np.asarray(Y, np.float64) - ref64["Y"])
    e6, ef, eo = err(Y6), err(Yf), err(ref32["Y"])
    print("vs float64: six products max %.2e rms %.2e | fp32 kernel max %.2e rms %.2e | fp32 numpy oracle max %.2e rms %.2e"
          % (e6.max(), np.sqrt((e6 ** 2).mean()), ef.max(), np.sqrt((ef ** 2).mean()), eo.max(), np.sqrt((eo ** 2).mean())))
    worst32 = max(ef.max(), eo.max())
    assert e6.max() < max(1.5 * worst32, 5e-7), (e6.max(), ef.max(), eo.max())
    assert np.sqrt((e6 ** 2).mean()) < 1.5 * max(np.sqrt((ef ** 2).mean()), np.sqrt((eo ** 2).mean()))


@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_six_product_form_reproduces_goldens(tag):
    """Both real-SDD goldens through dims.bf16 = 3, IOC on the golden decoder output (so cells and bins are the golden's): the
    trajectories sit where the fp32 kernel's do."""
    import torch
    from desire_amd import _lib
    d, g, eps, grids, gos, w = load_case(tag)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past, fut, grids_t = t(g["past"]), t(g["fut"]), t(grids)
    out = {}
    for mode in (0, 3):
        h = _lib.Handle(d.replace(bf16=mode))
        h.set_weights(w)
        h.set_scene_grids(grids_t.data_ptr(), gos)
        h.encode(past.data_ptr(), fut.data_ptr())
        Y = t(g["Y0"]).clone(); score = torch.zeros((d.R,), device="cuda")
        h.ioc_refine(Y.data_ptr(), score.data_ptr())
        torch.cuda.synchronize()
        out[mode] = (Y.cpu().numpy(), score.cpu().numpy())
        h.close()
    e6, ef = np.abs(out[3][0] - g["Y"]).max(), np.abs(out[0][0] - g["Y"]).max()
    print("%s: six products vs golden %.2e, fp32 kernel vs golden %.2e" % (tag, e6, ef))
    assert e6 < max(2.0 * ef, 1e-6) and np.abs(out[3][0] - out[0][0]).max() < 2e-6
    assert np.abs(out[3][1] - g["score"]).max() < 1e-3


def test_six_product_form_refuses_training_and_falls_back_on_other_shapes(torch_cuda):
    from desire_amd import _lib
    d = small_dims(mno=96, n_scenes=1, K=2, n_grids=1)
    w = init_weights(d, 7)
    past, fut, eps, grids, gos = make_case(d, seed=8, n_absent=5)
    ha, Ya, sa = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    Y0 = ha.read_buffer("Y0", (d.R, d.T_pred, 2))
    _, Ya, sa = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=Y0)
    _, Yb, sb = run_gpu(torch_cuda, d.replace(bf16=3), w, past, fut, eps, grids, gos, Y_in=Y0)    # no six-product IOC form for 96-agent groups (cluster
    assert np.array_equal(Ya, Yb) and np.array_equal(sa, sb)                                      # form): the fp32 kernel runs, bit for bit
    h = _lib.Handle(small_dims().replace(bf16=3)); h.set_weights(init_weights(small_dims(), 1))
    with pytest.raises(_lib.DesireError):
        h.set_training(True)
    h.close()


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3, L=64), dict(T_pred=40, K=2), dict(posterior=0, K=3),
                                dict(H=16, T_pred=8, T_obs=8, K=2, mno=4, n_scenes=3), dict(H=256, K=2, T_pred=9), dict(H=256, mno=64, n_scenes=1, K=2, n_grids=1, T_pred=40)])
def test_six_product_sample_generation_stays_in_the_fp32_kernels_class(torch_cuda, kw):
    """dims.bf16 = 3 also runs the GRU decoder, deconv1-3 and the mask fc as six bf16 MFMAs per fp32 product (kernels_x6.hip).  Sample
    generation feeds a DISCONTINUOUS refinement (cells and bins are floors of the sampled positions), so the claim is strict: every stage
    sits where the fp32 kernels sit -- against the oracle no further than twice the fp32 kernel's own distance (or 1e-6), and within 2e-6
    of the fp32 kernels themselves.  (The fused deconv3 + deconv4 six-product kernel of round 3 was measured slower
    than the two kernels and is gone.)"""
    d = small_dims(**kw)
    w = init_weights(d, 9)
    past, fut, eps, grids, gos = make_case(d, seed=10, n_absent=min(3, d.mno - 1))
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    h6, _, _ = run_gpu(torch_cuda, d.replace(bf16=3), w, past, fut, eps, grids, gos)
    hf, _, _ = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    rep = {}
    for name, shp in (("d2", (d.R, 4096)), ("d3", (d.R, 8192)), ("xhat", (d.R, 1024)), ("xz", (d.R, d.H)), ("Y0", (d.R, d.T_pred, 2))):
        g6, gf, r = h6.read_buffer(name, shp), hf.read_buffer(name, shp), ref[name].reshape(shp)
        e6, ef, e6f = float(np.abs(g6 - r).max()), float(np.abs(gf - r).max()), float(np.abs(g6 - gf).max())
        rep[name] = (e6, ef, e6f)
        assert e6 < max(2.0 * ef, 1e-6), (name, e6, ef)
        assert e6f < 2e-6 * max(1.0, float(np.abs(r).max())), (name, e6f)
    print({k: tuple("%.1e" % x for x in v) for k, v in rep.items()})
    assert rep["Y0"][2] > 0                                  # (a different code path, not the fp32 kernels again)


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3), dict(mno=64, n_scenes=2, K=3, n_grids=1),
                                dict(mno=8, n_scenes=5, K=3), dict(nb_w=0.04, nb_h=0.04, K=2), dict(iters=2, K=2), dict(mno=1, n_scenes=3, K=2),
                                dict(bin_mode=1, grid_size=4, nb_w=0.45, nb_h=0.04, K=2), dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2)])
def test_six_product_ioc_on_64_row_tiles_matches_the_32_row_form(torch_cuda, kw):
    """kernels_x6r2.hip (two row blocks per wave, fp32 operand tiles split on the fly; the default six-product IOC kernel wherever a
    64-row tile fits) issues the same products in the same per-accumulator order as k_ioc_x3<NP = 3> (dims.ioc_form = DESIRE_IOC_X6_TILE32): refined
    trajectories and scores agree to an ulp or two -- ragged last tiles, 64-agent groups (which the 32-row form does not have: there
    the reference point is the fp32 kernel, 2e-6) and the 36-bin fallback included."""
    d = small_dims(bf16=3, **kw)
    w = init_weights(d, 33)
    past, fut, eps, grids, gos = make_case(d, seed=34, n_absent=min(2, d.mno - 1))
    ha, _, _ = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    Y0 = ha.read_buffer("Y0", (d.R, d.T_pred, 2))
    # DESIRE_IOC_X6_TILE64: 64-row tiles whatever the launch size (by default only launches of >= 256 such tiles)
    _, Ya, sa = run_gpu(torch_cuda, d.replace(ioc_form=14), w, past, fut, eps, grids, gos, Y_in=Y0)
    _, Yb, sb = run_gpu(torch_cuda, d.replace(bf16=3, ioc_form=13) if d.mno <= 32 else d.replace(bf16=0), w, past, fut, eps, grids, gos, Y_in=Y0)   # (64-agent groups: the fp32 kernel)
    assert np.isfinite(Ya).all() and np.abs(Ya - Y0).max() > 0
    # same products, same per-accumulator order -- but not the same bits: the 32-row form splits r*h (and friends) where it computes
    # them, and hipcc contracts the product into the split's subtraction (the pieces then carry the UNROUNDED product); a tile's
    # occupied-bin set (two row blocks vs one) also regroups the partial sums.  One or two ulp.  (At 64 agents per group the
    # comparison point is the fp32 kernel.)
    assert np.abs(Ya - Yb).max() < (5e-7 if d.mno <= 32 and d.iters == 1 else 2e-6)
    assert np.abs(sa - sb).max() < 2e-5 * max(1.0, np.abs(sb).max())


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("kw", [dict(mno=64, H=256, n_scenes=1, K=2, n_grids=1, T_pred=6), dict(mno=32, H=256, n_scenes=2, K=2, T_pred=7),
                                dict(mno=16, H=256, n_scenes=3, K=3, T_pred=5, nb_w=0.04, nb_h=0.04), dict(mno=128, H=256, n_scenes=1, K=1, n_grids=1, T_pred=5),
                                dict(mno=192, H=128, n_scenes=1, K=2, n_grids=1, T_pred=5)])
def test_hidden_256_and_large_scenes_have_a_split_ioc_form(torch_cuda, kw, mode):
    """VERDICT r03 Missing 5 / Next 4b: dims.bf16 = 2 / 3 at H = 256 (BASELINE configs[3]) used to fall back to the fp32 kernels.  The
    step-wise kernel (k_ioc_step<.., NP>: one launch per step, fp32 LDS tiles split on the fly, [hi | lo (| lo2)] weight packs in plain
    k order) gives those shapes -- and scenes of 160 .. 256 agents -- three / six bf16 MFMAs per product: against the oracle as close as
    the split kernels of the other shapes (1e-4 gate for two pieces; the fp32 kernels' own class for three)."""
    d = small_dims(**kw)
    w = init_weights(d, 27)
    past, fut, eps, grids, gos = make_case(d, seed=28, n_absent=min(5, d.mno - 2), spread=0.3)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    _, Yf, sf = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    _, Ys, ss = run_gpu(torch_cuda, d.replace(bf16=mode), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    ef, es = float(np.abs(Yf - ref["Y"]).max()), float(np.abs(Ys - ref["Y"]).max())
    assert (np.asarray(ref["Y"]) != np.asarray(ref["Y0"])).any() and not np.array_equal(Ys, Yf)          # (a different code path)
    if mode == 2:
        assert es < 1e-4, (es, ef)
    else:
        assert es < max(2.0 * ef, 2e-6), (es, ef)
    assert np.abs(ss - ref["score"]).max() < 5e-3
