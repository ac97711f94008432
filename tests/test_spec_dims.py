"""Host-side dimension contract (CPU): what spec.Dims accepts / refuses for the round-2 modes, and that the model's argument
mapping produces the reference graph's dims for `ref_compat`."""
import argparse

import pytest

from desire_amd.spec import Dims, flops_per_sample, init_weights, weight_shapes


def test_small_hidden_widths_are_accepted_with_logical_shapes():
    for H in (16, 32, 64, 128, 256):
        d = Dims(H=H)
        d.validate()
        s = weight_shapes(d)
        assert s["dec/gates/kernel"] == (2 * H, 2 * H) and s["ioc/social_fc/w"] == (16 * H, H) and s["gauss_head/w"] == (H, 5)
    with pytest.raises(ValueError):
        Dims(H=48).validate()
    assert flops_per_sample(Dims(H=16)) < flops_per_sample(Dims(H=128))


def test_new_weights_are_appended_so_committed_goldens_keep_their_values():
    names = list(weight_shapes(Dims()))
    assert names[-2:] == ["gauss_head/w", "gauss_head/b"]          # init_weights draws in this order (tests/golden/e2e_*.npz)
    w = init_weights(Dims(), 0)
    assert w["gauss_head/w"].shape == (128, 5)


def test_ref_compat_constraints():
    ok = Dims(K=1, T_obs=8, T_pred=8, H=16, bn_mode=1, ref_compat=1, n_dec=7, sx=1.0, sy=1.0)
    ok.validate()
    for bad in (dict(K=2), dict(H=32), dict(T_pred=12), dict(bn_mode=0), dict(n_dec=0), dict(bf16=1), dict(posterior=0)):
        with pytest.raises(ValueError):
            ok.replace(**bad).validate()
    with pytest.raises(ValueError):
        Dims(n_dec=7).validate()                                     # n_dec belongs to ref_compat


def test_dims_from_args_reference_graph():
    from desire_amd.model import dims_from_args
    from desire_amd.train import build_parser
    args = build_parser().parse_args([])
    d = dims_from_args(args, 3, True, ref_compat=True)
    assert (d.H, d.T_obs, d.T_pred, d.K, d.n_dec, d.mno, d.sx, d.sy, d.bn_mode, d.ref_compat) == (16, 8, 8, 1, 7, 64, 1.0, 1.0, 1, 1)
    d.validate()
    d2 = dims_from_args(args, 3, True)
    assert (d2.H, d2.T_pred, d2.K, d2.bn_mode) == (16, 8, 20, 0)
    args.batch_norm = "batch"
    assert dims_from_args(args, 1).bn_mode == 2
