"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/desire_hip.h declares, argument validation works without a GPU, and the product refuses
to compute without one (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from desire_amd import _lib
    return _lib.load()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "desire_hip.h")).read()
    declared = set(re.findall(r"\b(desire_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 16
    from desire_amd import _lib
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_dims_struct_matches_header():
    from desire_amd._lib import DesireDims
    hdr = open(os.path.join(ROOT, "include", "desire_hip.h")).read()
    body = hdr[hdr.index("typedef struct desire_dims {"):hdr.index("} desire_dims;")]
    names = [n.strip() for line in re.findall(r"(?:int32_t|float)\s+([^;]+);", body) for n in line.split(",")]
    assert names == [f[0] for f in DesireDims._fields_]
    assert ctypes.sizeof(DesireDims) == 4 * len(names)


def test_loaded_library_is_the_trees_code(lib):
    """built objects travel to the GPU box next to the sources (git-ignored, not gpurun-ignored); the build is content-hashed
    and the library carries the hash of the tree it was built from, so a stale .so cannot pass for the tree's code."""
    from desire_amd._build import source_hash
    assert lib.desire_build_hash().decode() == source_hash()
    assert lib.desire_dims_size() == ctypes.sizeof(__import__("desire_amd._lib", fromlist=["DesireDims"]).DesireDims)


def test_create_validates_dims_and_needs_a_gpu(lib):
    import torch
    from desire_amd import _lib
    from desire_amd.spec import Dims
    bad = _lib.DesireDims.from_dims(Dims(mno=32))
    bad.mno = 24
    h = ctypes.c_void_p()
    assert lib.desire_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert b"mno" in lib.desire_last_error()
    if not torch.cuda.is_available():
        ok = _lib.DesireDims.from_dims(Dims(mno=32))
        rc = lib.desire_create(ctypes.byref(ok), ctypes.byref(h))
        assert rc == -4 and b"no CPU path" in lib.desire_last_error() or rc == -3


def test_model_refuses_without_gpu():
    import argparse
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from desire_amd import _lib
    from desire_amd.model import DESIREModel
    args = argparse.Namespace(rnn_size=512, seq_length=8, d_dim=128, latent_size=128, max_num_obj=32,
                              learning_rate=0.005, grad_clip=10.0)
    with pytest.raises(_lib.DesireError):
        DESIREModel(args)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "desire_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace("(oracle/ is test-only)", ""), f


def test_dims_from_args_reference_defaults():
    """train.py:30-88 defaults -> dims (H = d_dim, S = int(sqrt(2*rnn_size)), mno padded 60 -> 64)."""
    import argparse
    from desire_amd.model import dims_from_args
    args = argparse.Namespace(rnn_size=512, num_layers=1, batch_size=10, seq_length=8, d_dim=128, e_dim=256,
                              latent_size=128, max_num_obj=60, learning_rate=0.005, grad_clip=10.0, stride=1,
                              neighborhood_size=32, grid_size=4)
    d = dims_from_args(args, 10)
    assert (d.S, d.V, d.mno, d.H, d.L, d.T_obs, d.T_pred, d.grid_size, d.B) == (32, 1024, 64, 128, 128, 8, 8, 4, 16)
    d.validate()
    # round 6: padding is skipped by default on the drop-in surface (max_num_obj 60 against ~8 objects per SDD frame, train.py:74); the opt-outs
    from desire_amd.spec import FLAG_COMPACT_IOC, FLAG_COMPACT_ROWS, FLAG_TRAIN_FWD_3P
    both = FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC
    assert d.flags == both
    assert dims_from_args(argparse.Namespace(**vars(args), keep_padding=True), 10).flags == 0
    assert dims_from_args(argparse.Namespace(**vars(args), dims_flags=0), 10).flags == 0                       # an explicit value is taken literally
    assert dims_from_args(argparse.Namespace(**vars(args), dims_flags=0, skip_padding=True), 10).flags == both
    assert dims_from_args(argparse.Namespace(**vars(args), batch_norm="batch"), 10).flags == 0                 # whole-batch statistics see the padding rows
    assert dims_from_args(argparse.Namespace(**vars(args), batch_norm="per_object", two_piece_forward=True), 10).flags == both | FLAG_TRAIN_FWD_3P
    assert dims_from_args(args, 10, ref_compat=True).flags == 0
    from desire_amd.train import build_parser
    assert dims_from_args(build_parser().parse_args([]), 10).flags == both
    assert dims_from_args(build_parser().parse_args(["--keep_padding"]), 10).flags == 0


def test_one_hip_runtime_in_the_process_whatever_the_import_order():
    """PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  If libdesire_hip.so is mapped BEFORE torch, the process ends up
    with two HIP runtimes and the second one to initialise sees no device (found on the GPU box: build() followed by smoke() in one
    process failed with DESIRE_ERR_NODEV).  _lib.load() therefore imports torch first; checked here on the mapped files."""
    import subprocess
    import sys
    code = ("from desire_amd import _lib; _lib.load(); import torch, re; "
            "m = set(re.findall(r'(/\\S*libamdhip64\\S*)', open('/proc/self/maps').read())); "
            "import os; print(len({os.path.realpath(p) for p in m}))")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().splitlines()[-1] == "1", p.stdout


def test_per_file_flags_are_part_of_the_build_hash(monkeypatch):
    """A library built with other per-file code-generation flags (desire_amd/_build.py FILE_FLAGS, round 5) is another build: the hash the
    library reports (desire_build_hash) and the tree's must then differ, so a stale object cannot pass for the tree."""
    from desire_amd import _build
    h0 = _build.source_hash()
    flags = dict(_build.FILE_FLAGS)
    flags["kernels_rnn.hip"] = ["-mllvm", "-sink-insts-to-avoid-spills"]
    monkeypatch.setattr(_build, "FILE_FLAGS", flags)
    assert _build.source_hash() != h0
    monkeypatch.setattr(_build, "FILE_FLAGS", {})
    assert _build.source_hash() != h0
