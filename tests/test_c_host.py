"""The C ABI driven from plain C: tests/c_host/host_forward.c is compiled with gcc against include/desire_hip.h (the header
is valid C, every symbol links) and, on the GPU box, run on a blob of dims + weights + inputs; its trajectories must equal
the ctypes path bit for bit and match the oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from desire_amd import _lib
from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "host_forward.c")
ROCM = "/opt/rocm"


def _build(tmp_path):
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(ROCM, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc / HIP headers not available")
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdesire_hip.so not built")
    exe = str(tmp_path / "host_forward")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROCM, "include"), SRC, "-o", exe, _lib.LIB_PATH, "-L", os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_links(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_forward_from_c_equals_the_ctypes_path(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from oracle import desire_oracle as O
    exe = _build(tmp_path)
    d = small_dims(K=3, T_pred=9)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6)
    blob = tmp_path / "in.blob"
    with open(blob, "wb") as f:
        f.write(bytes(_lib.DesireDims.from_dims(d)))
        f.write(struct.pack("<i", len(w)))
        for name, v in w.items():
            nb = name.encode()
            f.write(struct.pack("<i", len(nb))); f.write(nb)
            f.write(struct.pack("<q", v.size)); f.write(np.ascontiguousarray(v, np.float32).tobytes())
        for arr in (past, fut, eps, grids):
            f.write(np.ascontiguousarray(arr, np.float32).tobytes())
        f.write(np.ascontiguousarray(gos, np.int32).tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([exe, str(blob), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = np.fromfile(out, np.float32)
    Yc, sc = res[:d.R * d.T_pred * 2].reshape(d.R, d.T_pred, 2), res[d.R * d.T_pred * 2:]
    # the same call sequence through ctypes + torch-owned buffers
    h = _lib.Handle(d)
    h.set_weights(w)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    p_t, f_t, e_t, g_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(g_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); score = torch.zeros((d.R,), device="cuda")
    h.forward(p_t.data_ptr(), f_t.data_ptr(), e_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(Yc, Y.cpu().numpy())
    np.testing.assert_array_equal(sc, score.cpu().numpy())
    ref = O.forward(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    assert np.abs(ref["Y0"] - h.read_buffer("Y0", (d.R, d.T_pred, 2))).max() < 1e-3
