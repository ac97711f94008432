"""Every entry point of include/desire_hip.h called with NULL handles / NULL pointers: a negative code and a message, never a
crash.  Runs in a subprocess so that a segfault would be seen as a failed test, not as a dead test session."""
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent('''
    import ctypes as C, re, sys
    import numpy as np, torch
    sys.path.insert(0, %r)
    from desire_amd import _lib
    from desire_amd.spec import init_weights
    from tests.helpers import small_dims
    lib = _lib.load()
    d = small_dims(n_scenes=1, K=1, T_pred=4)
    h = _lib.Handle(d); h.set_weights(init_weights(d, 0))
    hp = h._h
    src = open(%r).read()
    protos = re.findall(r"^int (desire_\\w+)\\(([^;]*)\\);", src, re.M | re.S)
    z = torch.zeros(1 << 20, device="cuda")
    n_calls = 0
    for name, args in protos:
        if name in ("desire_create", "desire_destroy", "desire_version", "desire_dims_size"):
            continue
        fn = getattr(lib, name)
        params = [a.strip() for a in args.replace("\\n", " ").split(",")]
        def build(null_handle, null_ptrs):
            vals = []
            for p in params:
                if "desire_handle*" in p:
                    vals.append(None if null_handle else hp)
                elif "*" in p:
                    vals.append(None if null_ptrs else C.c_void_p(z.data_ptr()))
                elif "float" in p:
                    vals.append(C.c_float(0.0))
                elif "size_t" in p:
                    vals.append(C.c_size_t(0))
                else:
                    vals.append(C.c_int32(0))
            return vals
        fn.argtypes = None
        for null_handle, null_ptrs in ((True, True), (True, False), (False, True)):
            rc = fn(*build(null_handle, null_ptrs))
            n_calls += 1
            has_ptr = any("*" in p and "desire_handle*" not in p and "stream" not in p for p in params)
            if not null_handle and not has_ptr:
                continue                                     # nothing but the handle, scalars and a stream: a legal call
            assert rc < 0, (name, null_handle, null_ptrs, rc)
            assert len(lib.desire_last_error()) > 0, name
    torch.cuda.synchronize()
    # the handle is still usable afterwards
    h.set_scene_grids(z.data_ptr(), [0])
    print("null sweep ok:", n_calls, "calls over", len(protos), "prototypes")
''')


def test_null_arguments_are_refused_not_dereferenced():
    import os
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = SCRIPT % (root, os.path.join(root, "include", "desire_hip.h"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])
    assert "null sweep ok" in p.stdout


def test_destroy_returns_the_device_memory():
    """Handles own their workspace, packed weights, training buffers and captured graphs: create / use / destroy in a loop
    must not grow the device footprint."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from desire_amd import _lib
    from desire_amd.spec import init_weights
    from tests.helpers import make_case, small_dims
    d = small_dims(n_scenes=4, K=4)
    w = init_weights(d, 0)
    past, fut, eps, grids, gos = make_case(d, seed=1)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")

    def cycle(train, bf16):
        h = _lib.Handle(d.replace(bf16=bf16))
        h.set_weights(w)
        if train:
            h.set_training(True)
        h.set_scene_grids(g.data_ptr(), gos)
        h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
        if train:
            h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
            h.adam_step(1e-4)
        torch.cuda.synchronize()
        h.close()

    for args in ((False, 0), (True, 0), (False, 1)):
        cycle(*args)                                         # first use of each mode: runtime-side one-off allocations
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(24):
        cycle(i % 3 == 1, int(i % 3 == 2))
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 32 << 20, (free0 - free1) / 2**20
