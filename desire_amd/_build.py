"""Build libdesire_hip.so (gfx950 only) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object (include/desire_hip.h) loaded through ctypes."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdesire_hip.so")
SOURCES = ["api.hip", "kernels_gemm.hip", "kernels_conv.hip", "kernels_rnn.hip", "kernels_aux.hip", "kernels_compact.hip", "kernels_bwd.hip", "kernels_bwd_x3.hip", "kernels_bwd_cl.hip", "kernels_bf16.hip", "kernels_bf16_cl.hip", "kernels_x3.hip", "kernels_x6.hip", "kernels_x6r2.hip", "train.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]
FLAGS += os.environ.get("DESIRE_HIPCC_FLAGS", "").split()      # e.g. -DDESIRE_IOC_TIMING for the per-phase cycle counters (build_lib(force=True))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "desire_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdesire_hip.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_t = max(hdr_t, os.path.getmtime(os.path.join(HERE, "..", "include", "desire_hip.h")), os.path.getmtime(__file__))

    def cc(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            return obj                       # object newer than its source and every header: keep it
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
