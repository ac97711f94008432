"""Build libdesire_hip.so (gfx950 only) in-tree with hipcc.  No torch extension machinery: the
library is a plain C-ABI shared object (include/desire_hip.h) loaded through ctypes."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdesire_hip.so")
SOURCES = ["api.hip", "api_pack.hip", "api_forward.hip", "api_peer.hip", "api_ops.hip", "kernels_gemm.hip", "kernels_conv.hip", "kernels_rnn.hip", "kernels_aux.hip", "kernels_compact.hip", "kernels_bwd.hip", "kernels_bwd_x3.hip", "kernels_bwd_cl.hip", "kernels_bf16.hip", "kernels_bf16_cl.hip", "kernels_x3.hip", "kernels_x6.hip", "kernels_x6r2.hip", "train.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]
FLAGS += os.environ.get("DESIRE_HIPCC_FLAGS", "").split()      # e.g. -DDESIRE_IOC_TIMING for the per-phase cycle counters (build_lib(force=True))
# Per-file code-generation flags (part of source_hash).  -sink-insts-to-avoid-spills: MachineLICM sinks hoisted loop invariants (the
# fragment base addresses and bias splats of the time loops) back into the loop instead of spilling them.  Registers spilled with /
# without it: k_ioc_bf16_cl<128,..,4> 63 / 190 (256 / 540 B of scratch per lane), k_ioc_x3<128> 2 - 15 / 40 - 52.  Same-box A/B (round 5):
# configs[2] IOC 5.7 -> 5.1 ms, split IOC 29.4 -> 28.0 ms (training-mode 9.5 -> 8.6 ms).  Measured neutral (spills gone, time not) and
# therefore left at the default: kernels_rnn (headline k_ioc 6 -> 0 spilled, 79.5 ms both), kernels_bf16, kernels_x6r2, kernels_bwd*.
_SINK = ["-mllvm", "-sink-insts-to-avoid-spills"]
FILE_FLAGS = {f: list(_SINK) for f in ("kernels_bf16_cl.hip", "kernels_x3.hip")}
_extra = os.environ.get("DESIRE_FILE_FLAGS", "")                 # A/B: "file.hip=-mllvm,-x;other.hip=..." replaces the table's entry
for _item in filter(None, _extra.split(";")):
    _f, _, _v = _item.partition("=")
    FILE_FLAGS[_f] = [x for x in _v.split(",") if x]


def _sha(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(HERE, "..", "include", "desire_hip.h")]


def source_hash() -> str:
    """sha256 over every file of csrc/, the public header and the compile flags: what libdesire_hip.so was built FROM.  build_lib compiles its first
    16 hex digits into the library (desire_build_hash()), so `the tested .so == the tree` is checkable (tests/test_abi.py) instead of trusted to
    file times -- built objects travel to the GPU box with the snapshot, git-ignored but not gpurun-ignored."""
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    return _sha(srcs + _headers(), " ".join(FLAGS) + repr(sorted(FILE_FLAGS.items())))[:16]


def _read(path: str) -> str:
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return ""


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Content-addressed incremental build: an object is reused only if the hash of (its source, every header, the flags) equals the stamp written
    next to it when it was compiled; the library is relinked whenever the tree's source_hash differs from the one stamped beside the .so."""
    want = source_hash()
    objdir = os.path.join(HERE, "build")
    lib_stamp = os.path.join(objdir, "lib.stamp")
    if not force and os.path.exists(LIB) and _read(lib_stamp) == want:
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdesire_hip.so")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()

    def cc(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        flags = list(FLAGS) + FILE_FLAGS.get(src, [])
        if src == "api.hip":                      # the library reports the tree it was built from
            flags.append('-DDESIRE_SRC_HASH="%s"' % want)
        key = _sha([os.path.join(CSRC, src)] + hdrs, " ".join(flags))
        if not force and os.path.exists(obj) and _read(obj + ".stamp") == key:
            return obj
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose and r.stderr:
            print(r.stderr)
        with open(obj + ".stamp", "w") as fh:
            fh.write(key)
        return obj

    with ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(lib_stamp, "w") as fh:
        fh.write(want)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
