#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for V in "-DSTEP_X2_IMAGES=0 -DSTEP_RING=0" "-DSTEP_X2_IMAGES=0" ""; do
  DESIRE_HIPCC_FLAGS="$V" python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
  DESIRE_HIPCC_FLAGS="$V" python profiles/ab/x2_dump.py "/tmp/v_$(echo $V | tr -c 'A-Za-z0-9' '_').npz"
done
python -c "
import numpy as np, glob
fs=sorted(glob.glob('/tmp/v_*.npz')); print(fs)
ref=np.load(fs[0])
for f in fs[1:]:
    b=np.load(f); print(f, {k: ('same' if np.array_equal(ref[k], b[k]) else '%.2e' % np.abs(ref[k]-b[k]).max()) for k in ref.files})"
python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
