#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for V in "-DSTEP_PW=0" "-DSTEP_PW=4" "-DSTEP_PW=2"; do
  DESIRE_HIPCC_FLAGS="$V" python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
  DESIRE_HIPCC_FLAGS="$V" python profiles/ab/x2_dump.py "/tmp/pw_$(echo $V | tr -c 'A-Za-z0-9' '_').npz"
  DESIRE_HIPCC_FLAGS="$V" python profiles/ab/run_leg.py config3_shape 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('$V', {k:(round(v['ms_per_step'],2), round(v.get('ioc_ms',0),2)) for k,v in j.items() if isinstance(v,dict) and 'x3' in k})"
done
python -c "
import numpy as np, glob
fs=sorted(glob.glob('/tmp/pw_*.npz')); ref=np.load(fs[0])
for f in fs[1:]:
    b=np.load(f); print(f, {k: ('same' if np.array_equal(ref[k], b[k]) else '%.2e' % np.abs(ref[k]-b[k]).max()) for k in ref.files})"
python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
