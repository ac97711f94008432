#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
DESIRE_HIPCC_FLAGS="-DSTEP_X2_IMAGES=0" python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
DESIRE_HIPCC_FLAGS="-DSTEP_X2_IMAGES=0" python profiles/ab/x2_dump.py /tmp/old.npz
DESIRE_HIPCC_FLAGS="-DSTEP_X2_IMAGES=0" python profiles/ab/run_leg.py config3_shape 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('on-the-fly split', {k:(round(v['ms_per_step'],2), round(v.get('ioc_ms',0),2)) for k,v in j.items() if isinstance(v,dict)})"
python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
python profiles/ab/x2_dump.py /tmp/new.npz
python profiles/ab/run_leg.py config3_shape 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('piece images    ', {k:(round(v['ms_per_step'],2), round(v.get('ioc_ms',0),2)) for k,v in j.items() if isinstance(v,dict)})"
python -c "
import numpy as np
a=np.load('/tmp/old.npz'); b=np.load('/tmp/new.npz')
for k in a.files: print(k, 'bit-identical' if np.array_equal(a[k], b[k]) else 'DIFF %.3e' % np.abs(a[k]-b[k]).max(), float(np.abs(a[k]).max()))"
python profiles/ab/x2_dump.py /tmp/new2.npz
python -c "
import numpy as np
a=np.load('/tmp/new.npz'); b=np.load('/tmp/new2.npz')
print('run-to-run:', {k: bool(np.array_equal(a[k], b[k])) for k in a.files})"
