#!/bin/bash
# A/B of the step-wise IOC kernel's fragment ring depth (STEP_RING) on the GPU box: rebuilds kernels_rnn.hip per variant
cd ${GRAFT_REPO_ROOT:-/root/repo}
for R in 8 4 2 0; do
  DESIRE_HIPCC_FLAGS="-DSTEP_RING=$R" python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
  DESIRE_HIPCC_FLAGS="-DSTEP_RING=$R" python profiles/ab/run_leg.py config3_shape 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); print('STEP_RING=$R', {k:(round(v['ms_per_step'],2), round(v.get('ioc_ms',0),2)) for k,v in j.items() if isinstance(v,dict)})"
done
python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
