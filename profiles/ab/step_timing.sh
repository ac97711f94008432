#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
DESIRE_HIPCC_FLAGS="-DSTEP_TIMING" python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -3
DESIRE_HIPCC_FLAGS="-DSTEP_TIMING" python profiles/ab/run_leg.py config3_shape 2>&1 | grep "k_ioc_step" | sort | uniq -c | sort -rn | head -12
python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
