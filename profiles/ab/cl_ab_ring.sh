#!/bin/bash
# same-box A/B of a k_ioc_bf16_cl build variant ($1 = extra flags for kernels_bf16_cl.hip, e.g. -DCL_HRING=4): default, variant (rebuilt on the box), default again
cd ${GRAFT_REPO_ROOT:-/root/repo}
run3() { for i in 1 2 3; do python bench.py --bf16 --mno 128 --windows 32 --steps 10 --warmup 3 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ioc %.3f  step %.3f' % (d['kernel_ms']['ioc'], d['ms_per_step']))"; done; }
echo "default"; run3
export DESIRE_FILE_FLAGS="kernels_bf16_cl.hip=-mllvm,-sink-insts-to-avoid-spills,$1"
python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
echo "variant $1"; run3
unset DESIRE_FILE_FLAGS
python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
echo "default again"; run3
