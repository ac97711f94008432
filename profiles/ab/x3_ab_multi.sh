#!/bin/bash
# same-box A/B of several kernels_x3.hip build variants on the split-operand headline shape
cd ${GRAFT_REPO_ROOT:-/root/repo}
run2() { for i in 1 2; do python bench.py --split --steps 8 --warmup 3 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ioc %.3f  step %.3f' % (d['kernel_ms']['ioc'], d['ms_per_step']))"; done; }
echo "default"; run2
for v in "$@"; do
  export DESIRE_FILE_FLAGS="kernels_x3.hip=-mllvm,-sink-insts-to-avoid-spills,$v"
  python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
  echo "variant $v"; run2
done
unset DESIRE_FILE_FLAGS
python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
echo "default again"; run2
