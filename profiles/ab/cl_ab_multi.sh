#!/bin/bash
# same-box A/B of several k_ioc_bf16_cl build variants: default, then each argument (extra flags for kernels_bf16_cl.hip, comma-separated), default again
cd ${GRAFT_REPO_ROOT:-/root/repo}
run3() { for i in 1 2 3; do python bench.py --bf16 --mno 128 --windows 32 --steps 10 --warmup 3 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ioc %.3f  step %.3f' % (d['kernel_ms']['ioc'], d['ms_per_step']))"; done; }
echo "default"; run3
for v in "$@"; do
  export DESIRE_FILE_FLAGS="kernels_bf16_cl.hip=-mllvm,-sink-insts-to-avoid-spills,$v"
  python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
  echo "variant $v"; run3
done
unset DESIRE_FILE_FLAGS
python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
echo "default again"; run3
