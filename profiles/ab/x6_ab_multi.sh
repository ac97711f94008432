#!/bin/bash
# same-box A/B of kernels_x6.hip build variants on the split-operand headline shape (sample generation runs on the six-product kernels)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run2() { for i in 1 2; do python bench.py --split --steps 8 --warmup 3 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('   step %.3f' % d['ms_per_step'], ' '.join('%s %.3f' % (n, k[n]) for n in ('deconv1','mask_fc','deconv2','deconv3','decoder') if n in k))"; done; }
[ -n "$NOTEST" ] || (timeout 900 python -m pytest tests/test_gpu_split.py -q -m gpu -x -n 3 2>&1 | tail -2)
echo "default"; run2
for v in "$@"; do
  export DESIRE_FILE_FLAGS="kernels_x6.hip=$v"
  python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
  echo "variant $v"; run2
done
unset DESIRE_FILE_FLAGS
python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -1
