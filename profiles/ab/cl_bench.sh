#!/bin/bash
# bf16 cluster IOC kernel: the parity tests that cover it (skip with NOTEST=1) and the configs[2] bench line, three times
cd ${GRAFT_REPO_ROOT:-/root/repo}
[ -n "$NOTEST" ] || timeout 600 python -m pytest tests/test_gpu_config2.py tests/test_gpu_bf16.py -q -m gpu -x -n 3 2>&1 | tail -2
for i in 1 2 3; do python bench.py --bf16 --mno 128 --windows 32 --steps 10 --warmup 3 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mno128 ms', d['ms_per_step'], 'ioc', d['kernel_ms']['ioc'], 'frac', d['roofline']['frac'])"; done
