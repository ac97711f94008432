#!/bin/bash
# A/B helper of the bf16 cluster IOC kernel: parity tests that cover it, the configs[2] bench line, then the phase timers.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/cl
timeout 600 python -m pytest tests/test_gpu_config2.py tests/test_gpu_bf16.py -q -m gpu -x -n 3 2>&1 | tail -4
for i in 1 2; do python bench.py --bf16 --mno 128 --windows 32 --steps 10 --warmup 3 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mno128 ms', d['ms_per_step'], 'ioc', d['kernel_ms']['ioc'], 'frac', d['roofline']['frac'])"; done
python bench.py --bf16 --steps 10 --warmup 3 --windows 128 --headline-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mno32 ms', d['ms_per_step'], 'ioc', d['kernel_ms']['ioc'], 'frac', d['roofline']['frac'])"
bash profiles/ab/cl_timing.sh
