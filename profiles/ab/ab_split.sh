#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/ab
[ -n "$NOTEST" ] || (timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_bin_edges.py -q -m gpu -x -n 3 2>&1 | tail -2)
line() { tag=$1; shift; python bench.py "$@" --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('$tag', 'ms/step %.3f' % d['ms_per_step'], 'min %.3f' % d.get('step_ms_min', 0), 'ioc %.3f' % k.get('ioc', 0))"; }
line split_w512 --split --steps 8 --warmup 3
line split_w512 --split --steps 8 --warmup 3
line sdd_compact_split --data sdd --flags 12 --split --steps 8 --warmup 3
line train_split --train --split --steps 6 --warmup 2
