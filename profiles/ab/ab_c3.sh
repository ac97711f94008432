#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { tag=$1; shift; python bench.py "$@" --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('$tag', 'ms/step %.3f' % d['ms_per_step'], 'min %.3f' % d.get('step_ms_min', 0), 'ioc %.3f' % k.get('ioc', 0))"; }
line config3_fp32 --mno 64 --H 256 --K 50 --windows 4 --steps 10 --warmup 3
line config3_fp32 --mno 64 --H 256 --K 50 --windows 4 --steps 10 --warmup 3
line fp32_w512 --steps 10 --warmup 3
line fp32_mno64 --mno 64 --windows 64 --steps 6 --warmup 2
line fp32_w2 --windows 2 --steps 20 --warmup 5
