#!/bin/bash
# One compact line per launch class: same-box A/B of whole-path and IOC-kernel times (run it on two builds, compare).
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { tag=$1; shift; python bench.py "$@" --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('$tag', 'ms/step %.3f' % d['ms_per_step'], 'min %.3f' % d.get('step_ms_min', 0), 'ioc %.3f' % k.get('ioc', 0), ' '.join('%s %.2f' % (n, k[n]) for n in ('ioc_bwd','decoder','deconv2','deconv34') if n in k))"; }
line fp32_w512 --steps 10 --warmup 3
line split_w512 --split --steps 8 --warmup 3
line x6_w512 --x6 --steps 8 --warmup 3
line bf16_mno32 --bf16 --windows 128 --steps 10 --warmup 3
line bf16_mno128 --bf16 --mno 128 --windows 32 --steps 10 --warmup 3
line train_fp32 --train --steps 6 --warmup 2
line train_split --train --split --steps 6 --warmup 2
line sdd_compact --data sdd --flags 12 --steps 8 --warmup 3
line sdd_compact_split --data sdd --flags 12 --split --steps 8 --warmup 3
line config3_split --mno 64 --H 256 --K 50 --windows 4 --split --steps 10 --warmup 3
