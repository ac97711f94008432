#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "--train --split" "--train"; do python bench.py $m --steps 6 --warmup 2 --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('$m', 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s %.2f' % (n, v) for n, v in sorted(k.items(), key=lambda x:-x[1])[:14]))"; done
