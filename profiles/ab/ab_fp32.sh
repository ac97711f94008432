#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do python bench.py --steps 10 --warmup 3 --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('fp32_w512 ms/step %.3f min %.3f' % (d['ms_per_step'], d.get('step_ms_min', 0)), ' '.join('%s %.3f' % (n, k[n]) for n in ('ioc','deconv2','deconv3','decoder','deconv4','deconv1','mask_fc') if n in k))"; done
