#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { tag=$1; shift; python bench.py "$@" --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{})
print('$tag', 'ms/step %.3f' % d['ms_per_step'], 'ioc %.3f' % k.get('ioc', 0))"; }
line bf16_mno64_tile64 --bf16 --mno 64 --windows 64 --steps 10 --warmup 3
line bf16_mno64_cluster_bins --bf16 --mno 64 --windows 64 --ioc_form 6 --steps 10 --warmup 3
line bf16_mno64_tile64 --bf16 --mno 64 --windows 64 --steps 10 --warmup 3
line bf16_mno64_cluster_bins --bf16 --mno 64 --windows 64 --ioc_form 6 --steps 10 --warmup 3
line bf16_mno96 --bf16 --mno 96 --windows 42 --steps 10 --warmup 3
