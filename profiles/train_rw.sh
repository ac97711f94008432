# FETCH_SIZE / WRITE_SIZE per launch of the BPTT kernels of a training step (two PMC passes, kernel-trace only):
# gpurun -- 'bash profiles/train_rw.sh [--split]'.  KiB counters -> GB; HBM bytes = 2 x FETCH + WRITE (MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_rw_*
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_rw_$i -o p$i -- python $R/bench.py --train "$@" --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
done
python $R/profiles/summarise_pmc.py /tmp/pmc_rw.json $(find /tmp/pmc_rw_* -name "*.db") > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open('/tmp/pmc_rw.json'))
for k, v in d.items():
    if ('decoder_bwd' in k or 'ioc_bwd' in k or 'k_ioc' in k) and v.get('workgroups', 0) >= 2560:
        print(k[:70], 'FETCH GB', round(v['FETCH_SIZE'] / 1e6, 2), 'WRITE GB', round(v['WRITE_SIZE'] / 1e6, 2))
PY
