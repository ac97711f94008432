# Per-call kernel timeline of ONE training step (between two k_adam launches): gpurun -- 'bash profiles/train_timeline.sh [--split] [MIN_MS]'
# (the round-4 findings of DESIGN.md 12a -- which weight-gradient call takes what, what a kernel edit moved -- were read off this)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=""; MIN=0.25
for a in "$@"; do case $a in --split) MODE=--split;; *) MIN=$a;; esac; done
rm -rf /tmp/st_tl
rocprofv3 --kernel-trace -d /tmp/st_tl -o tr -- python $R/bench.py --train $MODE --steps 3 --warmup 1 --headline-only > /dev/null 2>&1
python $R/profiles/summarise_db.py $(find /tmp/st_tl -name "*.db" | head -1) --timeline k_adam | sed -n '/start_ms/,$p' | awk -F, -v m=$MIN 'NR==1 || $2 > m' | cut -c1-150
