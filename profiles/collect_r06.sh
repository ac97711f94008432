# Round-6 profile set (run on the GPU box: gpurun -- 'bash profiles/collect_r06.sh [section ...]'; no section = all).
# As in round 5: every pass profiles `bench.py --headline-only` (the timed region and nothing else), kernel-trace passes and PMC passes are separate
# rocprofv3 runs, summaries are keyed per launch class (symbol, workgroups, workgroup size).  bench.py now prints two lines (`#full ...` and the driver's
# compact line LAST): the *_bench.json files keep the last one.  Output: gpurun_out/prof_r06/; what DESIGN.md / bench.py quote is copied to profiles/r06_*.
# Section `slow`: the GPU tests kept out of the default `pytest -m gpu` run (tests/conftest.py: marker `slow`).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r06
mkdir -p $O
want() { [ -z "$SECTIONS" ] || echo " $SECTIONS " | grep -q " $1 "; }
SECTIONS="$*"
stats() {   # tag, bench args...
  tag=$1; shift
  python $R/bench.py "$@" --headline-only 2>/dev/null | tail -n 1 > $O/${tag}_bench.json
  rm -rf /tmp/st_$tag
  rocprofv3 --kernel-trace -d /tmp/st_$tag -o tr -- python $R/bench.py "$@" --headline-only > /dev/null 2>&1
  python $R/profiles/summarise_db.py $(find /tmp/st_$tag -name "*.db" | head -1) > $O/${tag}_kernel_stats.csv
}
pmc() {     # tag, bench args...
  tag=$1; shift
  i=0
  rm -rf /tmp/pmc_${tag}_*
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${tag}_$i -o p$i -- python $R/bench.py "$@" --steps 2 --warmup 1 --headline-only > /dev/null 2>&1
  done
  python $R/profiles/summarise_pmc.py $O/${tag}_pmc_per_kernel.json $(find /tmp/pmc_${tag}_* -name "*.db")
}
if want headline; then
  stats bench_w512 --steps 10 --warmup 3            # the driver's default shape: calls = steps + warmup per symbol
  pmc   bench_w512
  stats bench_w128 --windows 128 --steps 10 --warmup 3
  pmc   bench_w128 --windows 128
fi
if want full; then      # the driver's command: `#full` record + the compact line; both kept
  ( cd $R && python bench.py --steps 20 --warmup 5 > $O/bench_default_stdout.txt 2>/dev/null; tail -n 1 $O/bench_default_stdout.txt > $O/bench_default_line.json; cp bench_full.json $O/bench_default_full.json )
fi
if want sdd; then       # the timed region on REAL SDD windows with both compaction bits: device-side counts (the default) and the read-back A/B is in the full record
  stats bench_sdd_compact --data sdd --flags 12 --steps 5 --warmup 2
  pmc   bench_sdd_compact --data sdd --flags 12
fi
if want bf16; then
  stats bench_bf16_mno128 --bf16 --mno 128 --windows 32 --steps 5 --warmup 2     # BASELINE configs[2]
  pmc   bench_bf16_mno128 --bf16 --mno 128 --windows 32
fi
if want config3; then
  stats bench_config3_shape_split --mno 64 --H 256 --K 50 --windows 4 --split --steps 10 --warmup 3
fi
if want x6; then        # the six-product / split sample-generation kernels got tile-stride loops this round (device-side counts): same time as r05 expected
  stats bench_w512_split --split --steps 5 --warmup 2
  stats bench_w512_x6 --x6 --steps 5 --warmup 2
fi
if want train; then
  stats train_split --train --split --steps 5 --warmup 2
  pmc   train_split --train --split
fi
if want slow; then
  ( cd $R && DESIRE_SLOW_TESTS=1 timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q -k "eight_ranks" > $O/slow_tests.log 2>&1; tail -3 $O/slow_tests.log )
fi
ls -la $O
