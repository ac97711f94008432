# Round-5 profile set (run on the GPU box: gpurun -- 'bash profiles/collect_r05.sh [section ...]'; no section = all).
# Every pass profiles `bench.py --headline-only` (the timed region and nothing else), so a kernel symbol appears at ONE launch size;
# the summaries are keyed per launch class anyway (symbol, workgroups, workgroup size).  Kernel-trace passes and PMC passes are
# separate rocprofv3 runs (PMC never together with sys / runtime / hip tracing).  Output: gpurun_out/prof_r05/; the files quoted in
# DESIGN.md / bench.py are copied into profiles/ with the prefix r05_.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05
mkdir -p $O
want() { [ -z "$SECTIONS" ] || echo " $SECTIONS " | grep -q " $1 "; }
SECTIONS="$*"
stats() {   # tag, bench args...
  tag=$1; shift
  python $R/bench.py "$@" --headline-only > $O/${tag}_bench.json 2>/dev/null
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
if want full; then
  python $R/bench.py --steps 20 --warmup 5 > $O/bench_default_full.json 2>/dev/null      # the driver's line: headline + alt legs + sdd leg + cpu baseline
fi
if want x6; then
  stats bench_w512_x6 --x6 --steps 5 --warmup 2
  pmc   bench_w512_x6 --x6
  stats bench_w512_split --split --steps 5 --warmup 2
  pmc   bench_w512_split --split
fi
if want bf16; then
  stats bench_bf16_mno128 --bf16 --mno 128 --windows 32 --steps 5 --warmup 2     # BASELINE configs[2]
  pmc   bench_bf16_mno128 --bf16 --mno 128 --windows 32
  stats bench_w128_bf16 --bf16 --steps 5 --warmup 2
  pmc   bench_w128_bf16 --bf16
fi
if want config3; then
  stats bench_config3_shape --mno 64 --H 256 --K 50 --windows 4 --steps 10 --warmup 3
  pmc   bench_config3_shape --mno 64 --H 256 --K 50 --windows 4
  stats bench_config3_shape_split --mno 64 --H 256 --K 50 --windows 4 --split --steps 10 --warmup 3       # k_ioc_step<256, 16, 32, 2>: one launch per step
  pmc   bench_config3_shape_split --mno 64 --H 256 --K 50 --windows 4 --split
  stats bench_config3_shape_x6 --mno 64 --H 256 --K 50 --windows 4 --x6 --steps 10 --warmup 3
fi
if want train; then
  stats train --train --steps 5 --warmup 2
  pmc   train --train
  stats train_split --train --split --steps 5 --warmup 2
  pmc   train_split --train --split
fi
if want sdd; then       # the timed region on REAL SDD windows: as is, DESIRE_FLAG_COMPACT_ROWS, + DESIRE_FLAG_COMPACT_IOC (slot classes)
  stats bench_sdd --data sdd --steps 5 --warmup 2
  pmc   bench_sdd --data sdd
  stats bench_sdd_compact --data sdd --flags 12 --steps 5 --warmup 2
  pmc   bench_sdd_compact --data sdd --flags 12
  stats train_split_sdd_compact --train --split --data sdd --flags 12 --steps 5 --warmup 2
fi
if want small; then
  ( cd $R && bash profiles/small_batch.sh > /dev/null 2>&1; cp gpurun_out/sb3/summary.json $O/small_batch_w1_w2_w8.json )
fi
ls -la $O
