# Round-3 profile set (run on the GPU box: gpurun -- 'bash profiles/collect_r03.sh').  Kernel-trace passes and PMC passes are
# separate rocprofv3 runs (PMC never together with sys / runtime / hip tracing); summaries land in gpurun_out/prof_r03/ and the
# ones quoted in DESIGN.md / bench.py are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
mkdir -p $O
stats() {   # tag, bench args...
  tag=$1; shift
  python $R/bench.py "$@" --data synthetic > $O/${tag}_bench.json 2>/dev/null
  rocprofv3 --kernel-trace -d /tmp/st_$tag -o tr -- python $R/bench.py "$@" --no-cpu-baseline --data synthetic > /dev/null 2>&1
  python $R/profiles/summarise_db.py $(find /tmp/st_$tag -name "*.db" | head -1) > $O/${tag}_kernel_stats.csv
}
pmc() {     # tag, bench args...
  tag=$1; shift
  i=0
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TA_BUSY_avr GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${tag}_$i -o p$i -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --data synthetic > /dev/null 2>&1
  done
  python $R/profiles/summarise_pmc.py $O/${tag}_pmc_per_kernel.json $(find /tmp/pmc_${tag}_* -name "*.db")
}
python $R/bench.py --steps 20 --warmup 5 > $O/bench_default_full.json 2>/dev/null      # the driver's line: headline + alt legs + sdd leg + cpu baseline
stats bench_w512 --steps 5 --warmup 2 --no-cpu-baseline
pmc   bench_w512
stats bench_w512_x6 --x6 --steps 5 --warmup 2 --no-cpu-baseline            # three-piece operands: IOC + decoder + deconv2 + deconv3 (dims.bf16 = 3)
pmc   bench_w512_x6 --x6
stats bench_bf16_mno128 --bf16 --mno 128 --windows 32 --steps 5 --warmup 2 --no-cpu-baseline     # BASELINE configs[2]
pmc   bench_bf16_mno128 --bf16 --mno 128 --windows 32
stats bench_w128_bf16 --bf16 --steps 5 --warmup 2 --no-cpu-baseline
pmc   bench_w128_bf16 --bf16
stats bench_w512_split --split --steps 5 --warmup 2 --no-cpu-baseline
stats train --train --steps 5 --warmup 2
stats train_split --train --split --steps 5 --warmup 2                      # dims.bf16 = 2: split operands in forward IOC, IOC BPTT, weight-gradient reductions, data-gradient convolutions
stats train_bn2 --train --bn batch --steps 3 --warmup 1
# few windows per call (literal configs[1] = one window): per-kernel ms of every form, with and without hipGraph replay
( cd $R && bash profiles/small_batch.sh > /dev/null 2>&1; cp gpurun_out/sb3/summary.json $O/small_batch_w1_w2_w8.json )
ls -la $O
