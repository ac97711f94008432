set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for W in 128 512; do
  tag=w$W
  python $R/bench.py --windows $W --steps 5 --warmup 2 > $R/gpurun_out/bench_$tag.json 2>/dev/null
  rocprofv3 --kernel-trace -d /tmp/st_$tag -o tr -- python $R/bench.py --windows $W --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/profiles/summarise_db.py $(find /tmp/st_$tag -name "*.db" | head -1) > $R/gpurun_out/stats_$tag.csv
  i=0
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${tag}_$i -o p$i -- python $R/bench.py --windows $W --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  done
  python $R/profiles/summarise_pmc.py $R/gpurun_out/pmc_$tag.json $(find /tmp/pmc_${tag}_* -name "*.db")
done
ls -la $R/gpurun_out/
