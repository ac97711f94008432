cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # label, bench args
  lab=$1; shift
  python bench.py "$@" --headline-only 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
}
for v in "" "-DPRIO16_POOL=3 -DPRIO16_GATE=2 -DPRIOX3_POOL=3 -DPRIOX3_GATE=2 -DPRIO32_POOL=2 -DPRIO32_GATE=1" "-DPRIO16_POOL=2 -DPRIO16_GATE=3 -DPRIOX3_POOL=2 -DPRIOX3_GATE=3 -DPRIO32_POOL=1 -DPRIO32_GATE=2" "-DPRIO16_POOL=3 -DPRIOX3_POOL=3 -DPRIO32_POOL=3 -DPRIO32_GATE=2"; do
  export DESIRE_HIPCC_FLAGS="$v"
  python -c "from desire_amd._build import build_lib; build_lib(force=False)" > /dev/null 2>&1
  echo "== flags [$v]"
  run bf16_mno32 --bf16 --steps 10 --warmup 3
  run bf16_mno32 --bf16 --steps 10 --warmup 3
  run split_w512 --split --steps 5 --warmup 2
  run fp32_w512 --steps 5 --warmup 2
  run fp32_w512 --steps 5 --warmup 2
done
