cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { lab=$1; shift; python bench.py "$@" --headline-only 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lab', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"; }
for v in "" "-DPRIO32_POOL=3 -DPRIO32_GATE=2" "" "-DPRIO32_POOL=3 -DPRIO32_GATE=2" "-DPRIO32_POOL=3 -DPRIO32_GATE=1" "-DPRIO32_POOL=1 -DPRIO32_GATE=3"; do
  export DESIRE_HIPCC_FLAGS="$v"
  python -c "from desire_amd._build import build_lib; build_lib(force=False)" > /dev/null 2>&1
  echo "== flags [$v]"
  run fp32_w512 --steps 10 --warmup 3
  run fp32_w512 --steps 10 --warmup 3
done
