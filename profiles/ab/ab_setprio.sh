cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "-DCL_SETPRIO=3 -DCL_SETPRIO2=2" "-DCL_SETPRIO=1 -DCL_SETPRIO2=2 -DCL_SETPRIO3=3" "-DCL_SETPRIO=3 -DCL_SETPRIO2=2 -DCL_SETPRIO3=1" "-DCL_SETPRIO0=1 -DCL_SETPRIO=3 -DCL_SETPRIO2=2" "-DCL_SETPRIO0=0 -DCL_SETPRIO=2 -DCL_SETPRIO2=1 -DCL_SETPRIO3=3" "-DCL_SETPRIO=3 -DCL_SETPRIO2=1" "-DCL_SETPRIO=2 -DCL_SETPRIO2=1"; do
  export DESIRE_HIPCC_FLAGS="$v"
  python -c "from desire_amd._build import build_lib; build_lib(force=False)" > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --bf16 --mno 128 --windows 32 --steps 10 --warmup 3 --headline-only 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags [$v]', round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
  done
done
