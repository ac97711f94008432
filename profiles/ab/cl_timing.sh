#!/bin/bash
# Phase timers of the bf16 cluster IOC kernel (BASELINE configs[2] shape): one workgroup's cycle counters per phase.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/cl
DESIRE_HIPCC_FLAGS="-DDESIRE_IOC_TIMING $CLFLAGS" python -c "from desire_amd._build import build_lib; build_lib()" 2>&1 | tail -3
DESIRE_HIPCC_FLAGS="-DDESIRE_IOC_TIMING $CLFLAGS" python bench.py --bf16 --mno 128 --windows 32 --steps 2 --warmup 1 --headline-only 2>&1 >/dev/null | grep "ioc timing" | tail -12
python -c "from desire_amd._build import build_lib; build_lib()" > /dev/null 2>&1
