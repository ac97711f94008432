// cluster.h -- hand-off between the workgroups of one (scene, k) group (cluster-form IOC kernels).  The per-XCD L2s are not
// coherent and a CU's L1 is never refreshed by another CU's stores, so every hand-off is: plain stores -> s_waitcnt vmcnt(0)
// -> __syncthreads -> ONE lane agent-scope release -> relaxed agent-scope add on the group's arrival counter; consumers poll
// that one word relaxed, then one agent-scope acquire, __syncthreads, plain loads (MI355X_MICROARCH.md, inter-workgroup
// visibility).  Every spin is bounded and reports through an error word.
#pragma once
#include "common.h"

__device__ __forceinline__ bool group_wait(int* cnt, int target, int* err) {
    bool ok = true;
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 40000000L) { atomicOr(err, 1); ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__device__ __forceinline__ void group_publish(int* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

