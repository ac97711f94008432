// cluster.h -- hand-off between the workgroups of one (scene, k) group (cluster-form IOC kernels).  The per-XCD L2s are not
// coherent and a CU's L1 is never refreshed by another CU's stores, so every hand-off is: plain stores -> s_waitcnt vmcnt(0)
// -> __syncthreads -> ONE lane agent-scope release -> relaxed agent-scope add on the group's arrival counter; consumers poll
// that one word relaxed, then one agent-scope acquire, __syncthreads, plain loads (MI355X_MICROARCH.md, inter-workgroup
// visibility).  Every spin is bounded and reports through an error word (system-scope atomic: the word may be mapped host memory).
#pragma once
#include "common.h"

__device__ __forceinline__ bool group_wait(int* cnt, int target, int* err) {
    bool ok = true;
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > 40000000L) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ok = false; break; }
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


// Write-through form of the same hand-off, for payloads moved as naturally aligned 8-byte words: the producer stores them with
// agent-scope relaxed atomic stores (global_store_dwordx2 ... sc1: written through, not left dirty in the XCD's L2), drains them
// (s_waitcnt vmcnt(0), workgroup barrier) and bumps the arrival counter; the consumer polls the counter and reads the payload
// with agent-scope relaxed atomic loads (sc1: served past the CU's L1).  No release / acquire fence on either side -- a release
// writes the whole L2's dirty lines back (1.7-6.5 us per publish) and an acquire invalidates the CU's L1 (1.7-7 us), per step;
// MI355X_MICROARCH.md lists "8-B agent atomics both sides" among the valid forms.
__device__ __forceinline__ void st_agent_u64(void* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint2 ld_agent_u64(const void* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
// 16-byte forms of the same (a relaxed agent-scope __hip_atomic_* lowers to sc1 only up to 8 bytes): buffer accesses with aux = sc1
// through a descriptor over the exchange buffer (wave-uniform), byte offsets in a VGPR.  16-byte write-through stores run at the
// plain-store rate where 8-byte ones cost 2.7x per byte, and 16-byte sc1 loads at 1.4 - 1.9x the 8-byte rate (MI355X_MICROARCH.md)
typedef unsigned v4u32_ __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t agent_buf;
__device__ __forceinline__ agent_buf agent_buffer(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void st_agent_u128(agent_buf r, unsigned byte_off, uint4 v) {
    const v4u32_ x = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ uint4 ld_agent_u128(agent_buf r, unsigned byte_off) {
    const v4u32_ x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    return make_uint4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void group_publish_wt(int* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool group_wait_wt(int* cnt, int target, int* err) {
    bool ok = true;
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 160000000L) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}


// SYSTEM-scope forms for memory another process / another GPU reads or writes (the peer exchange of k_ioc_step): relaxed atomic
// 8-byte / 4-byte accesses are written through to memory and served from memory, whatever the allocation's caching attributes and
// whatever cache maintenance the runtime does (or elides) between kernels of different queues.
__device__ __forceinline__ void st_sys_u64(void* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint2 ld_sys_u64(const void* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ float2 ld_sys_f2(const float* p) { const uint2 v = ld_sys_u64(p); return make_float2(__uint_as_float(v.x), __uint_as_float(v.y)); }
__device__ __forceinline__ float4 ld_sys_f4(const float* p) {
    const uint2 a = ld_sys_u64(p), b = ld_sys_u64(p + 2);
    return make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x), __uint_as_float(b.y));
}
__device__ __forceinline__ unsigned ld_sys_u32(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys_f32(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
