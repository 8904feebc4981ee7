// dasr_common.h — shared helpers of the HIP kernel sources (gfx950 only).
//
// Every .hip file in this directory is compiled by hipcc --offload-arch=gfx950 into
// libdasr_hip.so.  The same sources are compiled a second time as host C++ against
// tests/hipemu/hipemu.h (-DDASR_HIPEMU) so that unit tests can execute the kernels on the
// CPU of the GPU-less build container; that build is test infrastructure, never shipped.
#pragma once

#ifdef DASR_HIPEMU
#include "hipemu.h"
#define DASR_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
#define DASR_DYN_SMEM(name) char* name = hipemu::dyn_smem
#define DASR_DEVICE_BUILD 0
#define DASR_UNIFORM(x) (x)
#define DASR_MUL24(a, b) ((a) * (b))
#define DASR_SCHED_BARRIER() ((void)0)
#define DASR_WAVE_SYNC() hipemu::wave_barrier()
// LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane from a per-lane global address to wave-uniform LDS base + 16 * lane.
// The emulator copies synchronously, so the counted waits are no-ops there and the barrier is the block barrier.
typedef char* dasr_lds_addr_t;
#define DASR_LDS_ADDR(p) ((char*)(p))
#define DASR_GLDS16(gsrc, lds_base) memcpy((lds_base) + 16 * hipemu::lane_id(), (const void*)(gsrc), 16)
#define DASR_WAIT_VM(n) ((void)0)
#define DASR_RAW_BARRIER() hipemu::block_barrier()
#define DASR_LDS_BARRIER() hipemu::block_barrier()
#define DASR_SETPRIO(n) ((void)0)
#define DASR_SCHED_GROUP(mask, n) ((void)0)
#define DASR_DEVICE_CONST static const
#else
#include <hip/hip_runtime.h>
#define DASR_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (hipStream_t)(stream), __VA_ARGS__)
#define DASR_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define DASR_DEVICE_BUILD 1
// a value the program knows to be the same in every lane of the wave: move it to an SGPR
#define DASR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define DASR_MUL24(a, b) __mul24((a), (b))      /* operands below 2^23 in magnitude: v_mul_i32_i24, full rate */
#define DASR_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// Ordering point between LDS accesses of ONE wave that communicate across its lanes (wave-private LDS slices): the
// hardware executes a wave's DS instructions in issue order, so no s_barrier is needed - only that the compiler keeps the
// program order (fence) and, on the CPU emulator, that the wave's fibers meet
#define DASR_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// LDS-DMA, hidden from the compiler on purpose (cdna_hip_programming.md, "Pipelining across barriers"): behind the
// builtin hipcc waits vmcnt(0) before the next LDS read that may alias, which serialises every prefetch; in inline asm
// the transfer is invisible to its counters and the kernel counts it by hand (DASR_WAIT_VM) - every wave must then issue
// the SAME number of vector-memory operations between two waits (EXEC full, no skipped pieces).  M0 carries the LDS
// base and is compiler-reserved: saved and restored inside the statement.
typedef unsigned dasr_lds_addr_t;
#define DASR_LDS_ADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) char*)(p))
__device__ __forceinline__ void dasr_glds16(const void* gsrc, unsigned lds_base_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_base_uniform) : "memory");
}
#define DASR_GLDS16(gsrc, lds_base) dasr_glds16((const void*)(gsrc), __builtin_amdgcn_readfirstlane(lds_base))
#define DASR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define DASR_RAW_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// the same after this wave's LDS operations have completed (LDS written by one wave, read by another) - and nothing else:
// __syncthreads() would also drain vmcnt, i.e. the LDS-DMA prefetches the kernel keeps in flight on purpose
#define DASR_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define DASR_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// scheduling hint: the next n instructions of the masked kind (0x8 MFMA, 0x100 LDS read, 0x2 VALU) go here, in this order
#define DASR_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define DASR_DEVICE_CONST __device__ const
#endif

#include "../../include/dasr.h"

#define DASR_CHECK_PTR(p) \
    do {                  \
        if (!(p)) return DASR_E_NULL; \
    } while (0)
#define DASR_CHECK_SHAPE(cond) \
    do {                       \
        if (!(cond)) return DASR_E_SHAPE; \
    } while (0)
#define DASR_RETURN_LAUNCH_STATUS() return (int)hipGetLastError()

static inline unsigned dasr_cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
// grid size for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, no more
static inline unsigned dasr_ew_grid(size_t n, unsigned block = 256) {
    size_t g = (n + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---- max |.| bookkeeping for the fp16 x 2 split convolutions (conv_split_bf16.hip).  A kernel that PRODUCES a tensor one of
// them reads keeps a running maximum of the magnitudes it stores (one register per lane); at its end every WORKGROUP stores
// its maximum into its own word of the caller's buffer - no atomics (same-address atomics cost ~10 ns EACH on the MI355X,
// measured: one per wave made the 30 us mask layer 104 us), nothing to clear beforehand.  Layout of an `amax` buffer
// (DASR_AMAX_FLOATS floats): word 0 = n, the number of partial maxima that follow (an int, written by workgroup 0),
// words 1 .. n = one partial maximum per workgroup of the producing launch (n <= DASR_AMAX_MAX_PARTS); the consumer takes
// the maximum of them in its prologue (sp_amax_read).  Values a kernel computes for clamped / shadow lanes are duplicates
// of stored ones and may take part.
__device__ __forceinline__ float dasr_amax1(float m, float v) { return fmaxf(m, fabsf(v)); }
__device__ __forceinline__ float dasr_amax4(float m, float4 o) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
}
// EVERY thread of the workgroup must arrive (no early-exited lanes or waves).  s_part: >= 16 floats of LDS nothing else is
// using at that point; wg / nwg: this workgroup's index in, and the size of, the flattened grid.
__device__ __forceinline__ void dasr_amax_commit(float* amax, float m, float* s_part, unsigned wg, unsigned nwg) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __syncthreads();                                   // (s_part may alias a buffer the workgroup was still reading)
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)((blockDim.x * blockDim.y * blockDim.z + 63) >> 6);
        float r = s_part[0];
        for (int w = 1; w < nw; ++w) r = fmaxf(r, s_part[w]);
        amax[1 + wg] = r;
        if (wg == 0) {
            const int n = (int)nwg;
            memcpy(amax, &n, 4);
        }
    }
}
// a workgroup that leaves before doing any work still owns a word
__device__ __forceinline__ void dasr_amax_commit_idle(float* amax, unsigned wg, unsigned nwg) {
    if (threadIdx.x == 0) {
        amax[1 + wg] = 0.f;
        if (wg == 0) {
            const int n = (int)nwg;
            memcpy(amax, &n, 4);
        }
    }
}
__device__ __forceinline__ unsigned dasr_flat_wg() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }
__device__ __forceinline__ unsigned dasr_flat_nwg() { return gridDim.x * gridDim.y * gridDim.z; }

__device__ __forceinline__ float dasr_act(float v, int act) {
    if (act == DASR_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DASR_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
    return v;
}
// derivative wrt the pre-activation, evaluated from the saved OUTPUT (sign-preserving activations)
__device__ __forceinline__ float dasr_act_grad_from_out(float out, int act) {
    if (act == DASR_ACT_RELU) return out > 0.f ? 1.f : 0.f;
    if (act == DASR_ACT_LRELU02) return out > 0.f ? 1.f : 0.2f;
    return 1.f;
}

// Closed form of the two back-to-back InstanceNorm2d(affine=False, eps):
//   xhat = (x - mean) * s(var),  s(v) = (v+eps)^-1/2 * (v/(v+eps) + eps)^-1/2
__device__ __forceinline__ float dasr_double_in_scale(float var, float eps) {
    float r1 = 1.0f / sqrtf(var + eps);
    float v2 = var / (var + eps);
    return r1 / sqrtf(v2 + eps);
}
// ds/dv of the above
__device__ __forceinline__ float dasr_double_in_dscale(float var, float eps) {
    float a = var + eps;
    float v2 = var / a;
    float b = v2 + eps;
    float s = (1.0f / sqrtf(a)) / sqrtf(b);
    return s * (-0.5f / a - 0.5f * (eps / (a * a)) / b);
}
