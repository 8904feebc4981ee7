// hipemu.h — a small CPU emulator of the HIP execution model, TEST INFRASTRUCTURE ONLY.
//
// The container this repository is developed in has no GPU.  To unit-test the HIP kernel
// sources under csrc/ on the CPU (indexing, barriers, LDS staging, wave shuffles, MFMA
// fragment layouts) they are compiled a second time as plain C++ against this header
// (-DDASR_HIPEMU) into tests/hipemu/libdasr_emu.so, which only tests load.  One workgroup
// runs at a time; its threads are cooperative fibers that switch at __syncthreads() and at
// wave-collective operations (__shfl*, MFMA), so barrier and cross-lane semantics are real.
// The product never loads this library: dasr_amd._lib refuses it unless a test sets
// DASR_HIPEMU_LIB explicitly, and dasr_is_device_build() returns 0 for it.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define warpSize 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float float4_emu __attribute__((ext_vector_type(4)));
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

namespace hipemu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void block_barrier();
void wave_barrier();
int lane_id();
void* wave_buf();  // >= 64 * 64 bytes of per-wave exchange scratch
extern char* dyn_smem;
}  // namespace hipemu

#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
static inline void __syncthreads() { hipemu::block_barrier(); }

template <class T>
static inline T hipemu_xchg(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "exchange slot too small");
    char* buf = (char*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    memcpy(buf + 64 * l, &v, sizeof(T));
    hipemu::wave_barrier();
    T r;
    memcpy(&r, buf + 64 * src_lane, sizeof(T));
    hipemu::wave_barrier();
    return r;
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = hipemu::lane_id();
    int src = l ^ mask;
    if ((src / width) != (l / width)) src = l;
    return hipemu_xchg(v, src);
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l + (int)d;
    if ((src / width) != (l / width)) src = l;
    return hipemu_xchg(v, src);
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::lane_id();
    return hipemu_xchg(v, (l / width) * width + (src % width));
}

static inline int __all(int pred) {
    int* buf = (int*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    buf[l] = 2 + (pred ? 1 : 0);               // 2 marks "this lane voted in this round"
    hipemu::wave_barrier();
    int r = 1;
    for (int i = 0; i < 64; ++i)
        if (buf[i] == 2) r = 0;
    hipemu::wave_barrier();
    buf[l] = 0;                                // exited / absent lanes never vote: their slot stays 0 = ignored
    hipemu::wave_barrier();
    return r;
}
static inline int __any(int pred) { return !__all(!pred); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }

// ---- MFMA (f32 in / f32 accumulate), fragment layouts per cdna_hip_programming.md §3 -------------
static inline f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    float* buf = (float*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    buf[l] = a;
    buf[64 + l] = b;
    hipemu::wave_barrier();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(buf[row + 32 * k], buf[64 + col + 32 * k], acc);
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
static inline f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    float* buf = (float*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    buf[l] = a;
    buf[64 + l] = b;
    hipemu::wave_barrier();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(buf[row + 16 * k], buf[64 + col + 16 * k], acc);
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32

// ---- bf16 MFMA (v_mfma_f32_32x32x16_bf16): A lane (r = l&31, h = l>>5) holds A[row r][k = 8h + j], B lane holds
// B[k = 8h + j][col r], j = 0..7; C/D as the f32 form (cdna_hip_programming.md section 3).  Products of two bf16 values are
// exact in fp32; the accumulation is emulated as an fp32 fma chain in k order.
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline float hipemu_bf2f(__bf16 b) {
    unsigned short h;
    memcpy(&h, &b, 2);
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline f32x16 hipemu_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, f32x16 c, int, int, int) {
    char* buf = (char*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    memcpy(buf + 64 * l, &a, 16);
    memcpy(buf + 64 * l + 16, &b, 16);
    hipemu::wave_barrier();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            __bf16 av, bv;
            memcpy(&av, buf + 64 * (row + 32 * (k >> 3)) + 2 * (k & 7), 2);
            memcpy(&bv, buf + 64 * (col + 32 * (k >> 3)) + 16 + 2 * (k & 7), 2);
            acc = fmaf(hipemu_bf2f(av), hipemu_bf2f(bv), acc);
        }
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu_mfma_f32_32x32x16_bf16
// v_mfma_f32_32x32x16_f16: the same operand and result maps with fp16 elements (products exact in fp32)
typedef _Float16 hipemu_f16x8 __attribute__((ext_vector_type(8)));
static inline f32x16 hipemu_mfma_f32_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, f32x16 c, int, int, int) {
    char* buf = (char*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    memcpy(buf + 64 * l, &a, 16);
    memcpy(buf + 64 * l + 16, &b, 16);
    hipemu::wave_barrier();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 av, bv;
            memcpy(&av, buf + 64 * (row + 32 * (k >> 3)) + 2 * (k & 7), 2);
            memcpy(&bv, buf + 64 * (col + 32 * (k >> 3)) + 16 + 2 * (k & 7), 2);
            acc = fmaf((float)av, (float)bv, acc);
        }
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu_mfma_f32_32x32x16_f16
// v_mfma_f32_16x16x32_bf16: A lane l holds A[row l&15][k = 8(l>>4) + j], B lane holds B[k = 8(l>>4) + j][col l&15];
// C/D: col = lane&15, row = 4*(lane>>4) + reg
static inline f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, f32x4 c, int, int, int) {
    char* buf = (char*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    memcpy(buf + 64 * l, &a, 16);
    memcpy(buf + 64 * l + 16, &b, 16);
    hipemu::wave_barrier();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            __bf16 av, bv;
            memcpy(&av, buf + 64 * (row + 16 * (k >> 3)) + 2 * (k & 7), 2);
            memcpy(&bv, buf + 64 * (col + 16 * (k >> 3)) + 16 + 2 * (k & 7), 2);
            acc = fmaf(hipemu_bf2f(av), hipemu_bf2f(bv), acc);
        }
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_f32_16x16x32_bf16

// ---- ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per group of 16 consecutive lanes, lane 4q+p supplies the address
// of row q, columns 4p..4p+3 of a 4 x 16 block of 16-bit elements; lane i of the group receives column i, row q in its
// element q.
typedef short hipemu_s16x4 __attribute__((ext_vector_type(4)));
static inline hipemu_s16x4 hipemu_ds_read_tr16_b64(const void* addr) {
    char* buf = (char*)hipemu::wave_buf();
    int l = hipemu::lane_id();
    memcpy(buf + 64 * l, &addr, sizeof(void*));
    hipemu::wave_barrier();
    int g = l >> 4, i = l & 15;
    hipemu_s16x4 out;
    for (int q = 0; q < 4; ++q) {
        const char* src;
        memcpy(&src, buf + 64 * (16 * g + 4 * q + (i >> 2)), sizeof(void*));
        short v;
        memcpy(&v, src + 2 * (i & 3), 2);
        out[q] = v;
    }
    hipemu::wave_barrier();
    return out;
}
