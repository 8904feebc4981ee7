// bf16.h — storage-type helpers of the mixed-precision path (BASELINE.json configs[2..3]: bf16 activations).
//
// Arithmetic stays fp32 everywhere (MFMA accumulators, statistics, epilogues); only what lives in HBM between
// kernels is bf16.  Kernels that are byte movers are templates over the storage type T (float or bf16_t) and touch
// memory through ld4 / st4 / ld1 / st1 below; the bf16 trunk convolutions have their own MFMA kernels
// (conv_bf16_mfma.hip, v_mfma_f32_32x32x16_bf16).
#pragma once
#include "dasr_common.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifdef DASR_HIPEMU
// round-to-nearest-even on the bits (the emulator's inputs are finite; NaN handling is the device cast's business)
static inline bf16_t dasr_f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    unsigned short h = (unsigned short)u;
    bf16_t r;
    memcpy(&r, &h, 2);
    return r;
}
static inline float dasr_bf2f(bf16_t b) {
    unsigned short h;
    memcpy(&h, &b, 2);
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#else
// a plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN - MI355X_MICROARCH.md, correctness boundaries)
__device__ __forceinline__ bf16_t dasr_f2bf(float f) { return (bf16_t)f; }
__device__ __forceinline__ float dasr_bf2f(bf16_t b) { return (float)b; }
#endif

// two floats -> two bf16 with one v_cvt_pk_bf16_f32
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#ifdef DASR_HIPEMU
static inline bf16x2_t dasr_f2bf2(float a, float b) {
    bf16x2_t r;
    r[0] = dasr_f2bf(a);
    r[1] = dasr_f2bf(b);
    return r;
}
#else
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x2_t dasr_f2bf2(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_convertvector(v, bf16x2_t);
}
#endif

// ---- 4 consecutive elements <-> float4 -------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return make_float4(dasr_bf2f(v[0]), dasr_bf2f(v[1]), dasr_bf2f(v[2]), dasr_bf2f(v[3]));
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *(float4*)p = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = dasr_f2bf(v.x); o[1] = dasr_f2bf(v.y); o[2] = dasr_f2bf(v.z); o[3] = dasr_f2bf(v.w);
    *(bf16x4*)p = o;
}
// the same 4 elements left unconverted (software-pipelined loops hold the next trip's loads in this form)
template <typename T> struct raw4;
template <> struct raw4<float> { typedef float4 type; };
template <> struct raw4<bf16_t> { typedef bf16x4 type; };
__device__ __forceinline__ float4 ld4_raw(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ bf16x4 ld4_raw(const bf16_t* p) { return *(const bf16x4*)p; }
template <typename T> __device__ __forceinline__ float4 cvt4(typename raw4<T>::type v);
template <> __device__ __forceinline__ float4 cvt4<float>(float4 v) { return v; }
template <> __device__ __forceinline__ float4 cvt4<bf16_t>(bf16x4 v) {
    return make_float4(dasr_bf2f(v[0]), dasr_bf2f(v[1]), dasr_bf2f(v[2]), dasr_bf2f(v[3]));
}
// streaming variants (read once / written once: keep them out of the caches)
template <typename T> __device__ __forceinline__ float4 ld4_nt(const T* p);
template <> __device__ __forceinline__ float4 ld4_nt<float>(const float* p) {
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ float4 ld4_nt<bf16_t>(const bf16_t* p) {
    const s16x4 r = __builtin_nontemporal_load((const s16x4*)p);
    bf16x4 v;
    __builtin_memcpy(&v, &r, 8);
    return make_float4(dasr_bf2f(v[0]), dasr_bf2f(v[1]), dasr_bf2f(v[2]), dasr_bf2f(v[3]));
}
template <typename T> __device__ __forceinline__ void st4_nt(T* p, float4 v);
template <> __device__ __forceinline__ void st4_nt<float>(float* p, float4 v) {
    f32x4 o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, (f32x4*)p);
}
template <> __device__ __forceinline__ void st4_nt<bf16_t>(bf16_t* p, float4 v) {
    bf16x4 o;
    o[0] = dasr_f2bf(v.x); o[1] = dasr_f2bf(v.y); o[2] = dasr_f2bf(v.z); o[3] = dasr_f2bf(v.w);
    s16x4 r;
    __builtin_memcpy(&r, &o, 8);
    __builtin_nontemporal_store(r, (s16x4*)p);
}

// ---- single elements ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return dasr_bf2f(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = dasr_f2bf(v); }

// what the storage type does to a value (the fp32 path: nothing) - used where a kernel must reproduce the rounding
// another kernel applied when it stored the same quantity
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return dasr_bf2f(dasr_f2bf(v)); }

// ---- transposed LDS read (ds_read_b64_tr_b16, gfx950): see the wgrad kernel of conv_bf16_mfma.hip -----------------------
// Per group of 16 consecutive lanes: lane 4q+p passes the address of row q, columns 4p..4p+3 (8-byte aligned) of a
// 4 x 16 block of bf16; lane i of the group gets column i of the 4 rows.  EXEC must be all ones (never call it under
// divergent control flow).
__device__ __forceinline__ bf16x4 lds_read_tr16(const bf16_t* p) {
#ifdef DASR_HIPEMU
    const hipemu_s16x4 r = hipemu_ds_read_tr16_b64((const void*)p);
#else
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
#endif
    bf16x4 v;
    __builtin_memcpy(&v, &r, 8);
    return v;
}

// ---- 16-byte loads through a buffer descriptor -------------------------------------------------------------------------
// buffer_load_dwordx4 with a 128-bit resource (base, num_records): the address is base + a 32-bit byte offset and the
// hardware range-checks it - an offset at or beyond num_records returns zeros.  The staging loops of the bf16 convolutions
// use that as the convolution's zero padding: per-piece offsets are computed once per tile, pieces outside the image get
// DASR_OOB, and the per-chunk loop is "add the chunk offset, load" with no index arithmetic, compares or branches
// (the predicated global loads it replaces cost ~25 VALU instructions and an exec-mask branch per piece per chunk,
// more than the MFMA work of a 32-channel bf16 chunk).  The descriptor must be built from wave-uniform values.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define DASR_OOB 0x80000000u       // + any chunk offset stays out of range: tensors addressed this way are < 2 GiB
#ifdef DASR_HIPEMU
struct BufRsrc {
    const char* base;
    size_t bytes;
};
static inline BufRsrc dasr_make_rsrc(const void* p, size_t bytes) { return BufRsrc{(const char*)p, bytes}; }
static inline u32x4_t dasr_buffer_load16(const BufRsrc& r, unsigned off) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if ((size_t)off + 16 <= r.bytes) memcpy(&v, r.base + off, 16);
    return v;
}
#else
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc dasr_make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4_t dasr_buffer_load16(BufRsrc r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
#endif
