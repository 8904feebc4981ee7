// edge.hip — layout conversion at the API edge, output clamp, nearest resize, elementwise helpers.
// All HBM-bound, grid-stride, one element per thread iteration (3-channel tensors: the wide path
// is the NHWC side, which is written/read as consecutive floats across lanes).
#include "dasr_common.h"
#include "bf16.h"

extern "C" int dasr_version(void) { return 100; }
extern "C" int dasr_is_device_build(void) { return DASR_DEVICE_BUILD; }
extern "C" const char* dasr_error_string(int code) {
    switch (code) {
        case DASR_OK: return "ok";
        case DASR_E_NULL: return "null pointer argument";
        case DASR_E_SHAPE: return "inconsistent or non-positive sizes";
        case DASR_E_UNSUPPORTED: return "unsupported configuration";
        case DASR_E_WORKSPACE: return "workspace too small";
        default: return code > 0 ? "hip runtime error (hipError_t)" : "unknown error";
    }
}

__global__ void k_nchw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t c = i % C, p = (i / C) % HW, b = i / ((size_t)C * HW);
        dst[i] = src[(b * C + c) * HW + p];
    }
}
__global__ void k_nhwc_to_nchw(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i % HW, c = (i / HW) % C, b = i / ((size_t)C * HW);
        dst[i] = src[(b * HW + p) * C + c];
    }
}
extern "C" int dasr_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
    DASR_CHECK_PTR(src); DASR_CHECK_PTR(dst);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && H > 0 && W > 0);
    size_t n = (size_t)B * C * H * W;
    DASR_LAUNCH(k_nchw_to_nhwc, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, src, dst, C, H * W, n);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
    DASR_CHECK_PTR(src); DASR_CHECK_PTR(dst);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && H > 0 && W > 0);
    size_t n = (size_t)B * C * H * W;
    DASR_LAUNCH(k_nhwc_to_nchw, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, src, dst, C, H * W, n);
    DASR_RETURN_LAUNCH_STATUS();
}

// out_nchw[b,c,p] = clamp(y_nhwc[b,p,c]); indexed by the NCHW element so stores are coalesced
__global__ void k_clamp_to_nchw(const float* __restrict__ y, float* __restrict__ out, int C, int HW, size_t n, float lo,
                                float hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i % HW, c = (i / HW) % C, b = i / ((size_t)C * HW);
        float v = y[(b * HW + p) * C + c];
        out[i] = fminf(fmaxf(v, lo), hi);
    }
}
__global__ void k_clamp_to_nchw_bwd(const float* __restrict__ dout, const float* __restrict__ y, float* __restrict__ dy,
                                    int C, int HW, size_t n, float lo, float hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t c = i % C, p = (i / C) % HW, b = i / ((size_t)C * HW);  // i indexes NHWC
        float v = y[i];
        float g = dout[(b * C + c) * HW + p];
        dy[i] = (v >= lo && v <= hi) ? g : 0.f;  // torch.clamp passes the gradient on the closed interval
    }
}
extern "C" int dasr_clamp_to_nchw(const float* y, float* out, int B, int C, int H, int W, float lo, float hi,
                                  void* stream) {
    DASR_CHECK_PTR(y); DASR_CHECK_PTR(out);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && H > 0 && W > 0);
    size_t n = (size_t)B * C * H * W;
    DASR_LAUNCH(k_clamp_to_nchw, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, y, out, C, H * W, n, lo, hi);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_clamp_to_nchw_bwd(const float* dout, const float* y, float* dy, int B, int C, int H, int W, float lo,
                                      float hi, void* stream) {
    DASR_CHECK_PTR(dout); DASR_CHECK_PTR(y); DASR_CHECK_PTR(dy);
    DASR_CHECK_SHAPE(B > 0 && C > 0 && H > 0 && W > 0);
    size_t n = (size_t)B * C * H * W;
    DASR_LAUNCH(k_clamp_to_nchw_bwd, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, dout, y, dy, C, H * W, n, lo, hi);
    DASR_RETURN_LAUNCH_STATUS();
}

// F.interpolate(mode='nearest'): src index = min(floor(dst * (in/out)), in-1), scale in float (ATen
// nearest_neighbor_compute_source_index with scale = in/out).
__global__ void k_resize_nearest(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int H, int W,
                                 size_t n) {
    float sh = (float)h / (float)H, sw = (float)w / (float)W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int X = (int)(i % W), Y = (int)((i / W) % H);
        size_t bc = i / ((size_t)W * H);
        int y = (int)floorf((float)Y * sh); if (y > h - 1) y = h - 1;
        int x = (int)floorf((float)X * sw); if (x > w - 1) x = w - 1;
        dst[i] = src[(bc * h + y) * w + x];
    }
}
extern "C" int dasr_resize_nearest_nchw(const float* src, float* dst, int BC, int h, int w, int H, int W, void* stream) {
    DASR_CHECK_PTR(src); DASR_CHECK_PTR(dst);
    DASR_CHECK_SHAPE(BC > 0 && h > 0 && w > 0 && H > 0 && W > 0);
    size_t n = (size_t)BC * H * W;
    DASR_LAUNCH(k_resize_nearest, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, src, dst, h, w, H, W, n);
    DASR_RETURN_LAUNCH_STATUS();
}

// the same index map on the region bytes of one-hot masks (nearest resize commutes with the one-hot encoding)
__global__ void k_resize_nearest_u8(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int h, int w,
                                    int H, int W) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int Y = blockIdx.y, b = blockIdx.z;
    int y = (int)floorf((float)Y * sh); if (y > h - 1) y = h - 1;
    const unsigned char* row = src + ((size_t)b * h + y) * w;
    unsigned char* out = dst + ((size_t)b * H + Y) * W;
    for (int X = blockIdx.x * blockDim.x + threadIdx.x; X < W; X += gridDim.x * blockDim.x) {
        int x = (int)floorf((float)X * sw); if (x > w - 1) x = w - 1;
        out[X] = row[x];
    }
}
extern "C" int dasr_resize_nearest_u8(const unsigned char* src, unsigned char* dst, int B, int h, int w, int H, int W,
                                      void* stream) {
    DASR_CHECK_PTR(src); DASR_CHECK_PTR(dst);
    DASR_CHECK_SHAPE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && H <= 65535 && B <= 65535);
    DASR_LAUNCH(k_resize_nearest_u8, dim3(dasr_cdiv((size_t)W, 256), H, B), dim3(256), 0, stream, src, dst, h, w, H, W);
    DASR_RETURN_LAUNCH_STATUS();
}

__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}
__global__ void k_accumulate(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}
extern "C" int dasr_add(const float* a, const float* b, float* out, size_t n, void* stream) {
    DASR_CHECK_PTR(a); DASR_CHECK_PTR(b); DASR_CHECK_PTR(out);
    DASR_CHECK_SHAPE(n > 0);
    DASR_LAUNCH(k_add, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, a, b, out, n);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_accumulate(float* dst, const float* src, size_t n, void* stream) {
    DASR_CHECK_PTR(dst); DASR_CHECK_PTR(src);
    DASR_CHECK_SHAPE(n > 0);
    DASR_LAUNCH(k_accumulate, dim3(dasr_ew_grid(n)), dim3(256), 0, stream, dst, src, n);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_copy(float* dst, const float* src, size_t n, void* stream) {
    DASR_CHECK_PTR(dst); DASR_CHECK_PTR(src);
    DASR_CHECK_SHAPE(n > 0);
    return (int)hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

// ---- bf16 activations (mixed-precision path): elementwise helpers on 4-element groups (n % 4 == 0: every activation
// tensor of the net has a channel count that is a multiple of 4), fp32 arithmetic, one rounding on the store
__global__ void k_add_bf16(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = ld4(a + 4 * i), y = ld4(b + 4 * i);
        st4(out + 4 * i, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
    }
}
// dst (fp32 or bf16) += src (bf16 or fp32) / dst = src: the casts at the fp32 encoder <-> bf16 trunk boundary
template <typename TD, typename TS, bool ACC>
__global__ void k_cast4(TD* __restrict__ dst, const TS* __restrict__ src, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = ld4(src + 4 * i);
        if (ACC) {
            const float4 d = ld4(dst + 4 * i);
            v = make_float4(v.x + d.x, v.y + d.y, v.z + d.z, v.w + d.w);
        }
        st4(dst + 4 * i, v);
    }
}
extern "C" int dasr_add_bf16(const unsigned short* a, const unsigned short* b, unsigned short* out, size_t n, void* stream) {
    DASR_CHECK_PTR(a); DASR_CHECK_PTR(b); DASR_CHECK_PTR(out);
    DASR_CHECK_SHAPE(n > 0 && (n % 4) == 0);
    DASR_LAUNCH(k_add_bf16, dim3(dasr_ew_grid(n / 4)), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out,
                n / 4);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_accumulate_bf16(unsigned short* dst, const unsigned short* src, size_t n, void* stream) {
    DASR_CHECK_PTR(dst); DASR_CHECK_PTR(src);
    DASR_CHECK_SHAPE(n > 0 && (n % 4) == 0);
    DASR_LAUNCH((k_cast4<bf16_t, bf16_t, true>), dim3(dasr_ew_grid(n / 4)), dim3(256), 0, stream, (bf16_t*)dst,
                (const bf16_t*)src, n / 4);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_cast_f32_to_bf16(const float* src, unsigned short* dst, size_t n, void* stream) {
    DASR_CHECK_PTR(dst); DASR_CHECK_PTR(src);
    DASR_CHECK_SHAPE(n > 0 && (n % 4) == 0);
    DASR_LAUNCH((k_cast4<bf16_t, float, false>), dim3(dasr_ew_grid(n / 4)), dim3(256), 0, stream, (bf16_t*)dst, src, n / 4);
    DASR_RETURN_LAUNCH_STATUS();
}
extern "C" int dasr_cast_bf16_to_f32(const unsigned short* src, float* dst, int accumulate, size_t n, void* stream) {
    DASR_CHECK_PTR(dst); DASR_CHECK_PTR(src);
    DASR_CHECK_SHAPE(n > 0 && (n % 4) == 0);
    if (accumulate)
        DASR_LAUNCH((k_cast4<float, bf16_t, true>), dim3(dasr_ew_grid(n / 4)), dim3(256), 0, stream, dst, (const bf16_t*)src, n / 4);
    else
        DASR_LAUNCH((k_cast4<float, bf16_t, false>), dim3(dasr_ew_grid(n / 4)), dim3(256), 0, stream, dst, (const bf16_t*)src, n / 4);
    DASR_RETURN_LAUNCH_STATUS();
}
