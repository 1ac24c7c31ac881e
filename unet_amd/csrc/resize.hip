// Bilinear resize with align_corners=True and its adjoint: the branch of UNet.forward that fires when the input size is
// not divisible by 2^levels and an upsampled tensor misses its skip connection's size by a pixel
// (reference models/unet.py:56-57: F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=True)).
// NHWC, thread = (pixel, 16-byte channel group); source index = dst * (in - 1) / (out - 1) in fp32, as ATen computes it
// (area_pixel_compute_scale / compute_source_index).  Never on the path of the shipped configs (256 / 512 / 1024 inputs).
#include "common.h"

__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& lam) {
    const float s = scale * (float)dst;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = s - (float)i0;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ x, int ldx, int xoff, int h, int w,
                                                         T* __restrict__ y, int ldy, int yoff, int N, int H, int W, int C,
                                                         float sh, float sw) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV;
    const long long total = (long long)N * H * W * G;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const int cg = (int)(v % G);
        long long p = v / G;
        const int ox = (int)(p % W); p /= W;
        const int oy = (int)(p % H);
        const int n = (int)(p / H);
        int y0, y1, x0, x1; float ly, lx;
        src_index(oy, sh, h, y0, y1, ly);
        src_index(ox, sw, w, x0, x1, lx);
        const T* b = x + (long long)n * h * w * ldx + xoff + cg * EPV;
        float a00[EPV], a01[EPV], a10[EPV], a11[EPV], o[EPV];
        Vec16<T>::unpack(*(const uint4*)(b + ((long long)y0 * w + x0) * ldx), a00);
        Vec16<T>::unpack(*(const uint4*)(b + ((long long)y0 * w + x1) * ldx), a01);
        Vec16<T>::unpack(*(const uint4*)(b + ((long long)y1 * w + x0) * ldx), a10);
        Vec16<T>::unpack(*(const uint4*)(b + ((long long)y1 * w + x1) * ldx), a11);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = w00 * a00[e] + w01 * a01[e] + w10 * a10[e] + w11 * a11[e];
        *(uint4*)(y + (((long long)n * H + oy) * W + ox) * ldy + yoff + cg * EPV) = Vec16<T>::pack(o);
    }
}

// adjoint as a GATHER (deterministic, no atomics): input pixel (iy, ix) collects from every output pixel whose two source
// rows / columns include it.  Output rows that can touch iy lie in a window of a few rows around iy / scale.
__device__ __forceinline__ void adj_range(int i, float scale, int out_size, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }          // in_size == 1: every output reads input 0
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}
__device__ __forceinline__ float adj_weight(int dst, int i, float scale, int in_size) {
    int i0, i1; float lam;
    src_index(dst, scale, in_size, i0, i1, lam);
    return (i0 == i ? 1.f - lam : 0.f) + (i1 == i ? lam : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, int ldy, int yoff, int H, int W,
                                                         T* __restrict__ dx, int ldx, int xoff, int N, int h, int w, int C,
                                                         float sh, float sw) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV;
    const long long total = (long long)N * h * w * G;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const int cg = (int)(v % G);
        long long p = v / G;
        const int ix = (int)(p % w); p /= w;
        const int iy = (int)(p % h);
        const int n = (int)(p / h);
        int ylo, yhi, xlo, xhi;
        adj_range(iy, sh, H, ylo, yhi);
        adj_range(ix, sw, W, xlo, xhi);
        float acc[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
        const T* b = dy + (long long)n * H * W * ldy + yoff + cg * EPV;
        for (int oy = ylo; oy <= yhi; ++oy) {
            const float wy = adj_weight(oy, iy, sh, h);
            if (wy == 0.f) continue;
            for (int ox = xlo; ox <= xhi; ++ox) {
                const float wx = adj_weight(ox, ix, sw, w);
                if (wx == 0.f) continue;
                float g[EPV];
                Vec16<T>::unpack(*(const uint4*)(b + ((long long)oy * W + ox) * ldy), g);
                // the forward weight of this (output, input) pair, formed as the forward forms it
                const float wt = wy * wx;
#pragma unroll
                for (int e = 0; e < EPV; ++e) acc[e] = fmaf(wt, g[e], acc[e]);
            }
        }
        *(uint4*)(dx + (((long long)n * h + iy) * w + ix) * ldx + xoff + cg * EPV) = Vec16<T>::pack(acc);
    }
}

static int resize_check(int dtype, int C, int ld1, int off1, int ld2, int off2, int N, int a, int b, int c, int d, const char* what) {
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "%s: bad dtype %d", what, dtype);
    UNET_CHECK(C > 0 && C % 8 == 0 && ld1 % 8 == 0 && off1 % 8 == 0 && ld2 % 8 == 0 && off2 % 8 == 0 && C % epv == 0,
               "%s: channels, strides and offsets must be multiples of 8", what);
    UNET_CHECK(N > 0 && a > 0 && b > 0 && c > 0 && d > 0, "%s: empty shape", what);
    return 0;
}
static inline float ac_scale(int in_size, int out_size) { return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f; }
static inline int resize_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

extern "C" int unet_resize_bilinear_fwd(int dtype, const void* x, int ldx, int xoff, int h, int w,
                                        void* y, int ldy, int yoff, int H, int W, int N, int C, void* stream) {
    if (int rc = resize_check(dtype, C, ldx, xoff, ldy, yoff, N, h, w, H, W, "resize_bilinear_fwd")) return rc;
    UNET_CHECK(x && y, "resize_bilinear_fwd: null pointer");
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const int grid = resize_grid((long long)N * H * W * (C / epv));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16)
        hipLaunchKernelGGL((resize_fwd_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, xoff, h, w, (__bf16*)y, ldy, yoff, N, H, W, C, ac_scale(h, H), ac_scale(w, W));
    else
        hipLaunchKernelGGL((resize_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, xoff, h, w, (float*)y, ldy, yoff, N, H, W, C, ac_scale(h, H), ac_scale(w, W));
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_resize_bilinear_bwd(int dtype, const void* dy, int ldy, int yoff, int H, int W,
                                        void* dx, int ldx, int xoff, int h, int w, int N, int C, void* stream) {
    if (int rc = resize_check(dtype, C, ldx, xoff, ldy, yoff, N, h, w, H, W, "resize_bilinear_bwd")) return rc;
    UNET_CHECK(dy && dx, "resize_bilinear_bwd: null pointer");
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const int grid = resize_grid((long long)N * h * w * (C / epv));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16)
        hipLaunchKernelGGL((resize_bwd_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)dy, ldy, yoff, H, W, (__bf16*)dx, ldx, xoff, N, h, w, C, ac_scale(h, H), ac_scale(w, W));
    else
        hipLaunchKernelGGL((resize_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dy, ldy, yoff, H, W, (float*)dx, ldx, xoff, N, h, w, C, ac_scale(h, H), ac_scale(w, W));
    UNET_LAUNCH_CHECK();
    return 0;
}
