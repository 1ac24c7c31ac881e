// Device input pipeline: what CT_NpyDataset.__getitem__ does per slice on CPU workers
// (/root/reference/data/ct_dataset.py:25-84), for a whole batch in one or two launches:
//
//   lung window clip[-1000,500] -> [0,1]        (:32-33)
//   safety resize: image bilinear, mask nearest (:40-47, F.interpolate align_corners=False)
//   synchronized hflip / vflip                   (:52-59)
//   rotation by `angle` degrees, image bilinear / mask nearest, zero fill
//                                                (:62-66, torchvision TF.rotate = affine grid + grid_sample)
//   mask = (mask > 0)                            (:82)
//
// The random draws stay on the host (same `random` call order as the reference); the kernel receives the
// per-sample decisions: flags (1 hflip, 2 vflip, 4 rotate) and the 2x3 rescaled inverse affine matrix that
// torchvision's _gen_affine_grid would multiply the base grid with.  HBM-bound: 5 B read + 12 B written
// per output pixel (fp32 image + uint8 mask in, fp32 image + int64 mask out).
#include "common.h"

constexpr int CT_MAX_BATCH = 64;

struct CTAug {
    int flags[CT_MAX_BATCH];
    float theta[CT_MAX_BATCH][6];      // [T00 T01 T10 T11 T20 T21]: grid = [x y 1] * T
};

__device__ __forceinline__ float ct_window(float v) {
    v = fminf(fmaxf(v, -1000.f), 500.f);
    return (v + 1000.f) / 1500.f;
}

// upsample_bilinear2d / upsample_nearest2d source indexing of ATen (align_corners = False)
__device__ __forceinline__ void lin_src(int dst, float scale, int in, int& i0, int& i1, float& l1) {
    float src = scale * (dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// pass A (only when the raw size differs from S): window + resize into tmp
__global__ __launch_bounds__(256) void ct_resize_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                                        int N, int H0, int W0, int S, float* __restrict__ timg,
                                                        uint8_t* __restrict__ tmask) {
    const float sh = (float)H0 / (float)S, sw = (float)W0 / (float)S;
    const long long total = (long long)N * S * S;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = (int)(e % S); const long long r = e / S; const int i = (int)(r % S); const int n = (int)(r / S);
        const float* im = img + (long long)n * H0 * W0;
        const uint8_t* mk = mask + (long long)n * H0 * W0;
        int h0, h1, w0, w1; float lh, lw;
        lin_src(i, sh, H0, h0, h1, lh);
        lin_src(j, sw, W0, w0, w1, lw);
        const float a = (1.f - lw) * ct_window(im[(long long)h0 * W0 + w0]) + lw * ct_window(im[(long long)h0 * W0 + w1]);
        const float b = (1.f - lw) * ct_window(im[(long long)h1 * W0 + w0]) + lw * ct_window(im[(long long)h1 * W0 + w1]);
        timg[e] = (1.f - lh) * a + lh * b;
        int nh = (int)floorf(i * sh); if (nh > H0 - 1) nh = H0 - 1;
        int nw = (int)floorf(j * sw); if (nw > W0 - 1) nw = W0 - 1;
        tmask[e] = mk[(long long)nh * W0 + nw];
    }
}

// pass B: (window) + flips + rotation + mask binarisation.  WINDOW: the source is the raw slice.
template <bool WINDOW>
__global__ __launch_bounds__(256) void ct_augment_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                                         int N, int S, CTAug aug, float* __restrict__ oimg,
                                                         long long* __restrict__ omask) {
    const long long total = (long long)N * S * S;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = (int)(e % S); const long long r = e / S; const int i = (int)(r % S); const int n = (int)(r / S);
        const float* im = img + (long long)n * S * S;
        const uint8_t* mk = mask + (long long)n * S * S;
        const int fl = aug.flags[n];
        const bool hf = fl & 1, vf = fl & 2;
        auto px = [&](int y, int x) -> float {            // flipped image; zero outside (grid_sample padding_mode="zeros")
            if ((unsigned)y >= (unsigned)S || (unsigned)x >= (unsigned)S) return 0.f;
            const float v = im[(long long)(vf ? S - 1 - y : y) * S + (hf ? S - 1 - x : x)];
            return WINDOW ? ct_window(v) : v;
        };
        auto mx = [&](int y, int x) -> int {
            if ((unsigned)y >= (unsigned)S || (unsigned)x >= (unsigned)S) return 0;
            return mk[(long long)(vf ? S - 1 - y : y) * S + (hf ? S - 1 - x : x)];
        };
        float v; int m;
        if (fl & 4) {
            const float* T = aug.theta[n];
            const float bx = -S * 0.5f + 0.5f + (float)j, by = -S * 0.5f + 0.5f + (float)i;   // torchvision base grid
            const float gx = bx * T[0] + by * T[2] + T[4];
            const float gy = bx * T[1] + by * T[3] + T[5];
            const float ix = ((gx + 1.f) * S - 1.f) / 2.f, iy = ((gy + 1.f) * S - 1.f) / 2.f;  // grid_sample unnormalize
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = ix - fx, ty = iy - fy;
            const float nw = (1.f - tx) * (1.f - ty), ne = tx * (1.f - ty), sw = (1.f - tx) * ty, se = tx * ty;
            v = px(y0, x0) * nw + px(y0, x0 + 1) * ne + px(y0 + 1, x0) * sw + px(y0 + 1, x0 + 1) * se;
            m = mx((int)nearbyintf(iy), (int)nearbyintf(ix));
        } else {
            v = px(i, j);
            m = mx(i, j);
        }
        oimg[e] = v;
        omask[e] = m > 0 ? 1 : 0;
    }
}

extern "C" int unet_ct_preprocess(const float* image, const uint8_t* mask, int N, int H0, int W0, int S,
                                  const int* flags, const float* theta, float* tmp_image, uint8_t* tmp_mask,
                                  float* out_image, long long* out_mask, void* stream) {
    UNET_CHECK(image && mask && out_image && out_mask && flags && theta, "ct_preprocess: null argument");
    UNET_CHECK(N >= 1 && N <= CT_MAX_BATCH, "ct_preprocess: batch must be 1..%d (got %d)", CT_MAX_BATCH, N);
    UNET_CHECK(H0 >= 1 && W0 >= 1 && S >= 1 && S <= 16384 && H0 <= 16384 && W0 <= 16384, "ct_preprocess: bad size");
    const bool resize = H0 != S || W0 != S;
    if (resize) UNET_CHECK(tmp_image && tmp_mask, "ct_preprocess: the resize path needs tmp_image / tmp_mask (N*S*S each)");
    CTAug aug;
    for (int n = 0; n < N; ++n) {
        aug.flags[n] = flags[n] & 7;
        for (int k = 0; k < 6; ++k) aug.theta[n][k] = theta[n * 6 + k];
    }
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)N * S * S;
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192;
    if (resize) {
        hipLaunchKernelGGL(ct_resize_kernel, dim3(grid), dim3(256), 0, s, image, mask, N, H0, W0, S, tmp_image, tmp_mask);
        UNET_LAUNCH_CHECK();
        hipLaunchKernelGGL((ct_augment_kernel<false>), dim3(grid), dim3(256), 0, s, (const float*)tmp_image, (const uint8_t*)tmp_mask,
                           N, S, aug, out_image, out_mask);
    } else {
        hipLaunchKernelGGL((ct_augment_kernel<true>), dim3(grid), dim3(256), 0, s, image, mask, N, S, aug, out_image, out_mask);
    }
    UNET_LAUNCH_CHECK();
    return 0;
}
