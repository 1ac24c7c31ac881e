// Fused Dice + cross-entropy loss (forward totals, backward), argmax, per-pixel ensemble fusion and
// the confusion counts behind the segmentation metrics (gfx950, bandwidth-bound).
//
// Reference semantics followed:
//   dice_loss / CEDiceLoss     /root/reference/losses/segmentation.py:5-27
//   logits.argmax(1)           /root/reference/trainer.py:119, utils/eval_utils.py:103
//   entropy-weighted fusion    /root/reference/eval_ensemble.py:9-11,77-99 (per pixel)
//   metric counts              /root/reference/utils/eval_utils.py:16-40
//
// logits are NCHW fp32 (what the final conv writes and what callers see); targets are int64.
#include "common.h"

// acc layout (fp64): [n][c][3] = {I = sum p*[t==c], P = sum p, T = sum [t==c]}, then [nll_sum, valid_cnt, bad_cnt]
template <int C>
__global__ __launch_bounds__(256) void dice_ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          long long HW, long long ignore_index, int ce_only,
                                                          double* __restrict__ acc, int N) {
    const int n = blockIdx.y;
    const float* lg = logits + (size_t)n * C * HW;
    const long long* tg = target + (size_t)n * HW;
    float sI[C], sP[C], sT[C], nll = 0.f, cnt = 0.f, bad = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) sI[c] = sP[c] = sT[c] = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        float l[C], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) { l[c] = lg[c * HW + i]; mx = fmaxf(mx, l[c]); }
        float s = 0.f, e[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { e[c] = expf(l[c] - mx); s += e[c]; }
        const float inv = 1.f / s;
        const long long t = tg[i];
        const bool in_range = t >= 0 && t < C;
        const bool ignored = t == ignore_index;
        if (!in_range && !(ce_only && ignored)) bad += 1.f;
        float lt = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float p = e[c] * inv;
            const bool hit = in_range && t == c;
            sP[c] += p;
            if (hit) { sI[c] += p; sT[c] += 1.f; lt = l[c]; }
        }
        if (!ignored && in_range) { nll += (mx + logf(s)) - lt; cnt += 1.f; }
    }
    // block reduction: wave shuffles then LDS across the 4 waves
    __shared__ float red[4][3 * C + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float a = wave_sum64(sI[c]), b = wave_sum64(sP[c]), d = wave_sum64(sT[c]);
        if (lane == 0) { red[wave][3 * c] = a; red[wave][3 * c + 1] = b; red[wave][3 * c + 2] = d; }
    }
    {
        const float a = wave_sum64(nll), b = wave_sum64(cnt), d = wave_sum64(bad);
        if (lane == 0) { red[wave][3 * C] = a; red[wave][3 * C + 1] = b; red[wave][3 * C + 2] = d; }
    }
    __syncthreads();
    if (threadIdx.x < 3 * C + 3) {
        const int k = threadIdx.x;
        const double s = (double)red[0][k] + (double)red[1][k] + (double)red[2][k] + (double)red[3][k];
        if (k < 3 * C) atomicAdd(&acc[(size_t)n * 3 * C + k], s);
        else atomicAdd(&acc[(size_t)N * 3 * C + (k - 3 * C)], s);
    }
}

// out[0] = loss, out[1] = CE, out[2] = Dice loss, out[3] = number of illegal labels
__global__ void dice_ce_finalize_kernel(const double* __restrict__ acc, int N, int C, float ce_w, float dice_w, double eps,
                                        float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double r = 0.0;
    for (int i = 0; i < N * C; ++i) {
        const double I = acc[i * 3], P = acc[i * 3 + 1], T = acc[i * 3 + 2];
        r += (2.0 * I + eps) / (P + T + eps);
    }
    const double dice = 1.0 - r / (double)(N * C);
    const double* tail = acc + (size_t)N * 3 * C;
    const double ce = tail[0] / tail[1];
    const float cef = (float)ce, dicef = (float)dice;
    float loss = 0.f;
    if (ce_w != 0.f) loss += ce_w * cef;
    if (dice_w != 0.f) loss += dice_w * dicef;
    out[0] = loss;
    out[1] = cef;
    out[2] = dicef;
    out[3] = (float)tail[2];
}

template <int C>
__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          long long HW, long long ignore_index, const double* __restrict__ acc,
                                                          int N, float ce_w, float dice_w, double eps,
                                                          const float* __restrict__ gout, float* __restrict__ dlogits) {
    const int n = blockIdx.y;
    const float* lg = logits + (size_t)n * C * HW;
    const long long* tg = target + (size_t)n * HW;
    float* dl = dlogits + (size_t)n * C * HW;
    const float g = gout != nullptr ? gout[0] : 1.f;
    // Dice: d/dp_c of -(1/(N*C)) * (2I+eps)/(U+eps), U = P+T:  -(1/(NC)) * (2[t==c]/(U+eps) - (2I+eps)/(U+eps)^2)
    float ka[C], kb[C];
    const double nc = (double)N * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const double I = acc[((size_t)n * C + c) * 3], U = acc[((size_t)n * C + c) * 3 + 1] + acc[((size_t)n * C + c) * 3 + 2];
        ka[c] = (float)(-(double)dice_w * 2.0 / ((U + eps) * nc));
        kb[c] = (float)((double)dice_w * (2.0 * I + eps) / ((U + eps) * (U + eps) * nc));
    }
    const double cntd = acc[(size_t)N * 3 * C + 1];
    const float kce = (float)((double)ce_w / cntd);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < HW; i += (long long)gridDim.x * 256) {
        float l[C], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) { l[c] = lg[c * HW + i]; mx = fmaxf(mx, l[c]); }
        float s = 0.f, p[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { p[c] = expf(l[c] - mx); s += p[c]; }
        const float inv = 1.f / s;
        const long long t = tg[i];
        const bool in_range = t >= 0 && t < C;
        const bool valid = in_range && t != ignore_index;
        float gp[C], dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            p[c] *= inv;
            gp[c] = kb[c] + ((in_range && t == c) ? ka[c] : 0.f);
            dot = fmaf(p[c], gp[c], dot);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float d = p[c] * (gp[c] - dot);
            if (valid) d += kce * (p[c] - ((t == c) ? 1.f : 0.f));
            dl[c * HW + i] = g * d;
        }
    }
}

// argmax over the class dim; first maximum wins, NaN counts as maximal (torch.argmax)
template <int C>
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, long long* __restrict__ out, long long HW, long long total) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const long long n = v / HW, i = v - n * HW;
        const float* lg = logits + (size_t)n * C * HW + i;
        float best = lg[0];
        int bi = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const float x = lg[c * HW];
            if (x > best || (x != x && best == best)) { best = x; bi = c; }
        }
        out[v] = bi;
    }
}

template <int C> __device__ __forceinline__ int argmax_arr(const float* p) {
    float best = p[0]; int bi = 0;
#pragma unroll
    for (int c = 1; c < C; ++c) if (p[c] > best || (p[c] != p[c] && best == best)) { best = p[c]; bi = c; }
    return bi;
}

// eval_ensemble.py:77-99 applied to every pixel; fp32 math in the reference's operation order
template <int C>
__global__ __launch_bounds__(256) void ensemble_fuse_kernel(const float* __restrict__ l1, const float* __restrict__ l2,
                                                            long long* __restrict__ out, long long HW, long long total, float thresh) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const long long n = v / HW, i = v - n * HW;
        const float* a = l1 + (size_t)n * C * HW + i;
        const float* b = l2 + (size_t)n * C * HW + i;
        float p1[C], p2[C], m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int c = 0; c < C; ++c) { p1[c] = a[c * HW]; p2[c] = b[c * HW]; m1 = fmaxf(m1, p1[c]); m2 = fmaxf(m2, p2[c]); }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { p1[c] = expf(p1[c] - m1); p2[c] = expf(p2[c] - m2); s1 += p1[c]; s2 += p2[c]; }
        float e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            p1[c] = p1[c] / s1; p2[c] = p2[c] / s2;
            e1 += p1[c] * logf(fmaxf(p1[c], 1e-8f));
            e2 += p2[c] * logf(fmaxf(p2[c], 1e-8f));
        }
        e1 = -e1; e2 = -e2;
        const float diff = fabsf(e1 - e2);
        const float w1 = 1.f / (e1 + 1e-8f), w2 = 1.f / (e2 + 1e-8f), ws = w1 + w2;
        float pe[C];
#pragma unroll
        for (int c = 0; c < C; ++c) pe[c] = (w1 * p1[c] + w2 * p2[c]) / ws;
        int r = argmax_arr<C>(pe);
        if (diff > thresh && e1 < e2) r = argmax_arr<C>(p1);
        if (diff > thresh && e2 < e1) r = argmax_arr<C>(p2);
        out[v] = r;
    }
}

// counts[K][K] (row = prediction, col = target) and sum |pred - target|, uint64 atomics
__global__ __launch_bounds__(256) void seg_counts_kernel(const long long* __restrict__ pred, const long long* __restrict__ target,
                                                         long long total, int K, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int h[65];
    for (int i = threadIdx.x; i < 65; i += 256) h[i] = 0;
    __shared__ unsigned long long mae;
    if (threadIdx.x == 0) mae = 0;
    __syncthreads();
    unsigned long long my = 0;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const long long p = pred[v], t = target[v];
        if (p >= 0 && p < K && t >= 0 && t < K) atomicAdd(&h[(int)p * K + (int)t], 1u);
        else atomicAdd(&h[64], 1u);
        my += (unsigned long long)(p > t ? p - t : t - p);
    }
    atomicAdd(&mae, my);
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += 256) if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);
    if (threadIdx.x == 0) { atomicAdd(&counts[64], mae); if (h[64]) atomicAdd(&counts[65], (unsigned long long)h[64]); }
}

// ------------------------------------------------------------------------------------------------
static inline int px_grid(long long n) {
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}
#define CLASS_SWITCH(K, STMT)                                                                  \
    switch (K) {                                                                               \
        case 1: { constexpr int CC = 1; STMT; } break; case 2: { constexpr int CC = 2; STMT; } break; \
        case 3: { constexpr int CC = 3; STMT; } break; case 4: { constexpr int CC = 4; STMT; } break; \
        case 5: { constexpr int CC = 5; STMT; } break; case 6: { constexpr int CC = 6; STMT; } break; \
        case 7: { constexpr int CC = 7; STMT; } break; case 8: { constexpr int CC = 8; STMT; } break; \
        default: unet_set_error("n_classes must be 1..8 (got %d)", K); return -1;              \
    }

// acc_ws: (N*C*3 + 3) doubles; out4: {loss, ce, dice, bad_label_count}
extern "C" int unet_dice_ce_fwd(const float* logits, const long long* target, int N, int C, long long HW,
                                float ce_weight, float dice_weight, long long ignore_index, int ce_only, double eps,
                                double* acc_ws, float* out4, void* stream) {
    UNET_CHECK(N > 0 && HW > 0, "dice_ce: empty input");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * ((size_t)N * C * 3 + 3), s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    int gx = px_grid(HW); if (gx * N > 4096) { gx = 4096 / N; if (gx < 1) gx = 1; }
    CLASS_SWITCH(C, hipLaunchKernelGGL((dice_ce_fwd_kernel<CC>), dim3(gx, N), dim3(256), 0, s, logits, target, HW, ignore_index, ce_only, acc_ws, N));
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(64), 0, s, acc_ws, N, C, ce_weight, ce_only ? 0.f : dice_weight, eps, out4);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_dice_ce_bwd(const float* logits, const long long* target, int N, int C, long long HW,
                                float ce_weight, float dice_weight, long long ignore_index, int ce_only, double eps,
                                const double* acc_ws, const float* grad_out /* device scalar or null */, float* dlogits, void* stream) {
    UNET_CHECK(N > 0 && HW > 0, "dice_ce: empty input");
    hipStream_t s = (hipStream_t)stream;
    int gx = px_grid(HW); if (gx * N > 4096) { gx = 4096 / N; if (gx < 1) gx = 1; }
    const float dwt = ce_only ? 0.f : dice_weight;
    CLASS_SWITCH(C, hipLaunchKernelGGL((dice_ce_bwd_kernel<CC>), dim3(gx, N), dim3(256), 0, s, logits, target, HW, ignore_index, acc_ws, N, ce_weight, dwt, eps, grad_out, dlogits));
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_argmax(const float* logits, long long* out, int N, int C, long long HW, void* stream) {
    const long long total = (long long)N * HW;
    CLASS_SWITCH(C, hipLaunchKernelGGL((argmax_kernel<CC>), dim3(px_grid(total) * 2), dim3(256), 0, (hipStream_t)stream, logits, out, HW, total));
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_ensemble_fuse(const float* logits1, const float* logits2, long long* out, int N, int C, long long HW,
                                  float entropy_thresh, void* stream) {
    const long long total = (long long)N * HW;
    CLASS_SWITCH(C, hipLaunchKernelGGL((ensemble_fuse_kernel<CC>), dim3(px_grid(total) * 2), dim3(256), 0, (hipStream_t)stream, logits1, logits2, out, HW, total, entropy_thresh));
    UNET_LAUNCH_CHECK();
    return 0;
}

// counts: 66 uint64: [K*K confusion (pred-major)] ... [64] = sum|p-t|, [65] = labels outside [0,K)
extern "C" int unet_seg_counts(const long long* pred, const long long* target, long long total, int K,
                               unsigned long long* counts, void* stream) {
    UNET_CHECK(K >= 1 && K <= 8, "seg_counts: n_classes must be 1..8");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, sizeof(unsigned long long) * 66, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    hipLaunchKernelGGL(seg_counts_kernel, dim3(px_grid(total)), dim3(256), 0, s, pred, target, total, K, counts);
    UNET_LAUNCH_CHECK();
    return 0;
}
