// Optimizer tail of step_batch (/root/reference/utils/train_step.py:56-63: unscale -> clip_grad_norm_
// -> AdamW step; the optimizer is torch.optim.AdamW from utils/builders.py:143-146) as multi-tensor
// kernels: one pass over the gradients for the global L2 norm, one pass that clips and updates.
// HBM-bound: 4 B/parameter for the norm, 16 B read + 12 B written per parameter for the update.
//
// Also the multi-tensor form of unet_pack_weights (one launch refreshes every packed image of a model).
//
// The tensor list is a HOST array at the C ABI; it travels to the device in the kernel-argument segment
// (up to MT_MAX tensors per launch), so nothing is allocated and no copy is enqueued.
#include "common.h"

constexpr int MT_MAX = 40;          // tensors per launch
constexpr int MT_CHUNK = 16384;     // elements per workgroup: 256 threads x 16 float4

struct MTNormTable {
    int n;
    int blk_end[MT_MAX];            // running end of each tensor's workgroup range inside this launch
    long long numel[MT_MAX];
    const float* g[MT_MAX];
};
struct MTAdamTable {
    int n;
    int blk_end[MT_MAX];
    long long numel[MT_MAX];
    float* p[MT_MAX];
    float* g[MT_MAX];
    float* m[MT_MAX];
    float* v[MT_MAX];
};
struct MTPackTable {
    int n;
    int blk_end[MT_MAX];
    const float* src[MT_MAX];
    void* dst[MT_MAX];
    int mode[MT_MAX], O[MT_MAX], I[MT_MAX], NT[MT_MAX];
};

__device__ __forceinline__ int mt_find(const int* blk_end, int n, int b) {
    int t = 0;
    while (t < n - 1 && b >= blk_end[t]) ++t;
    return t;
}

static inline int mt_chunks(long long numel) { return (int)((numel + MT_CHUNK - 1) / MT_CHUNK); }

__global__ __launch_bounds__(256) void mt_sumsq_kernel(MTNormTable tb, double* __restrict__ partials) {
    const int b = blockIdx.x;
    const int t = mt_find(tb.blk_end, tb.n, b);
    const int chunk = b - (t ? tb.blk_end[t - 1] : 0);
    const float* g = tb.g[t];
    const long long n = tb.numel[t];
    const long long e0 = (long long)chunk * MT_CHUNK;
    const long long e1 = e0 + MT_CHUNK < n ? e0 + MT_CHUNK : n;
    float s = 0.f;
    if ((((size_t)g) & 15) == 0) {
        const long long v1 = e0 + ((e1 - e0) & ~3LL);
        for (long long e = e0 + threadIdx.x * 4; e < v1; e += 1024) {
            const float4 x = *(const float4*)(g + e);
            s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
        for (long long e = v1 + threadIdx.x; e < e1; e += 256) s += g[e] * g[e];
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) s += g[e] * g[e];
    }
    __shared__ double red[4];
    const double w = wave_sum64d((double)s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) partials[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup, fixed summation order: bitwise reproducible total.  The tensors hold grad_scale^-1 times the gradients
// that are to be clipped (data parallel: the all-reduced SUM over W ranks, grad_scale = 1/W; otherwise 1):
// out[0] = ||grad_scale * g||_2, out[1] = grad_scale * min(1, max_norm / (norm + 1e-6)) -- clip_grad_norm_'s coefficient
// with the averaging folded in, so the update kernel's one multiply does both.
__global__ __launch_bounds__(256) void mt_norm_finalize_kernel(const double* __restrict__ partials, int n, float max_norm,
                                                               float grad_scale, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    __shared__ double red[4];
    const double w = wave_sum64d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)(sqrt((red[0] + red[1]) + (red[2] + red[3])) * (double)grad_scale);
        const float coef = max_norm / (norm + 1e-6f);
        out[0] = norm;
        out[1] = (coef < 1.0f ? coef : 1.0f) * grad_scale;
    }
}

struct AdamScalars { float decay, omb1, b2, omb2, bc2_sqrt, eps, step_size; };

__device__ __forceinline__ void adamw_elem(float& p, float& g, float& m, float& v, float coef, const AdamScalars& k) {
    g *= coef;
    p *= k.decay;
    m = m + (g - m) * k.omb1;
    v = v * k.b2 + (k.omb2 * g) * g;
    const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
    p = p - k.step_size * (m / denom);
}

// veto (may be NULL): a device float; > 0 means "this batch must not change anything" (step_batch hands the loss kernel's
// illegal-label counter over, so a batch the reference would have raised on before backward leaves weights, moments and
// gradients as they were, without a host sync before the optimizer)
__global__ __launch_bounds__(256) void mt_adamw_kernel(MTAdamTable tb, const float* __restrict__ clip, int write_g, AdamScalars k,
                                                       const float* __restrict__ veto) {
    if (veto != nullptr && veto[0] > 0.f) return;
    const int b = blockIdx.x;
    const int t = mt_find(tb.blk_end, tb.n, b);
    const int chunk = b - (t ? tb.blk_end[t - 1] : 0);
    float* p = tb.p[t]; float* g = tb.g[t]; float* m = tb.m[t]; float* v = tb.v[t];
    const long long n = tb.numel[t];
    const long long e0 = (long long)chunk * MT_CHUNK;
    const long long e1 = e0 + MT_CHUNK < n ? e0 + MT_CHUNK : n;
    const float coef = clip != nullptr ? clip[1] : 1.0f;
    const bool wg = write_g && coef != 1.0f;
    const bool vec = ((((size_t)p) | ((size_t)g) | ((size_t)m) | ((size_t)v)) & 15) == 0;
    long long s0 = e0;
    if (vec) {
        const long long v1 = e0 + ((e1 - e0) & ~3LL);
        for (long long e = e0 + threadIdx.x * 4; e < v1; e += 1024) {
            float4 P = *(float4*)(p + e), G = *(float4*)(g + e), M = *(float4*)(m + e), V = *(float4*)(v + e);
            adamw_elem(P.x, G.x, M.x, V.x, coef, k);
            adamw_elem(P.y, G.y, M.y, V.y, coef, k);
            adamw_elem(P.z, G.z, M.z, V.z, coef, k);
            adamw_elem(P.w, G.w, M.w, V.w, coef, k);
            *(float4*)(p + e) = P; *(float4*)(m + e) = M; *(float4*)(v + e) = V;
            if (wg) *(float4*)(g + e) = G;
        }
        s0 = v1;
    }
    for (long long e = s0 + threadIdx.x; e < e1; e += 256) {
        float P = p[e], G = g[e], M = m[e], V = v[e];
        adamw_elem(P, G, M, V, coef, k);
        p[e] = P; m[e] = M; v[e] = V;
        if (wg) g[e] = G;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mt_pack_kernel(MTPackTable tb) {
    const int b = blockIdx.x;
    const int t = mt_find(tb.blk_end, tb.n, b);
    const int chunk = b - (t ? tb.blk_end[t - 1] : 0);
    const float* __restrict__ src = tb.src[t];
    T* __restrict__ dst = (T*)tb.dst[t];
    const int mode = tb.mode[t], O = tb.O[t], I = tb.I[t], NT = tb.NT[t];
    const size_t total = (size_t)O * I * NT;
    const size_t d0 = (size_t)chunk * MT_CHUNK;
    const size_t d1 = d0 + MT_CHUNK < total ? d0 + MT_CHUNK : total;
    for (size_t d = d0 + threadIdx.x; d < d1; d += 256) {       // same index maps as pack_weights_kernel
        size_t s;
        if (mode == 0) {
            const int i = (int)(d % I); const size_t r = d / I; const int o = (int)(r % O); const int tap = (int)(r / O);
            s = ((size_t)o * I + i) * NT + tap;
        } else if (mode == 1) {
            const int o = (int)(d % O); const size_t r = d / O; const int i = (int)(r % I); const int tf = (int)(r / I);
            s = ((size_t)o * I + i) * NT + (NT - 1 - tf);
        } else if (mode == 2) {
            const int ci = (int)(d % O); const size_t r = d / O; const int c = (int)(r % I); const int q = (int)(r / I);
            s = ((size_t)ci * I + c) * 4 + q;
        } else {
            const int Co = I, Ci = O;
            const int cc = (int)(d % (2 * Co)); const size_t r = d / (2 * Co); const int ci = (int)(r % Ci); const int i2 = (int)(r / Ci);
            const int j2 = cc / Co, c = cc - j2 * Co;
            s = ((size_t)ci * Co + c) * 4 + i2 * 2 + j2;
        }
        dst[d] = from_f32<T>(src[s]);
    }
}

// Conv weights (modes 0 and 1 share one fp32 source): a workgroup stages a 32(o) x TI(i) x NT tile through LDS, reading
// each source row segment contiguously ONCE, and writes the forward image [tap][O][I] and/or the dgrad image
// [NT-1-tap][I][O] from it in 64-byte runs.  The per-element gather above touches one cache line per element for the
// dgrad image (stride I*NT floats between neighbouring lanes).
constexpr int PT_O = 32, PT_ROW = 288;            // TI*NT <= PT_ROW: TI = 32 for NT <= 9, 8 for NT <= 36
struct MTPackTileTable {
    int n;
    int blk_end[MT_MAX];
    const float* src[MT_MAX];
    void* dst0[MT_MAX];             // forward image or null
    void* dst1[MT_MAX];             // dgrad image or null
    int O[MT_MAX], I[MT_MAX], NT[MT_MAX];
};
static inline int pack_ti(int NT) { return NT <= 9 ? 32 : 8; }
static inline int pack_tiles(int O, int I, int NT) { return ((O + PT_O - 1) / PT_O) * ((I + pack_ti(NT) - 1) / pack_ti(NT)); }

template <typename T>
__global__ __launch_bounds__(256) void mt_pack_tile_kernel(MTPackTileTable tb) {
    __shared__ float tile[PT_O][PT_ROW + 1];
    const int b = blockIdx.x;
    const int t = mt_find(tb.blk_end, tb.n, b);
    const int tl = b - (t ? tb.blk_end[t - 1] : 0);
    const float* __restrict__ src = tb.src[t];
    T* __restrict__ d0 = (T*)tb.dst0[t];
    T* __restrict__ d1 = (T*)tb.dst1[t];
    const int O = tb.O[t], I = tb.I[t], NT = tb.NT[t];
    const int TI = NT <= 9 ? 32 : 8;
    const int tiles_i = (I + TI - 1) / TI;
    const int o0 = (tl / tiles_i) * PT_O, i0 = (tl % tiles_i) * TI;
    const int no = O - o0 < PT_O ? O - o0 : PT_O, ni = I - i0 < TI ? I - i0 : TI;
    const int span = ni * NT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = wave; r < no; r += 4) {
        const float* row = src + ((size_t)(o0 + r) * I + i0) * NT;
        for (int e = lane; e < span; e += 64) tile[r][e] = row[e];
    }
    __syncthreads();
    const int per_tap = PT_O * TI;
    if (d0) {
        for (int idx = threadIdx.x; idx < per_tap * NT; idx += 256) {
            const int tap = idx / per_tap, rem = idx - tap * per_tap;
            const int r = rem / TI, i = rem - r * TI;
            if (r < no && i < ni) d0[((size_t)tap * O + o0 + r) * I + i0 + i] = from_f32<T>(tile[r][i * NT + tap]);
        }
    }
    if (d1) {
        for (int idx = threadIdx.x; idx < per_tap * NT; idx += 256) {
            const int tap = idx / per_tap, rem = idx - tap * per_tap;
            const int i = rem / PT_O, r = rem - i * PT_O;
            if (r < no && i < ni) d1[((size_t)(NT - 1 - tap) * I + i0 + i) * O + o0 + r] = from_f32<T>(tile[r][i * NT + tap]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------

extern "C" int unet_mt_blocks(int nt, const long long* numel) {
    long long nb = 0;
    for (int i = 0; i < nt; ++i) nb += numel[i] > 0 ? mt_chunks(numel[i]) : 0;
    return nb > 0x7fffffffLL ? -1 : (int)nb;
}

extern "C" int unet_grad_sumsq(int nt, const void* const* grads, const long long* numel, double* partials, void* stream) {
    UNET_CHECK(nt >= 0 && (nt == 0 || (grads && numel && partials)), "grad_sumsq: null argument");
    hipStream_t s = (hipStream_t)stream;
    int done = 0;
    for (int i0 = 0; i0 < nt;) {
        MTNormTable tb; tb.n = 0; int nb = 0;
        for (; i0 < nt && tb.n < MT_MAX; ++i0) {
            if (numel[i0] <= 0) continue;
            UNET_CHECK(grads[i0] != nullptr, "grad_sumsq: tensor %d is null", i0);
            nb += mt_chunks(numel[i0]);
            tb.blk_end[tb.n] = nb; tb.numel[tb.n] = numel[i0]; tb.g[tb.n] = (const float*)grads[i0]; ++tb.n;
        }
        if (tb.n == 0) continue;
        hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nb), dim3(256), 0, s, tb, partials + done);
        UNET_LAUNCH_CHECK();
        done += nb;
    }
    return 0;
}

extern "C" int unet_grad_norm_finalize(const double* partials, int nblocks, float max_norm, float grad_scale, float* norm_out, void* stream) {
    UNET_CHECK(nblocks >= 0 && norm_out && (nblocks == 0 || partials), "grad_norm_finalize: null argument");
    UNET_CHECK(grad_scale > 0.f, "grad_norm_finalize: grad_scale must be positive");
    hipLaunchKernelGGL(mt_norm_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nblocks, max_norm, grad_scale, norm_out);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_adamw_step(int nt, void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                               const long long* numel, const float* norm_out, int write_grads, double lr, double beta1,
                               double beta2, double eps, double weight_decay, long long step, const float* veto, void* stream) {
    UNET_CHECK(nt >= 0 && (nt == 0 || (params && grads && exp_avg && exp_avg_sq && numel)), "adamw_step: null argument");
    UNET_CHECK(step >= 1, "adamw_step: step must be >= 1 (got %lld)", step);
    UNET_CHECK(beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && lr >= 0, "adamw_step: bad hyper-parameters");
    // scalar prologue of torch.optim.adamw._single_tensor_adamw, evaluated in double as Python does
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamScalars k;
    k.decay = (float)(1.0 - lr * weight_decay);
    k.omb1 = (float)(1.0 - beta1);
    k.b2 = (float)beta2;
    k.omb2 = (float)(1.0 - beta2);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.eps = (float)eps;
    k.step_size = (float)(lr / bc1);
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < nt;) {
        MTAdamTable tb; tb.n = 0; int nb = 0;
        for (; i0 < nt && tb.n < MT_MAX; ++i0) {
            if (numel[i0] <= 0) continue;
            UNET_CHECK(params[i0] && grads[i0] && exp_avg[i0] && exp_avg_sq[i0], "adamw_step: tensor %d has a null pointer", i0);
            nb += mt_chunks(numel[i0]);
            tb.blk_end[tb.n] = nb; tb.numel[tb.n] = numel[i0];
            tb.p[tb.n] = (float*)params[i0]; tb.g[tb.n] = (float*)grads[i0];
            tb.m[tb.n] = (float*)exp_avg[i0]; tb.v[tb.n] = (float*)exp_avg_sq[i0]; ++tb.n;
        }
        if (tb.n == 0) continue;
        hipLaunchKernelGGL(mt_adamw_kernel, dim3(nb), dim3(256), 0, s, tb, norm_out, write_grads, k, veto);
        UNET_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int unet_pack_weights_multi(int dtype, int nt, const int* modes, const void* const* src, void* const* dst,
                                       const int* O, const int* I, const int* ntaps, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    UNET_CHECK(nt >= 0 && (nt == 0 || (modes && src && dst && O && I && ntaps)), "pack_weights_multi: null argument");
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < nt; ++i) {
        UNET_CHECK(modes[i] >= 0 && modes[i] <= 3, "bad pack mode %d", modes[i]);
        if (modes[i] >= 2) UNET_CHECK(ntaps[i] == 4, "convT pack needs 4 taps");
        UNET_CHECK(src[i] && dst[i] && O[i] > 0 && I[i] > 0 && ntaps[i] > 0, "pack_weights_multi: bad tensor %d", i);
        UNET_CHECK((long long)O[i] * I[i] * ntaps[i] < (1LL << 31), "pack_weights_multi: tensor %d too large", i);
    }
    auto tiled = [&](int i) { return modes[i] <= 1 && ntaps[i] <= 36; };
    // conv weights: tiled kernel, the forward and dgrad images of one source share a table entry
    {
        MTPackTileTable tb; tb.n = 0; int nb = 0;
        auto flush = [&]() -> int {
            if (tb.n == 0) return 0;
            if (dtype == UNET_BF16) hipLaunchKernelGGL((mt_pack_tile_kernel<__bf16>), dim3(nb), dim3(256), 0, s, tb);
            else hipLaunchKernelGGL((mt_pack_tile_kernel<float>), dim3(nb), dim3(256), 0, s, tb);
            UNET_LAUNCH_CHECK();
            tb.n = 0; nb = 0;
            return 0;
        };
        for (int i = 0; i < nt; ++i) {
            if (!tiled(i)) continue;
            int e = -1;
            for (int j = 0; j < tb.n; ++j)
                if (tb.src[j] == (const float*)src[i] && tb.O[j] == O[i] && tb.I[j] == I[i] && tb.NT[j] == ntaps[i] &&
                    (modes[i] == 0 ? tb.dst0[j] : tb.dst1[j]) == nullptr) { e = j; break; }
            if (e < 0) {
                if (tb.n == MT_MAX) { int rc = flush(); if (rc) return rc; }
                e = tb.n++;
                nb += pack_tiles(O[i], I[i], ntaps[i]);
                tb.blk_end[e] = nb; tb.src[e] = (const float*)src[i]; tb.dst0[e] = nullptr; tb.dst1[e] = nullptr;
                tb.O[e] = O[i]; tb.I[e] = I[i]; tb.NT[e] = ntaps[i];
            }
            (modes[i] == 0 ? tb.dst0[e] : tb.dst1[e]) = dst[i];
        }
        int rc = flush(); if (rc) return rc;
    }
    // transposed-conv images (and anything the tile does not hold): per-element gather
    for (int i0 = 0; i0 < nt;) {
        MTPackTable tb; tb.n = 0; int nb = 0;
        for (; i0 < nt && tb.n < MT_MAX; ++i0) {
            if (tiled(i0)) continue;
            nb += mt_chunks((long long)O[i0] * I[i0] * ntaps[i0]);
            tb.blk_end[tb.n] = nb; tb.src[tb.n] = (const float*)src[i0]; tb.dst[tb.n] = dst[i0];
            tb.mode[tb.n] = modes[i0]; tb.O[tb.n] = O[i0]; tb.I[tb.n] = I[i0]; tb.NT[tb.n] = ntaps[i0]; ++tb.n;
        }
        if (tb.n == 0) continue;
        if (dtype == UNET_BF16) hipLaunchKernelGGL((mt_pack_kernel<__bf16>), dim3(nb), dim3(256), 0, s, tb);
        else hipLaunchKernelGGL((mt_pack_kernel<float>), dim3(nb), dim3(256), 0, s, tb);
        UNET_LAUNCH_CHECK();
    }
    return 0;
}
