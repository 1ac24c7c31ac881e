// RepUNet training-path kernels (gfx950, bandwidth-bound): GroupNorm statistics from the conv
// epilogue rows, the multi-branch "sum of normalised convs + ReLU", and GroupNorm backward.
//
// Reference semantics: getBN -> nn.GroupNorm(min(8,C), C, eps=1e-5) (/root/reference/models/reparm_unet.py:17-25),
// AsymmConvBlock.forward = GN(conv_kx1) + GN(conv_1xk) + GN(conv_kxk) (:67-72),
// MultiConvBlock.forward = sum over kernel sizes (:114-122), then ReLU (ReparamDoubleConv :162-167).
//
// GroupNorm is per (sample, group): kernels here run on a grid (gx, N) so a workgroup stays inside
// one image and its per-(n,c) coefficients live in registers.
#include "common.h"

#define MAX_BRANCH 8

// ---------------------------------------------------------------------------------------------
// rows [N*rpi][C][2] (per-tile sum / sumsq, tiles never cross images) -> per-(n,c) scale/shift and
// per-(n,g) mean/invstd.  One workgroup per (image, group): its C/groups channels x 256/(C/groups) row lanes walk the
// image's rows with coalesced 8-byte loads, the partials are added in lane order (deterministic), then the group's
// statistics and its channels' coefficients are written.  (The first version used one workgroup per IMAGE with a serial
// row loop per thread: 95 us per call at 1024 rows, 10 % of a RepUNet training step.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ rows, int rpi, int C, int groups, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    __shared__ double red[256][2];
    __shared__ double gstat[2];
    const int g = blockIdx.x, n = blockIdx.y;
    const int cg = C / groups;                          // channels of this group (<= 256, checked by the launcher)
    const int RL = 256 / cg;                            // row lanes
    const int ch = threadIdx.x % cg, rl = threadIdx.x / cg;
    double s1 = 0.0, s2 = 0.0;
    if (rl < RL) {
        const float* col = rows + ((size_t)n * rpi * C + (size_t)g * cg + ch) * 2;
        int r = rl;
        for (; r + 3 * RL < rpi; r += 4 * RL) {         // four independent loads in flight
            const float2 v0 = *(const float2*)(col + (size_t)r * C * 2);
            const float2 v1 = *(const float2*)(col + (size_t)(r + RL) * C * 2);
            const float2 v2 = *(const float2*)(col + (size_t)(r + 2 * RL) * C * 2);
            const float2 v3 = *(const float2*)(col + (size_t)(r + 3 * RL) * C * 2);
            s1 += (double)v0.x; s2 += (double)v0.y; s1 += (double)v1.x; s2 += (double)v1.y;
            s1 += (double)v2.x; s2 += (double)v2.y; s1 += (double)v3.x; s2 += (double)v3.y;
        }
        for (; r < rpi; r += RL) {
            const float2 v = *(const float2*)(col + (size_t)r * C * 2);
            s1 += (double)v.x; s2 += (double)v.y;
        }
    }
    red[threadIdx.x][0] = s1; red[threadIdx.x][1] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < RL * cg; ++i) { a += red[i][0]; b += red[i][1]; }
        const double m = a / count;
        double var = b / count - m * m;
        if (var < 0.0) var = 0.0;
        gstat[0] = m;
        gstat[1] = 1.0 / sqrt(var + (double)eps);
        mean_out[n * groups + g] = (float)m;
        invstd_out[n * groups + g] = (float)gstat[1];
    }
    __syncthreads();
    if (threadIdx.x < cg) {
        const int c = g * cg + threadIdx.x;
        const float sc = gamma[c] * (float)gstat[1];
        scale[(size_t)n * C + c] = sc;
        shift[(size_t)n * C + c] = beta[c] - (float)gstat[0] * sc;
    }
}

struct BranchArgs {
    const void* z[MAX_BRANCH];
    const float* scale[MAX_BRANCH];
    const float* shift[MAX_BRANCH];
    int nb;
    int coef_nstride;      // C for per-sample coefficients (GroupNorm), 0 for per-channel (BatchNorm)
};

// y[n,p,c] = act( sum_b z_b[n,p,c]*scale_b[n][c] + shift_b[n][c] ); branches dense [N,HW,C]
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void branch_sum_act_kernel(BranchArgs a, T* __restrict__ y, int ldy, int yoff, long long HW, int C) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    if (pl >= PL) return;
    const int n = blockIdx.y;
    float sc[MAX_BRANCH][EPV], sh[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) sh[e] = 0.f;
#pragma unroll
    for (int b = 0; b < MAX_BRANCH; ++b) {
        if (b < a.nb) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                sc[b][e] = a.scale[b][(size_t)n * a.coef_nstride + cg * EPV + e];
                sh[e] += a.shift[b][(size_t)n * a.coef_nstride + cg * EPV + e];
            }
        }
    }
    for (long long p = (long long)blockIdx.x * PL + pl; p < HW; p += (long long)gridDim.x * PL) {
        const long long pix = (long long)n * HW + p;
        float o[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = sh[e];
#pragma unroll
        for (int b = 0; b < MAX_BRANCH; ++b) {
            if (b < a.nb) {
                float f[EPV];
                Vec16<T>::unpack(*(const uint4*)((const T*)a.z[b] + pix * C + cg * EPV), f);
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = fmaf(f[e], sc[b][e], o[e]);
            }
        }
        if (RELU) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        *(uint4*)(y + pix * ldy + yoff + cg * EPV) = Vec16<T>::pack(o);
    }
}

// per-(n,c): S1 = sum_p u, S2 = sum_p u*v   (u = masked upstream gradient, v = branch conv output)
template <typename T>
__global__ __launch_bounds__(256) void pair_reduce_kernel(const T* __restrict__ u, const T* __restrict__ v, long long HW, int C,
                                                          double* __restrict__ acc /* [N][C][2] */) {
    constexpr int EPV = TT<T>::EPV;
    extern __shared__ float sm_f[];
    const int G = C / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    const int n = blockIdx.y;
    float p1[EPV], p2[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) p1[e] = p2[e] = 0.f;
    if (active) {
        for (long long p = (long long)blockIdx.x * PL + pl; p < HW; p += (long long)gridDim.x * PL) {
            const long long pix = (long long)n * HW + p;
            float fu[EPV], fv[EPV];
            Vec16<T>::unpack(*(const uint4*)(u + pix * C + cg * EPV), fu);
            Vec16<T>::unpack(*(const uint4*)(v + pix * C + cg * EPV), fv);
#pragma unroll
            for (int e = 0; e < EPV; ++e) { p1[e] += fu[e]; p2[e] = fmaf(fu[e], fv[e], p2[e]); }
        }
        // LDS [pl][cg][2*EPV]
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            sm_f[(pl * G + cg) * 2 * EPV + e] = p1[e];
            sm_f[(pl * G + cg) * 2 * EPV + EPV + e] = p2[e];
        }
    }
    __syncthreads();
    const int per = G * 2 * EPV;
    for (int i = threadIdx.x; i < per; i += 256) {
        float s = 0.f;
        for (int p = 0; p < PL; ++p) s += sm_f[p * per + i];
        const int cgi = i / (2 * EPV), r = i - cgi * 2 * EPV, k = r / EPV, e = r - k * EPV;
        atomicAdd(&acc[((size_t)n * C + cgi * EPV + e) * 2 + k], (double)s);
    }
}

// GroupNorm backward coefficients: dz = A*u + B*z + D per (n,c); dgamma/dbeta accumulated over n.
// One workgroup per image.  m = HW * C/groups elements per group.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ acc, int C, int groups, double m,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cD,
                                                              double* __restrict__ dgb /* [C][2] */) {
    extern __shared__ double sm_d[];          // [groups][2]
    const int n = blockIdx.x;
    const int cg = C / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        const double mu = mean[n * groups + g], is = invstd[n * groups + g];
        double t1 = 0.0, t2 = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const double s1 = acc[((size_t)n * C + c) * 2], s2 = acc[((size_t)n * C + c) * 2 + 1];
            t1 += (double)gamma[c] * s1;
            t2 += (double)gamma[c] * is * (s2 - mu * s1);
        }
        sm_d[g * 2] = t1 / m;          // c1 = mean(dxhat)
        sm_d[g * 2 + 1] = t2 / m;      // c2 = mean(dxhat * xhat)
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cg;
        const double mu = mean[n * groups + g], is = invstd[n * groups + g];
        const double c1 = sm_d[g * 2], c2 = sm_d[g * 2 + 1];
        cA[(size_t)n * C + c] = (float)(is * (double)gamma[c]);
        cB[(size_t)n * C + c] = (float)(-is * is * c2);
        cD[(size_t)n * C + c] = (float)(-is * c1 + is * is * mu * c2);
        const double s1 = acc[((size_t)n * C + c) * 2], s2 = acc[((size_t)n * C + c) * 2 + 1];
        atomicAdd(&dgb[c * 2], is * (s2 - mu * s1));
        atomicAdd(&dgb[c * 2 + 1], s1);
    }
}

// out = A[n][c]*u + B[n][c]*v + D[n][c]
template <typename T>
__global__ __launch_bounds__(256) void affine2_kernel(const T* __restrict__ u, const T* __restrict__ v, T* __restrict__ out,
                                                      const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cD,
                                                      long long HW, int C) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    if (pl >= PL) return;
    const int n = blockIdx.y;
    float A[EPV], B[EPV], D[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        A[e] = cA[(size_t)n * C + cg * EPV + e]; B[e] = cB[(size_t)n * C + cg * EPV + e]; D[e] = cD[(size_t)n * C + cg * EPV + e];
    }
    for (long long p = (long long)blockIdx.x * PL + pl; p < HW; p += (long long)gridDim.x * PL) {
        const long long pix = (long long)n * HW + p;
        float fu[EPV], fv[EPV];
        Vec16<T>::unpack(*(const uint4*)(u + pix * C + cg * EPV), fu);
        Vec16<T>::unpack(*(const uint4*)(v + pix * C + cg * EPV), fv);
#pragma unroll
        for (int e = 0; e < EPV; ++e) fu[e] = fmaf(A[e], fu[e], fmaf(B[e], fv[e], D[e]));
        *(uint4*)(out + pix * C + cg * EPV) = Vec16<T>::pack(fu);
    }
}

__global__ void dgb_store_kernel(const double* __restrict__ dgb, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { dgamma[c] = (float)dgb[c * 2]; dbeta[c] = (float)dgb[c * 2 + 1]; }
}

// ---- GroupNorm backward of ALL branches of a block in one pass each ---------------------------------------------
// The branches of a MultiConvBlock (models/reparm_unet.py:114-122) share their upstream gradient u (the block's ReLU-masked
// dy): per-branch launches read u six times in the reduction and six times in the apply pass.  Here u is read once per
// pass: reduce 12 -> 7 tensor reads, apply 18 -> 13 accesses for kernel_sizes [3,5].
struct GnMultiArgs {
    const void* z[MAX_BRANCH];
    void* dz[MAX_BRANCH];
    const float* gamma[MAX_BRANCH];
    const float* mean[MAX_BRANCH];
    const float* invstd[MAX_BRANCH];
    float* dgamma[MAX_BRANCH];
    float* dbeta[MAX_BRANCH];
    int nb;
};

// acc + b*acc_stride: [N][C][2] = (sum_p u, sum_p u*z_b)
template <typename T>
__global__ __launch_bounds__(256) void multi_pair_reduce_kernel(GnMultiArgs a, const T* __restrict__ u, long long HW, int C,
                                                                double* __restrict__ acc, size_t acc_stride) {
    constexpr int EPV = TT<T>::EPV;
    extern __shared__ float sm_f[];                 // [pl][cg][EPV], reused per reduced quantity
    const int G = C / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    const int n = blockIdx.y;
    float p1[EPV], p2[MAX_BRANCH][EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) p1[e] = 0.f;
#pragma unroll
    for (int b = 0; b < MAX_BRANCH; ++b)
#pragma unroll
        for (int e = 0; e < EPV; ++e) p2[b][e] = 0.f;
    if (active) {
        for (long long p = (long long)blockIdx.x * PL + pl; p < HW; p += (long long)gridDim.x * PL) {
            const long long off = ((long long)n * HW + p) * C + cg * EPV;
            uint4 rz[MAX_BRANCH];
            const uint4 ru = *(const uint4*)(u + off);
#pragma unroll
            for (int b = 0; b < MAX_BRANCH; ++b)
                if (b < a.nb) rz[b] = *(const uint4*)((const T*)a.z[b] + off);
            float fu[EPV];
            Vec16<T>::unpack(ru, fu);
#pragma unroll
            for (int e = 0; e < EPV; ++e) p1[e] += fu[e];
#pragma unroll
            for (int b = 0; b < MAX_BRANCH; ++b) {
                if (b < a.nb) {
                    float fv[EPV];
                    Vec16<T>::unpack(rz[b], fv);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) p2[b][e] = fmaf(fu[e], fv[e], p2[b][e]);
                }
            }
        }
    }
    const int per = G * EPV;
    // quantity q = 0: sum u (written to every branch's slot 0); q = 1 + b: sum u*z_b
#pragma unroll
    for (int q = 0; q <= MAX_BRANCH; ++q) {
        if (q > a.nb) break;
        __syncthreads();
        if (active) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) sm_f[(pl * G + cg) * EPV + e] = q == 0 ? p1[e] : p2[q > 0 ? q - 1 : 0][e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += 256) {
            float s = 0.f;
            for (int pp = 0; pp < PL; ++pp) s += sm_f[pp * per + i];
            if (q == 0) {
                for (int b = 0; b < a.nb; ++b) atomicAdd(&acc[b * acc_stride + ((size_t)n * C + i) * 2], (double)s);
            } else {
                atomicAdd(&acc[(q - 1) * acc_stride + ((size_t)n * C + i) * 2 + 1], (double)s);
            }
        }
    }
}

// blockIdx.y = branch: the per-branch finalize of gn_bwd_finalize_kernel
__global__ __launch_bounds__(256) void gn_bwd_finalize_multi_kernel(GnMultiArgs a, const double* __restrict__ acc0, size_t acc_stride,
                                                                    int N, int C, int groups, double m,
                                                                    float* __restrict__ coef0, size_t coef_stride) {
    extern __shared__ double sm_d[];          // [groups][2]
    const int n = blockIdx.x, b = blockIdx.y;
    const double* acc = acc0 + b * acc_stride;
    double* dgb = (double*)acc + (size_t)N * C * 2;
    float* cA = coef0 + b * coef_stride; float* cB = cA + (size_t)N * C; float* cD = cB + (size_t)N * C;
    const float* gamma = a.gamma[b]; const float* mean = a.mean[b]; const float* invstd = a.invstd[b];
    const int cg = C / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        const double mu = mean[n * groups + g], is = invstd[n * groups + g];
        double t1 = 0.0, t2 = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const double s1 = acc[((size_t)n * C + c) * 2], s2 = acc[((size_t)n * C + c) * 2 + 1];
            t1 += (double)gamma[c] * s1;
            t2 += (double)gamma[c] * is * (s2 - mu * s1);
        }
        sm_d[g * 2] = t1 / m;
        sm_d[g * 2 + 1] = t2 / m;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cg;
        const double mu = mean[n * groups + g], is = invstd[n * groups + g];
        const double c1 = sm_d[g * 2], c2 = sm_d[g * 2 + 1];
        cA[(size_t)n * C + c] = (float)(is * (double)gamma[c]);
        cB[(size_t)n * C + c] = (float)(-is * is * c2);
        cD[(size_t)n * C + c] = (float)(-is * c1 + is * is * mu * c2);
        const double s1 = acc[((size_t)n * C + c) * 2], s2 = acc[((size_t)n * C + c) * 2 + 1];
        atomicAdd(&dgb[c * 2], is * (s2 - mu * s1));
        atomicAdd(&dgb[c * 2 + 1], s1);
    }
}

// dz_b = A_b[n][c]*u + B_b[n][c]*z_b + D_b[n][c] for every branch; a thread keeps MU pixels of u in registers across the branches
template <typename T>
__global__ __launch_bounds__(256) void multi_affine2_kernel(GnMultiArgs a, const T* __restrict__ u, const float* __restrict__ coef0,
                                                            size_t coef_stride, int N, long long HW, int C) {
    constexpr int EPV = TT<T>::EPV, MU = 4;
    const int G = C / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    if (pl >= PL) return;
    const int n = blockIdx.y;
    const long long stride = (long long)gridDim.x * PL;
    const size_t NC = (size_t)N * C, co = (size_t)n * C + cg * EPV;
    for (long long p0 = (long long)blockIdx.x * PL + pl; p0 < HW; p0 += stride * MU) {
        uint4 ru[MU];
#pragma unroll
        for (int i = 0; i < MU; ++i)
            if (p0 + i * stride < HW) ru[i] = *(const uint4*)(u + ((long long)n * HW + p0 + i * stride) * C + cg * EPV);
#pragma unroll
        for (int b = 0; b < MAX_BRANCH; ++b) {
            if (b >= a.nb) break;
            const float* cf = coef0 + b * coef_stride + co;
            float A[EPV], B[EPV], D[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) { A[e] = cf[e]; B[e] = cf[NC + e]; D[e] = cf[2 * NC + e]; }
            uint4 rz[MU];
#pragma unroll
            for (int i = 0; i < MU; ++i)
                if (p0 + i * stride < HW) rz[i] = *(const uint4*)((const T*)a.z[b] + ((long long)n * HW + p0 + i * stride) * C + cg * EPV);
#pragma unroll
            for (int i = 0; i < MU; ++i) {
                if (p0 + i * stride >= HW) break;
                float fu[EPV], fv[EPV];
                Vec16<T>::unpack(ru[i], fu);
                Vec16<T>::unpack(rz[i], fv);
#pragma unroll
                for (int e = 0; e < EPV; ++e) fu[e] = fmaf(A[e], fu[e], fmaf(B[e], fv[e], D[e]));
                *(uint4*)((T*)a.dz[b] + ((long long)n * HW + p0 + i * stride) * C + cg * EPV) = Vec16<T>::pack(fu);
            }
        }
    }
}

__global__ void dgb_store_multi_kernel(GnMultiArgs a, const double* __restrict__ acc0, size_t acc_stride, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const double* dgb = acc0 + b * acc_stride + (size_t)N * C * 2;
    if (c < C) { a.dgamma[b][c] = (float)dgb[c * 2]; a.dbeta[b][c] = (float)dgb[c * 2 + 1]; }
}

// ---------------------------------------------------------------------------------------------
static int img_grid(long long HW, int PL, int N) {
    long long g = (HW + PL - 1) / PL;
    int cap = 2048 / (N > 0 ? N : 1); if (cap < 1) cap = 1;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
static int ccheck(int dtype, int C, const char* what) {
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    if (dtype != UNET_F32 && dtype != UNET_BF16) { unet_set_error("%s: bad dtype %d", what, dtype); return -1; }
    if (C <= 0 || C % 8 != 0 || C / epv > 256) { unet_set_error("%s: C=%d must be a multiple of 8 and at most %d", what, C, 256 * epv); return -1; }
    return 0;
}

extern "C" int unet_gn_finalize(const float* stat_rows, int N, int rows_per_image, int C, int groups, long long HW,
                                const float* gamma, const float* beta, float eps,
                                float* scale /* [N][C] */, float* shift /* [N][C] */, float* mean /* [N][groups] */, float* invstd,
                                void* stream) {
    UNET_CHECK(N > 0 && rows_per_image > 0 && C > 0 && groups > 0 && C % groups == 0, "gn_finalize: bad shape");
    UNET_CHECK(C / groups <= 256, "gn_finalize: at most 256 channels per group (got %d)", C / groups);
    UNET_CHECK(stat_rows && gamma && beta && scale && shift && mean && invstd, "gn_finalize: null argument");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, N), dim3(256), 0, (hipStream_t)stream, stat_rows, rows_per_image, C, groups,
                       (double)HW * (C / groups), gamma, beta, eps, scale, shift, mean, invstd);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_branch_sum_act(int dtype, int relu, int nb, const void* const* z, const float* const* scale, const float* const* shift,
                                   int coef_nstride, void* y, int ldy, int yoff, int N, long long HW, int C, void* stream) {
    if (ccheck(dtype, C, "branch_sum_act")) return -1;
    UNET_CHECK(nb >= 1 && nb <= MAX_BRANCH, "branch_sum_act: 1..%d branches", MAX_BRANCH);
    UNET_CHECK(coef_nstride == 0 || coef_nstride == C, "branch_sum_act: coefficient stride must be 0 or C");
    UNET_CHECK(ldy % 8 == 0 && yoff % 8 == 0, "branch_sum_act: output stride/offset must be multiples of 8");
    BranchArgs a;
    for (int b = 0; b < MAX_BRANCH; ++b) {
        a.z[b] = b < nb ? z[b] : nullptr; a.scale[b] = b < nb ? scale[b] : nullptr; a.shift[b] = b < nb ? shift[b] : nullptr;
    }
    a.nb = nb; a.coef_nstride = coef_nstride;
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    dim3 grid(img_grid(HW, PL, N), N);
    hipStream_t s = (hipStream_t)stream;
#define LB(T, R) hipLaunchKernelGGL((branch_sum_act_kernel<T, R>), grid, dim3(256), 0, s, a, (T*)y, ldy, yoff, HW, C)
    if (dtype == UNET_BF16) { if (relu) LB(__bf16, true); else LB(__bf16, false); }
    else { if (relu) LB(float, true); else LB(float, false); }
#undef LB
    UNET_LAUNCH_CHECK();
    return 0;
}

// GroupNorm backward for one branch: u = upstream gradient already masked by the ReLU, z = the
// branch's conv output.  acc_ws: N*C*2 + C*2 doubles; coef_ws: 3*N*C floats.
extern "C" int unet_gn_bwd(int dtype, const void* u, const void* z, void* dz, int N, long long HW, int C, int groups,
                           const float* gamma, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, double* acc_ws, float* coef_ws, void* stream) {
    if (ccheck(dtype, C, "gn_bwd")) return -1;
    UNET_CHECK(groups > 0 && C % groups == 0, "gn_bwd: bad group count");
    hipStream_t s = (hipStream_t)stream;
    const size_t nacc = (size_t)N * C * 2 + (size_t)C * 2;
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * nacc, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    dim3 grid(img_grid(HW, PL, N), N);
    const size_t lds = (size_t)256 * 2 * epv * sizeof(float);
    if (dtype == UNET_BF16) hipLaunchKernelGGL((pair_reduce_kernel<__bf16>), grid, dim3(256), lds, s, (const __bf16*)u, (const __bf16*)z, HW, C, acc_ws);
    else hipLaunchKernelGGL((pair_reduce_kernel<float>), grid, dim3(256), lds, s, (const float*)u, (const float*)z, HW, C, acc_ws);
    UNET_LAUNCH_CHECK();
    float* cA = coef_ws; float* cB = coef_ws + (size_t)N * C; float* cD = coef_ws + 2 * (size_t)N * C;
    double* dgb = acc_ws + (size_t)N * C * 2;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N), dim3(256), sizeof(double) * 2 * groups, s, acc_ws, C, groups,
                       (double)HW * (C / groups), gamma, mean, invstd, cA, cB, cD, dgb);
    UNET_LAUNCH_CHECK();
    if (dtype == UNET_BF16) hipLaunchKernelGGL((affine2_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)u, (const __bf16*)z, (__bf16*)dz, cA, cB, cD, HW, C);
    else hipLaunchKernelGGL((affine2_kernel<float>), grid, dim3(256), 0, s, (const float*)u, (const float*)z, (float*)dz, cA, cB, cD, HW, C);
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(dgb_store_kernel, dim3((C + 255) / 256), dim3(256), 0, s, dgb, dgamma, dbeta, C);
    UNET_LAUNCH_CHECK();
    return 0;
}

// GroupNorm backward of nb branches that share the upstream gradient u (the per-branch arithmetic of unet_gn_bwd).
// acc_ws: nb * (N*C*2 + C*2) doubles; coef_ws: nb * 3*N*C floats.
extern "C" int unet_gn_bwd_multi(int dtype, int nb, const void* u, const void* const* z, void* const* dz, int N, long long HW, int C, int groups,
                                 const float* const* gamma, const float* const* mean, const float* const* invstd,
                                 float* const* dgamma, float* const* dbeta, double* acc_ws, float* coef_ws, void* stream) {
    if (ccheck(dtype, C, "gn_bwd_multi")) return -1;
    UNET_CHECK(nb >= 1 && nb <= MAX_BRANCH, "gn_bwd_multi: 1..%d branches", MAX_BRANCH);
    UNET_CHECK(groups > 0 && C % groups == 0 && N > 0 && HW > 0, "gn_bwd_multi: bad shape");
    UNET_CHECK(u && z && dz && gamma && mean && invstd && dgamma && dbeta && acc_ws && coef_ws, "gn_bwd_multi: null argument");
    GnMultiArgs a;
    for (int b = 0; b < MAX_BRANCH; ++b) {
        const bool on = b < nb;
        a.z[b] = on ? z[b] : nullptr; a.dz[b] = on ? dz[b] : nullptr; a.gamma[b] = on ? gamma[b] : nullptr;
        a.mean[b] = on ? mean[b] : nullptr; a.invstd[b] = on ? invstd[b] : nullptr;
        a.dgamma[b] = on ? dgamma[b] : nullptr; a.dbeta[b] = on ? dbeta[b] : nullptr;
        UNET_CHECK(!on || (a.z[b] && a.dz[b] && a.gamma[b] && a.mean[b] && a.invstd[b] && a.dgamma[b] && a.dbeta[b]), "gn_bwd_multi: null branch argument");
    }
    a.nb = nb;
    hipStream_t s = (hipStream_t)stream;
    const size_t acc_stride = (size_t)N * C * 2 + (size_t)C * 2, coef_stride = (size_t)3 * N * C;
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * acc_stride * nb, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    dim3 grid(img_grid(HW, PL, N), N);
    const size_t lds = (size_t)256 * epv * sizeof(float);
    if (dtype == UNET_BF16) hipLaunchKernelGGL((multi_pair_reduce_kernel<__bf16>), grid, dim3(256), lds, s, a, (const __bf16*)u, HW, C, acc_ws, acc_stride);
    else hipLaunchKernelGGL((multi_pair_reduce_kernel<float>), grid, dim3(256), lds, s, a, (const float*)u, HW, C, acc_ws, acc_stride);
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_finalize_multi_kernel, dim3(N, nb), dim3(256), sizeof(double) * 2 * groups, s, a, acc_ws, acc_stride, N, C, groups,
                       (double)HW * (C / groups), coef_ws, coef_stride);
    UNET_LAUNCH_CHECK();
    if (dtype == UNET_BF16) hipLaunchKernelGGL((multi_affine2_kernel<__bf16>), grid, dim3(256), 0, s, a, (const __bf16*)u, coef_ws, coef_stride, N, HW, C);
    else hipLaunchKernelGGL((multi_affine2_kernel<float>), grid, dim3(256), 0, s, a, (const float*)u, coef_ws, coef_stride, N, HW, C);
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(dgb_store_multi_kernel, dim3((C + 255) / 256, nb), dim3(256), 0, s, a, acc_ws, acc_stride, N, C);
    UNET_LAUNCH_CHECK();
    return 0;
}
