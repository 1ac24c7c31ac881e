// The two convolutions of the U-Net that do not map to MFMA tiles (gfx950), plus weight packing.
//
//   first conv  : Conv2d(Cin<=4 -> Cout, kxk)   (reference models/unet.py:30 with in_channels=1)
//                 bandwidth-bound on its OUTPUT (reads 4 B/pixel, writes Cout*sizeof(T))
//   final conv  : Conv2d(C -> n_classes<=8, 1x1) (reference models/unet.py:39,62), raw NCHW fp32 logits
//                 bandwidth-bound on its INPUT
//   pack        : fp32 OIHW / (Cin,Cout,2,2) master weights -> [tap][Cout][Cin] compute-dtype images
#include "common.h"
#include "bn_shared.h"



// ------------------------------------------------------------------------------------------------
// first conv forward: x fp32 [N,H,W,Cin], w fp32 OIHW, z T [N,H,W,Cout] + BN partial rows [grid][Cout][2]
// PX consecutive pixels of a row per thread: a tap's weights are read from LDS once for all of them (PX = 4 when W is a
// multiple of 4: the 5x5 first conv of a deployed RepUNet at 16 x 1024^2 went 3.55 -> 2.09 ms; 8 pixels: 2.68; PX = 1 otherwise).
template <typename T, int KH, int KW, int PX>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ z,
                                                             float* __restrict__ stats, int N, int H, int W, int Cin, int Cout) {
    constexpr int EPV = TT<T>::EPV, NT = KH * KW, PH = KH / 2, PW = KW / 2, WIN = KW + PX - 1, MAXCI = 4;
    extern __shared__ float sm[];                       // weights [tap][ci][Cout] then reduction scratch
    const int G = Cout / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    for (int i = threadIdx.x; i < NT * Cin * Cout; i += 256) {
        const int co = i % Cout, r = i / Cout, ci = r % Cin, tap = r / Cin;
        sm[i] = w[((size_t)co * Cin + ci) * NT + tap];
    }
    __syncthreads();
    float s1[EPV], s2[EPV], b[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s1[e] = s2[e] = 0.f; b[e] = (bias != nullptr && active) ? bias[cg * EPV + e] : 0.f; }
    // grid (gx, N): a workgroup stays inside image blockIdx.y, so its statistics row belongs to one
    // sample (GroupNorm needs per-sample statistics); rows are ordered [n][bx]
    const int Wg = (W + PX - 1) / PX;
    const long long groups = (long long)H * Wg;
    if (active) {
        for (long long q = (long long)blockIdx.x * PL + pl; q < groups; q += (long long)gridDim.x * PL) {
            const int hq = (int)(q / Wg);
            const int w0 = (int)(q - (long long)hq * Wg) * PX;
            const long long nb = (long long)blockIdx.y * H;             // n*H
            float acc[PX][EPV];
#pragma unroll
            for (int px = 0; px < PX; ++px)
#pragma unroll
                for (int e = 0; e < EPV; ++e) acc[px][e] = b[e];
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const int hh = hq + kh - PH;
                if (hh < 0 || hh >= H) continue;
                // the row's input window (zero outside the image: a zero tap adds nothing, as skipping it did)
                float xw[MAXCI][WIN];
                for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
                    for (int j = 0; j < WIN; ++j) {
                        const int ww = w0 + j - PW;
                        xw[ci][j] = (ww >= 0 && ww < W) ? x[((nb + hh) * W + ww) * Cin + ci] : 0.f;
                    }
#pragma unroll
                for (int kw = 0; kw < KW; ++kw)
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float* wv = sm + ((kh * KW + kw) * Cin + ci) * Cout + cg * EPV;
                        float wr[EPV];
#pragma unroll
                        for (int e = 0; e < EPV; ++e) wr[e] = wv[e];
#pragma unroll
                        for (int px = 0; px < PX; ++px)
#pragma unroll
                            for (int e = 0; e < EPV; ++e) acc[px][e] = fmaf(xw[ci][px + kw], wr[e], acc[px][e]);
                    }
            }
#pragma unroll
            for (int px = 0; px < PX; ++px) {
                if (w0 + px >= W) break;
#pragma unroll
                for (int e = 0; e < EPV; ++e) { s1[e] += acc[px][e]; s2[e] += acc[px][e] * acc[px][e]; }
                const long long p = ((long long)blockIdx.y * H + hq) * W + w0 + px;
                *(uint4*)(z + p * Cout + cg * EPV) = Vec16<T>::pack(acc[px]);
            }
        }
    }
    if (stats != nullptr) {
        __syncthreads();
        float* red = sm;                                 // [pl][Cout][2], reuses the weight area (>= 256*EPV*2 floats allotted)
        if (active) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                red[(pl * Cout + cg * EPV + e) * 2 + 0] = s1[e];
                red[(pl * Cout + cg * EPV + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < Cout * 2; i += 256) {
            float s = 0.f;
            for (int p = 0; p < PL; ++p) s += red[p * Cout * 2 + i];
            stats[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * Cout * 2 + i] = s;
        }
    }
}

// first conv weight grad: dW[co][ci][tap] = sum_p dz[p][co] * x[p+tap][ci]; one launch handles input
// channel `ci`; fp64 atomics into acc[Cout][NT] then converted by the caller.
template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const T* __restrict__ dz, const float* __restrict__ x,
                                                               int N, int H, int W, int Cin, int ci, int Cout,
                                                               double* __restrict__ acc) {
    constexpr int EPV = TT<T>::EPV, NT = KH * KW, PH = KH / 2, PW = KW / 2;
    extern __shared__ float sm[];
    const int G = Cout / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    float part[NT][EPV];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < EPV; ++e) part[t][e] = 0.f;
    const long long M = (long long)N * H * W;
    if (active) {
        for (long long p = (long long)blockIdx.x * PL + pl; p < M; p += (long long)gridDim.x * PL) {
            const int wq = (int)(p % W);
            const long long r = p / W;
            const int hq = (int)(r % H);
            const long long nb = (r / H) * H;
            float g[EPV];
            Vec16<T>::unpack(*(const uint4*)(dz + p * Cout + cg * EPV), g);
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const int hh = hq + kh - PH;
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    const int ww = wq + kw - PW;
                    const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                    const float xv = ok ? x[((nb + hh) * W + ww) * Cin + ci] : 0.f;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) part[kh * KW + kw][e] = fmaf(g[e], xv, part[kh * KW + kw][e]);
                }
            }
        }
    }
    // reduce over pixel lanes one tap at a time: LDS [pl][cg][EPV]
    const int per = G * EPV;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (active) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) sm[(pl * G + cg) * EPV + e] = part[t][e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += 256) {
            float s = 0.f;
            for (int p = 0; p < PL; ++p) s += sm[p * per + i];
            atomicAdd(&acc[(size_t)i * NT + t], (double)s);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Cin == 1, 3x3 (the reference's CT configuration, models/unet.py:30 with in_channels=1): both kernels keep
// the thread's weights / weight-gradient partials in registers and walk DOWN a column strip with a sliding
// 3x3 window, so a pixel costs 3 input loads instead of 9 and no LDS traffic.  Thread = (pixel lane, group
// of EPV output channels); a workgroup covers PL = 256/(Cout/EPV) adjacent columns x FIRST_R rows per tile.
// 4 waves per SIMD (128 VGPRs, 6 spilled outside the row loop) instead of the 3 the 136 registers of the natural allocation
// allow: 428 -> 358 us at 32 x 512^2 x 64 (the kernel is latency-bound: one dependent 3-row load chain per thread).
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_first1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, T* __restrict__ z,
                                                              float* __restrict__ stats, int H, int W, int Cout) {
    constexpr int EPV = TT<T>::EPV;
    __shared__ float red[256 * EPV * 2];
    const int G = Cout / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    float wr[9][EPV], b[EPV], s1[EPV], s2[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        s1[e] = s2[e] = 0.f;
        b[e] = (bias != nullptr && active) ? bias[cg * EPV + e] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][e] = active ? w[(cg * EPV + e) * 9 + t] : 0.f;
    }
    const int tiles_w = (W + PL - 1) / PL, tiles_h = (H + FIRST_R - 1) / FIRST_R;
    const float* xi = x + (long long)blockIdx.y * H * W;
    T* zi = z + (long long)blockIdx.y * H * W * Cout;
    for (int tile = blockIdx.x; tile < tiles_w * tiles_h; tile += gridDim.x) {
        const int th = tile / tiles_w, tw = tile - th * tiles_w;
        const int h0 = th * FIRST_R, wq = tw * PL + pl;
        if (!active || wq >= W) continue;
        const int h1 = min(h0 + FIRST_R, H);
        float ra[3], rb[3], rc[3], rn[3];                 // rows h-1, h, h+1 and the prefetched row h+2
        first_row3(xi, h0 - 1, wq, H, W, ra);
        first_row3(xi, h0, wq, H, W, rb);
        first_row3(xi, h0 + 1, wq, H, W, rc);
        for (int h = h0; h < h1; ++h) {
            first_row3(xi, h + 2, wq, H, W, rn);          // in flight under this row's 72 FMAs
            float acc[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float v = b[e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    v = fmaf(ra[k], wr[k][e], v);
                    v = fmaf(rb[k], wr[3 + k][e], v);
                    v = fmaf(rc[k], wr[6 + k][e], v);
                }
                acc[e] = v;
                s1[e] += v; s2[e] = fmaf(v, v, s2[e]);
            }
            if (z != nullptr) *(uint4*)(zi + ((long long)h * W + wq) * Cout + cg * EPV) = Vec16<T>::pack(acc);      // NULL: statistics only
#pragma unroll
            for (int k = 0; k < 3; ++k) { ra[k] = rb[k]; rb[k] = rc[k]; rc[k] = rn[k]; }
        }
    }
    if (stats != nullptr) {
        if (active) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                red[(pl * Cout + cg * EPV + e) * 2 + 0] = s1[e];
                red[(pl * Cout + cg * EPV + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < Cout * 2; i += 256) {
            float s = 0.f;
            for (int p = 0; p < PL; ++p) s += red[p * Cout * 2 + i];
            stats[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * Cout * 2 + i] = s;
        }
    }
}

// dW[co][tap] = sum_p dz[p][co] * x[p + tap]; fp64 atomics into acc[Cout][9]
template <typename T>
__global__ __launch_bounds__(256) void conv_first1_wgrad_kernel(const T* __restrict__ dz, const float* __restrict__ x,
                                                                int N, int H, int W, int Cout, double* __restrict__ acc) {
    constexpr int EPV = TT<T>::EPV;
    __shared__ float sm[256 * EPV];               // one tap's partials [pixel lane][channel]
    const int G = Cout / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    float part[9][EPV];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < EPV; ++e) part[t][e] = 0.f;
    const int tiles_w = (W + PL - 1) / PL, tiles_h = (H + FIRST_R - 1) / FIRST_R;
    const int per_img = tiles_w * tiles_h;
    for (long long tile = blockIdx.x; tile < (long long)per_img * N; tile += gridDim.x) {
        const int n = (int)(tile / per_img), tt = (int)(tile - (long long)n * per_img);
        const int th = tt / tiles_w, tw = tt - th * tiles_w;
        const int h0 = th * FIRST_R, wq = tw * PL + pl;
        if (!active || wq >= W) continue;
        const int h1 = min(h0 + FIRST_R, H);
        const float* xi = x + (long long)n * H * W;
        const T* di = dz + (long long)n * H * W * Cout;
        float ra[3], rb[3], rc[3], rn[3];
        first_row3(xi, h0 - 1, wq, H, W, ra);
        first_row3(xi, h0, wq, H, W, rb);
        first_row3(xi, h0 + 1, wq, H, W, rc);
        uint4 gq = *(const uint4*)(di + ((long long)h0 * W + wq) * Cout + cg * EPV);
        for (int h = h0; h < h1; ++h) {
            first_row3(xi, h + 2, wq, H, W, rn);          // next row's inputs and gradient in flight under the FMAs
            const uint4 gn = h + 1 < h1 ? *(const uint4*)(di + ((long long)(h + 1) * W + wq) * Cout + cg * EPV) : gq;
            float g[EPV];
            Vec16<T>::unpack(gq, g);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    part[k][e] = fmaf(g[e], ra[k], part[k][e]);
                    part[3 + k][e] = fmaf(g[e], rb[k], part[3 + k][e]);
                    part[6 + k][e] = fmaf(g[e], rc[k], part[6 + k][e]);
                }
            gq = gn;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ra[k] = rb[k]; rb[k] = rc[k]; rc[k] = rn[k]; }
        }
    }
    // reduce over pixel lanes one tap at a time through LDS [pl][cg][EPV] (as the generic kernel)
    const int per = G * EPV;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (active) {
#pragma unroll
            for (int e = 0; e < EPV; ++e) sm[(pl * G + cg) * EPV + e] = part[t][e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += 256) {
            float s = 0.f;
            for (int p = 0; p < PL; ++p) s += sm[p * per + i];
            atomicAdd(&acc[(size_t)i * 9 + t], (double)s);
        }
        __syncthreads();
    }
}

__global__ void first_wgrad_store_kernel(const double* __restrict__ acc, float* __restrict__ dw, int Cout, int Cin, int ci, int NT) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * NT) return;
    const int co = i / NT, t = i - co * NT;
    dw[((size_t)co * Cin + ci) * NT + t] = (float)acc[i];
}

// ------------------------------------------------------------------------------------------------
// final 1x1 conv forward: x T [M][ld] -> logits fp32 NCHW [N][K][HW]; LP lanes cooperate on one pixel
template <typename T, int K>
__global__ __launch_bounds__(256) void conv_out_fwd_kernel(const T* __restrict__ x, int ldx, int xoff,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ logits, long long M, long long HW, int C, int LP) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV;
    const int sub = threadIdx.x & (LP - 1);               // lane inside the pixel group
    const int ppb = 256 / LP;                             // pixels per block pass
    const int pslot = threadIdx.x / LP;
    float wr[K][EPV];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < EPV; ++e) wr[k][e] = (sub < G) ? w[k * C + sub * EPV + e] : 0.f;
    float bk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) bk[k] = bias != nullptr ? bias[k] : 0.f;
    // every lane of a wave runs the same trip count so the shuffles below see all lanes
    const long long iters = (M + (long long)gridDim.x * ppb - 1) / ((long long)gridDim.x * ppb);
    for (long long it = 0; it < iters; ++it) {
        const long long p = (it * gridDim.x + blockIdx.x) * ppb + pslot;
        float f[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) f[e] = 0.f;
        if (p < M && sub < G) Vec16<T>::unpack(*(const uint4*)(x + p * ldx + xoff + sub * EPV), f);
        float d[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) s = fmaf(f[e], wr[k][e], s);
            for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o);
            d[k] = s + bk[k];
        }
        if (p < M && sub == 0) {
            const long long n = p / HW, hw = p - n * HW;
#pragma unroll
            for (int k = 0; k < K; ++k) logits[(n * K + k) * HW + hw] = d[k];
        }
    }
}

// final conv backward: dl fp32 NCHW -> dx T [M][C] (dense), dW[K][C], db[K] via fp64 acc [K*C + K]
template <typename T, int K>
__global__ __launch_bounds__(256) void conv_out_bwd_kernel(const float* __restrict__ dl, const T* __restrict__ x, int ldx, int xoff,
                                                           const float* __restrict__ w, T* __restrict__ dx,
                                                           long long M, long long HW, int C, double* __restrict__ acc) {
    constexpr int EPV = TT<T>::EPV;
    extern __shared__ float sm[];
    const int G = C / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    const bool active = pl < PL;
    float wr[K][EPV], pw[K][EPV], pb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        pb[k] = 0.f;
#pragma unroll
        for (int e = 0; e < EPV; ++e) { wr[k][e] = active ? w[k * C + cg * EPV + e] : 0.f; pw[k][e] = 0.f; }
    }
    if (active) {
        for (long long p = (long long)blockIdx.x * PL + pl; p < M; p += (long long)gridDim.x * PL) {
            const long long n = p / HW, hw = p - n * HW;
            float f[EPV], o[EPV], g[K];
            Vec16<T>::unpack(*(const uint4*)(x + p * ldx + xoff + cg * EPV), f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) o[e] = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                g[k] = dl[(n * K + k) * HW + hw];
                pb[k] += g[k];
#pragma unroll
                for (int e = 0; e < EPV; ++e) { o[e] = fmaf(g[k], wr[k][e], o[e]); pw[k][e] = fmaf(g[k], f[e], pw[k][e]); }
            }
            *(uint4*)(dx + p * C + cg * EPV) = Vec16<T>::pack(o);
        }
    }
    // reduce dW partials over pixel lanes: LDS [pl][cg][K*EPV]
    if (active) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < EPV; ++e) sm[(pl * G + cg) * K * EPV + k * EPV + e] = pw[k][e];
    }
    __syncthreads();
    const int per = G * K * EPV;
    for (int i = threadIdx.x; i < per; i += 256) {
        float s = 0.f;
        for (int p = 0; p < PL; ++p) s += sm[p * per + i];
        const int cgi = i / (K * EPV), r = i - cgi * K * EPV, k = r / EPV, e = r - k * EPV;
        atomicAdd(&acc[(size_t)k * C + cgi * EPV + e], (double)s);
    }
    __syncthreads();
    // db: only the cg == 0 thread of each pixel lane carries a distinct copy
    if (active && cg == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sm[pl * K + k] = pb[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float s = 0.f;
        for (int p = 0; p < PL; ++p) s += sm[p * K + threadIdx.x];
        atomicAdd(&acc[(size_t)K * C + threadIdx.x], (double)s);
    }
}

__global__ void acc_store_kernel(const double* __restrict__ acc, float* __restrict__ a, int na, float* __restrict__ b, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = (float)acc[i];
    else if (i < na + nb && b != nullptr) b[i - na] = (float)acc[i];
}

// ------------------------------------------------------------------------------------------------
// weight packing (one thread per destination element)
//   mode 0 conv fwd   : dst[(tap*O + o)*I + i]          = src[(o*I + i)*T + tap]
//   mode 1 conv dgrad : dst[((T-1-tap)*I + i)*O + o]     = src[(o*I + i)*T + tap]
//   mode 2 convT fwd  : src (Ci,Co,2,2): dst[(q*Co + c)*Ci + ci]            = src[(ci*Co + c)*4 + q]
//   mode 3 convT dgrad: dst[(i2*Ci + ci)*(2*Co) + j2*Co + c]                 = src[(ci*Co + c)*4 + i2*2 + j2]
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, int mode, int O, int I, int NT) {
    const size_t total = (size_t)O * I * NT;
    for (size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (size_t)gridDim.x * blockDim.x) {
        size_t s;
        if (mode == 0) {
            const int i = (int)(d % I); const size_t r = d / I; const int o = (int)(r % O); const int tap = (int)(r / O);
            s = ((size_t)o * I + i) * NT + tap;
        } else if (mode == 1) {
            const int o = (int)(d % O); const size_t r = d / O; const int i = (int)(r % I); const int tf = (int)(r / I);
            s = ((size_t)o * I + i) * NT + (NT - 1 - tf);
        } else if (mode == 2) {
            // O = Ci (convT in), I = Co (convT out), NT = 4
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

// convT weight-grad slab [nsplit][tap=i2][Ci][2*Co] -> dW (Ci,Co,2,2)
// KL lanes share an element: lane kl sums splits kl, kl + KL, ... in order, lane 0 adds the KL partials in lane order (fixed
// order, bitwise reproducible).  The top decoder level has 32768 elements but 256 splits: 8 lanes per element there.
template <int KL>
__global__ __launch_bounds__(256) void convT_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int nsplit, int Ci, int Co) {
    constexpr int EL = 256 / KL;
    __shared__ float part[KL][EL];
    const size_t total = (size_t)2 * Ci * 2 * Co;
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    for (size_t e0 = (size_t)blockIdx.x * EL; e0 < total; e0 += (size_t)gridDim.x * EL) {
        const size_t e = e0 + el;
        float s = 0.f;
        if (e < total)
            for (int k = kl; k < nsplit; k += KL) s += slab[(size_t)k * total + e];
        if (KL > 1) {
            part[kl][el] = s;
            __syncthreads();
            if (kl == 0)
                for (int q = 1; q < KL; ++q) s += part[q][el];
            __syncthreads();
        }
        if (kl == 0 && e < total) {
            const int cc = (int)(e % (2 * Co)); const size_t r = e / (2 * Co); const int ci = (int)(r % Ci); const int i2 = (int)(r / Ci);
            const int j2 = cc / Co, c = cc - j2 * Co;
            out[((size_t)ci * Co + c) * 4 + i2 * 2 + j2] = s;
        }
    }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = from_f32<T>(src[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = to_f32(src[i]);
}

// ------------------------------------------------------------------------------------------------
static inline int grid_for(long long M, int PL) {
    long long g = (M + PL - 1) / PL;
    if (g > MAX_GRID) g = MAX_GRID;
    if (g < 1) g = 1;
    return (int)g;
}

static int first_gx(int dtype, int N, int H, int W, int Cout) {
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    if (Cout <= 0 || Cout % epv || N <= 0) return -1;
    const int G = Cout / epv;
    if (G > 256) return -1;
    int cap = MAX_GRID / N; if (cap < 1) cap = 1;
    long long g = ((long long)H * W + 256 / G - 1) / (256 / G);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
// rows of per-image statistics the first conv writes: N * (rows per image), ordered [n][bx]
extern "C" int unet_conv_first_rows(int dtype, int N, int H, int W, int Cout) {
    const int gx = first_gx(dtype, N, H, W, Cout);
    return gx < 0 ? -1 : gx * N;
}

extern "C" int unet_conv_first_fwd(int dtype, int kh, int kw, const float* x, const float* w, const float* bias, void* z, float* stats,
                                   int N, int H, int W, int Cin, int Cout, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    UNET_CHECK(Cin >= 1 && Cin <= 4, "first-conv kernel serves Cin <= 4 (got %d)", Cin);
    UNET_CHECK(Cout % 8 == 0 && Cout > 0, "Cout must be a multiple of 8");
    UNET_CHECK(x != nullptr && w != nullptr, "first conv: null argument");
    UNET_CHECK(z != nullptr || (kh == 3 && kw == 3 && Cin == 1 && stats != nullptr),
               "first conv: z may be NULL (statistics only) for the Cin = 1 3x3 kernel with a statistics buffer");
    const int rows = unet_conv_first_rows(dtype, N, H, W, Cout);
    UNET_CHECK(rows > 0, "Cout=%d too wide for the first-conv kernel", Cout);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    size_t lds = (size_t)kh * kw * Cin * Cout * sizeof(float);
    const size_t red = (size_t)256 * epv * 2 * sizeof(float);
    if (lds < red) lds = red;
    UNET_CHECK(lds <= 65536, "first-conv weights do not fit LDS");
    hipStream_t s = (hipStream_t)stream;
    const int gx = rows / N;
#define LF(T, A, B, P) hipLaunchKernelGGL((conv_first_fwd_kernel<T, A, B, P>), dim3(gx, N), dim3(256), lds, s, x, w, bias, (T*)z, stats, N, H, W, Cin, Cout)
#define LFD(A, B) do { if (W % 4 == 0) { if (dtype == UNET_BF16) LF(__bf16, A, B, 4); else LF(float, A, B, 4); } \
                       else { if (dtype == UNET_BF16) LF(__bf16, A, B, 1); else LF(float, A, B, 1); } } while (0)
    if (kh == 3 && kw == 3 && Cin == 1) {
        if (dtype == UNET_BF16) hipLaunchKernelGGL((conv_first1_fwd_kernel<__bf16>), dim3(gx, N), dim3(256), 0, s, x, w, bias, (__bf16*)z, stats, H, W, Cout);
        else hipLaunchKernelGGL((conv_first1_fwd_kernel<float>), dim3(gx, N), dim3(256), 0, s, x, w, bias, (float*)z, stats, H, W, Cout);
    }
    else if (kh == 3 && kw == 3) LFD(3, 3);
    else if (kh == 5 && kw == 5) LFD(5, 5);
    else if (kh == 3 && kw == 1) LFD(3, 1);
    else if (kh == 1 && kw == 3) LFD(1, 3);
    else if (kh == 5 && kw == 1) LFD(5, 1);
    else if (kh == 1 && kw == 5) LFD(1, 5);
    else { unet_set_error("first conv: unsupported kernel %dx%d", kh, kw); return -1; }
#undef LF
#undef LFD
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_conv_first_wgrad(int dtype, int kh, int kw, const void* dz, const float* x, float* dw,
                                     int N, int H, int W, int Cin, int Cout, double* acc_ws /* Cout*kh*kw doubles */, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    UNET_CHECK(Cin >= 1 && Cin <= 4 && Cout % 8 == 0, "first-conv wgrad: Cin<=4, Cout%%8==0");
    const int rows = unet_conv_first_rows(dtype, N, H, W, Cout);
    UNET_CHECK(rows > 0, "Cout=%d too wide", Cout);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const int NT = kh * kw;
    const size_t lds = (size_t)256 * epv * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int grid = rows < 1024 ? rows : 1024;
    for (int ci = 0; ci < Cin; ++ci) {
        if (hipMemsetAsync(acc_ws, 0, sizeof(double) * Cout * NT, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
#define LW(T, A, B) hipLaunchKernelGGL((conv_first_wgrad_kernel<T, A, B>), dim3(grid), dim3(256), lds, s, (const T*)dz, x, N, H, W, Cin, ci, Cout, acc_ws)
#define LWD(A, B) do { if (dtype == UNET_BF16) LW(__bf16, A, B); else LW(float, A, B); } while (0)
        if (kh == 3 && kw == 3 && Cin == 1) {
            if (dtype == UNET_BF16) hipLaunchKernelGGL((conv_first1_wgrad_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)dz, x, N, H, W, Cout, acc_ws);
            else hipLaunchKernelGGL((conv_first1_wgrad_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dz, x, N, H, W, Cout, acc_ws);
        }
        else if (kh == 3 && kw == 3) LWD(3, 3);
        else if (kh == 3 && kw == 1) LWD(3, 1);
        else if (kh == 1 && kw == 3) LWD(1, 3);
        else if (kh == 5 && kw == 1) LWD(5, 1);
        else if (kh == 1 && kw == 5) LWD(1, 5);
        else if (kh == 5 && kw == 5) LWD(5, 5);
        else { unet_set_error("first conv wgrad: unsupported kernel %dx%d", kh, kw); return -1; }
#undef LW
#undef LWD
        UNET_LAUNCH_CHECK();
        hipLaunchKernelGGL(first_wgrad_store_kernel, dim3((Cout * NT + 255) / 256), dim3(256), 0, s, acc_ws, dw, Cout, Cin, ci, NT);
        UNET_LAUNCH_CHECK();
    }
    return 0;
}

static int lanes_per_pixel(int G) { int lp = 1; while (lp < G) lp <<= 1; return lp; }

extern "C" int unet_conv_out_fwd(int dtype, const void* x, int ldx, int xoff, const float* w, const float* bias, float* logits,
                                 long long M, long long HW, int C, int K, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    UNET_CHECK(C % 8 == 0 && C / epv <= 64, "final conv: C=%d must be a multiple of 8 and at most %d", C, 64 * epv);
    UNET_CHECK(K >= 1 && K <= 8, "final conv: n_classes must be 1..8 (got %d)", K);
    UNET_CHECK(M % HW == 0, "final conv: M must be N*H*W");
    const int LP = lanes_per_pixel(C / epv);
    const int grid = grid_for(M, 256 / LP);
    hipStream_t s = (hipStream_t)stream;
#define LO(T, KK) hipLaunchKernelGGL((conv_out_fwd_kernel<T, KK>), dim3(grid), dim3(256), 0, s, (const T*)x, ldx, xoff, w, bias, logits, M, HW, C, LP)
#define LOK(KK) case KK: if (dtype == UNET_BF16) LO(__bf16, KK); else LO(float, KK); break;
    switch (K) { LOK(1) LOK(2) LOK(3) LOK(4) LOK(5) LOK(6) LOK(7) LOK(8) }
#undef LO
#undef LOK
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_conv_out_bwd(int dtype, const float* dlogits, const void* x, int ldx, int xoff, const float* w,
                                 void* dx, float* dw, float* db, long long M, long long HW, int C, int K,
                                 double* acc_ws /* K*C + K doubles */, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    UNET_CHECK(C % 8 == 0 && C / epv <= 256, "final conv bwd: bad C=%d", C);
    UNET_CHECK(K >= 1 && K <= 8, "final conv: n_classes must be 1..8 (got %d)", K);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * (K * C + K), s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    const int PL = 256 / (C / epv);
    const int grid = grid_for(M, PL) < 1024 ? grid_for(M, PL) : 1024;
    const size_t lds = (size_t)256 * K * epv * sizeof(float);
#define LB(T, KK) hipLaunchKernelGGL((conv_out_bwd_kernel<T, KK>), dim3(grid), dim3(256), lds, s, dlogits, (const T*)x, ldx, xoff, w, (T*)dx, M, HW, C, acc_ws)
#define LBK(KK) case KK: if (dtype == UNET_BF16) LB(__bf16, KK); else LB(float, KK); break;
    switch (K) { LBK(1) LBK(2) LBK(3) LBK(4) LBK(5) LBK(6) LBK(7) LBK(8) }
#undef LB
#undef LBK
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(acc_store_kernel, dim3((K * C + K + 255) / 256), dim3(256), 0, s, acc_ws, dw, K * C, db, K);
    UNET_LAUNCH_CHECK();
    return 0;
}

// Inference-time fold of BatchNorm2d (running statistics) into the conv in front of it, DoubleConv of
// /root/reference/models/unet.py:9-16: w' = w * s, b' = beta + (b - mean) * s with s = gamma / sqrt(var + eps)
// -- the algebra of the reference's mergeBN (models/reparm_unet.py:28-31) applied to the plain UNet.
__global__ void fold_bn_conv_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv,
                                    float eps, int O, int per, float* __restrict__ wo, float* __restrict__ bo) {
    const size_t total = (size_t)O * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int o = (int)(i / per);
        const float sc = gamma[o] * (1.0f / sqrtf(rv[o] + eps));
        wo[i] = w[i] * sc;
        if (i - (size_t)o * per == 0) bo[o] = beta[o] + ((b != nullptr ? b[o] : 0.f) - rm[o]) * sc;
    }
}

extern "C" int unet_fold_bn_conv(const float* w, const float* bias, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, float eps, int O, int per,
                                 float* w_out, float* b_out, void* stream) {
    UNET_CHECK(w && gamma && beta && running_mean && running_var && w_out && b_out, "fold_bn_conv: null argument");
    UNET_CHECK(O > 0 && per > 0, "fold_bn_conv: bad shape %d x %d", O, per);
    const size_t total = (size_t)O * per;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fold_bn_conv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, bias, gamma, beta,
                       running_mean, running_var, eps, O, per, w_out, b_out);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_pack_weights(int dtype, int mode, const float* src, void* dst, int O, int I, int ntaps, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    UNET_CHECK(mode >= 0 && mode <= 3, "bad pack mode %d", mode);
    if (mode >= 2) UNET_CHECK(ntaps == 4, "convT pack needs 4 taps");
    const size_t total = (size_t)O * I * ntaps;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16) hipLaunchKernelGGL((pack_weights_kernel<__bf16>), dim3(grid), dim3(256), 0, s, src, (__bf16*)dst, mode, O, I, ntaps);
    else hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, s, src, (float*)dst, mode, O, I, ntaps);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_convT_wgrad_reduce(const float* slab, float* dw, int nsplit, int Ci, int Co, void* stream) {
    const size_t total = (size_t)4 * Ci * Co;
    UNET_CHECK(slab != nullptr && dw != nullptr && nsplit > 0 && Ci > 0 && Co > 0, "convT_wgrad_reduce: bad argument");
    if (total < 262144 && nsplit >= 16) {
        const int grid = (int)((total + 31) / 32);
        hipLaunchKernelGGL(convT_wgrad_reduce_kernel<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, dw, nsplit, Ci, Co);
    } else {
        int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(convT_wgrad_reduce_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, dw, nsplit, Ci, Co);
    }
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_cast(int dtype, int to_f32_dir, const void* src, void* dst, long long n, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16) {
        if (to_f32_dir) hipLaunchKernelGGL((cast_to_f32_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)src, (float*)dst, (size_t)n);
        else hipLaunchKernelGGL((cast_from_f32_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const float*)src, (__bf16*)dst, (size_t)n);
    } else {
        if (to_f32_dir) hipLaunchKernelGGL((cast_to_f32_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, (size_t)n);
        else hipLaunchKernelGGL((cast_from_f32_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, (size_t)n);
    }
    UNET_LAUNCH_CHECK();
    return 0;
}
