// Bandwidth-bound NHWC kernels of the U-Net hot path (gfx950): BatchNorm statistics / apply /
// backward fused with ReLU, and 2x2 max-pool with a 2-bit argmax code.
//
// Thread mapping shared by every per-channel kernel here: a thread owns ONE 16-byte channel group
// (8 bf16 / 4 f32 channels) and walks pixels; a 256-thread workgroup covers PL = 256/G pixels x
// G = C/EPV groups, so consecutive lanes read consecutive 16-byte vectors (fully coalesced) and
// per-channel constants live in registers.
//
// Reference semantics followed: nn.BatchNorm2d + nn.ReLU(inplace) in DoubleConv
// (/root/reference/models/unet.py:11-12,14-15) and nn.MaxPool2d(2,2) (models/unet.py:26,47).
#include "common.h"

#include <mutex>

#include "bn_shared.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// BN statistics: reduce the conv epilogue's per-tile (sum, sumsq) rows into fp64 totals.
// grid (ceil(C/32), S); rows [nrows][C][2] fp32; acc [C][2] fp64 (zeroed by the launcher).
__global__ void bn_stats_reduce_kernel(const float* __restrict__ rows, int nrows, int C, double* __restrict__ acc) {
    const int ch = threadIdx.x & 31, pl = threadIdx.x >> 5;     // 32 channels x 8 row lanes
    const int c = blockIdx.x * 32 + ch;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int r = blockIdx.y * 8 + pl; r < nrows; r += gridDim.y * 8) {
            const float2 v = *(const float2*)(rows + ((size_t)r * C + c) * 2);
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
    }
    __shared__ double red[8][32][2];
    red[pl][ch][0] = s1;
    red[pl][ch][1] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int ch2 = threadIdx.x >> 1, k = threadIdx.x & 1;
        double s = 0.0;
        for (int p = 0; p < 8; ++p) s += red[p][ch2][k];
        const int c2 = blockIdx.x * 32 + ch2;
        if (c2 < C) atomicAdd(&acc[c2 * 2 + k], s);
    }
}

// One thread per channel: batch mean / biased var -> (scale, shift, mean, invstd) and the running
// statistics update r = (1-m) r + m stat with UNBIASED variance, num_batches_tracked += 1.
__device__ __forceinline__ void bn_finalize_channel(int c, double sum, double sumsq, double count,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    long long* __restrict__ nbt, float momentum, float eps,
                                                    float* __restrict__ scale, float* __restrict__ shift,
                                                    float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    if (c == 0 && nbt != nullptr) *nbt += 1;
    const double mean = sum / count;
    double var = sumsq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean != nullptr) {
        const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_finalize_kernel(const double* __restrict__ acc, double count, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ nbt, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    bn_finalize_channel(c, acc[c * 2], acc[c * 2 + 1], count, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                        scale, shift, mean_out, invstd_out);
}

// Up to 4096 rows (every layer of the 512^2 U-Net but none of its slowest: the 64-channel convs write one row per workgroup):
// one launch instead of memset + reduce + finalize.  A workgroup owns
// 8 channels (one 64-byte run per row), its 32 row lanes walk all rows, the partials are added in lane order (no atomics:
// deterministic) and the channel's first lane finalizes.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ rows, int nrows, int C, double count,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                long long* __restrict__ nbt, float momentum, float eps,
                                                                float* __restrict__ scale, float* __restrict__ shift,
                                                                float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;      // 8 channels x 32 row lanes
    const int c = blockIdx.x * 8 + ch;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        const float* col = rows + (size_t)c * 2;
        int r = pl;
        for (; r + 96 < nrows; r += 128) {                      // four independent loads in flight
            const float2 v0 = *(const float2*)(col + (size_t)r * C * 2);
            const float2 v1 = *(const float2*)(col + (size_t)(r + 32) * C * 2);
            const float2 v2 = *(const float2*)(col + (size_t)(r + 64) * C * 2);
            const float2 v3 = *(const float2*)(col + (size_t)(r + 96) * C * 2);
            s1 += (double)v0.x; s2 += (double)v0.y;
            s1 += (double)v1.x; s2 += (double)v1.y;
            s1 += (double)v2.x; s2 += (double)v2.y;
            s1 += (double)v3.x; s2 += (double)v3.y;
        }
        for (; r < nrows; r += 32) {
            const float2 v = *(const float2*)(col + (size_t)r * C * 2);
            s1 += (double)v.x; s2 += (double)v.y;
        }
    }
    __shared__ double red[32][8][2];
    red[pl][ch][0] = s1;
    red[pl][ch][1] = s2;
    __syncthreads();
    if (pl == 0 && c < C) {
        double a = 0.0, b = 0.0;
        for (int p = 0; p < 32; ++p) { a += red[p][ch][0]; b += red[p][ch][1]; }
        bn_finalize_channel(c, a, b, count, gamma, beta, running_mean, running_var, nbt, momentum, eps, scale, shift, mean_out, invstd_out);
    }
}

// eval mode: coefficients from the running statistics
__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift,
                                      float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
    if (mean_out != nullptr) mean_out[c] = rm[c];          // saved for a backward through frozen statistics
    if (invstd_out != nullptr) invstd_out[c] = invstd;
}

// STREAMING SHAPE of the per-channel kernels below (measured on 32 x 512^2 x 64 bf16, tools/exp_affine.py history in DESIGN
// 9): HBM throughput peaks with FEW resident workgroups that each keep SEVERAL 16-byte requests per thread in flight --
// 2 workgroups per CU x 2 pixels per trip reached 5.7 TB/s where 8 per CU x 1 reached 4.97 (and grids that are not a
// multiple of the 256 CUs lose up to 15 % to the uneven tail).  STREAM_U pixels are therefore loaded before any is used, and
// the launchers cap the grid at stream_grid_cap() workgroups.
#define STREAM_U 2

// y = relu(z*scale + shift)   (relu optional)
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ z, int ldz, int zoff,
                                                         T* __restrict__ y, int ldy, int yoff,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         long long M, int C) {
    constexpr int EPV = TT<T>::EPV;
    ChanMap m; m.init<T>(C);
    if (!m.active) return;
    float sc[EPV], sh[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { sc[e] = scale[m.cg * EPV + e]; sh[e] = shift[m.cg * EPV + e]; }
    const long long stride = (long long)gridDim.x * m.PL;
    for (long long p0 = (long long)blockIdx.x * m.PL + m.pl; p0 < M; p0 += stride * STREAM_U) {
        uint4 v[STREAM_U];
#pragma unroll
        for (int u = 0; u < STREAM_U; ++u) {
            const long long p = p0 + u * stride;
            if (p < M) v[u] = *(const uint4*)(z + p * ldz + zoff + m.cg * EPV);
        }
#pragma unroll
        for (int u = 0; u < STREAM_U; ++u) {
            const long long p = p0 + u * stride;
            if (p >= M) break;
            float f[EPV];
            Vec16<T>::unpack(v[u], f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                f[e] = fmaf(f[e], sc[e], sh[e]);
                if (RELU) f[e] = fmaxf(f[e], 0.f);
            }
            *(uint4*)(y + p * ldy + yoff + m.cg * EPV) = Vec16<T>::pack(f);
        }
    }
}

// Where BN backward takes dy from.  Plain: a stored tensor.  POOL: the gradient of an encoder level's output,
// dy[n,h,w,c] = dskip[n,h,w,c] + (code[n,h/2,w/2,c] == 2(h&1)+(w&1) ? dpool[n,h/2,w/2,c] : 0), i.e. the backward of
// MaxPool2d(2,2) plus the skip-connection gradient (models/unet.py:47,59) evaluated on the fly in fp32, so that dy is
// never written and read back (the two passes read 2.75 B/element more, the un-pool kernel's 4.75 B/element go away).
struct PoolSrc { const void* dpool; const uint8_t* code; int H, W; };

template <typename T> struct DyRaw { uint4 d, g; uint2 code; unsigned k; };

// issue the loads of one pixel's dy (nothing is unpacked yet, so several pixels' requests can be in flight together)
template <typename T, bool POOL>
__device__ __forceinline__ DyRaw<T> fetch_dy(const T* __restrict__ dy, int ldd, int doff, const PoolSrc& ps, long long p, int cg, int C) {
    constexpr int EPV = TT<T>::EPV;
    DyRaw<T> r;
    r.d = *(const uint4*)(dy + p * ldd + doff + cg * EPV);
    if (POOL) {
        const int pi = (int)p;
        const int w = pi % ps.W, t = pi / ps.W, h = t % ps.H, n = t / ps.H;
        const long long o = (((long long)n * (ps.H >> 1) + (h >> 1)) * (ps.W >> 1) + (w >> 1)) * C + cg * EPV;
        r.k = (unsigned)(2 * (h & 1) + (w & 1));
        r.g = *(const uint4*)((const T*)ps.dpool + o);
        if (EPV == 8) r.code = *(const uint2*)(ps.code + o);
        else { r.code.x = *(const uint32_t*)(ps.code + o); r.code.y = 0; }
    }
    return r;
}

template <typename T, bool POOL>
__device__ __forceinline__ void finish_dy(const DyRaw<T>& r, float (&fd)[TT<T>::EPV]) {
    constexpr int EPV = TT<T>::EPV;
    Vec16<T>::unpack(r.d, fd);
    if (POOL) {
        float g[EPV];
        Vec16<T>::unpack(r.g, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (((r.code.x >> (8 * e)) & 0xff) == r.k) fd[e] += g[e];
            if (EPV == 8 && ((r.code.y >> (8 * e)) & 0xff) == r.k) fd[(4 + e) % EPV] += g[(4 + e) % EPV];
        }
    }
}

// BN+ReLU backward, pass 1: S1 = sum dy', S2 = sum dy' * xhat with dy' = dy * [z*scale+shift > 0]
template <typename T, bool RELU, bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(PoolSrc ps, const T* __restrict__ dy, int ldd, int doff,
                                                            const T* __restrict__ z, int ldz, int zoff,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            long long M, int C, double* __restrict__ acc) {
    constexpr int EPV = TT<T>::EPV;
    ChanMap m; m.init<T>(C);
    float part[2 * EPV];
#pragma unroll
    for (int i = 0; i < 2 * EPV; ++i) part[i] = 0.f;
    if (m.active) {
        float sc[EPV], sh[EPV], mu[EPV], is[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const int c = m.cg * EPV + e;
            sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
        }
        const long long stride = (long long)gridDim.x * m.PL;
        for (long long p0 = (long long)blockIdx.x * m.PL + m.pl; p0 < M; p0 += stride * STREAM_U) {
            uint4 vz[STREAM_U];
            DyRaw<T> vd[STREAM_U];
#pragma unroll
            for (int u = 0; u < STREAM_U; ++u) {
                const long long p = p0 + u * stride;
                if (p < M) {
                    vz[u] = *(const uint4*)(z + p * ldz + zoff + m.cg * EPV);
                    vd[u] = fetch_dy<T, POOL>(dy, ldd, doff, ps, p, m.cg, C);
                }
            }
#pragma unroll
            for (int u = 0; u < STREAM_U; ++u) {
                if (p0 + u * stride >= M) break;
                float fz[EPV], fd[EPV];
                Vec16<T>::unpack(vz[u], fz);
                finish_dy<T, POOL>(vd[u], fd);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float d = (!RELU || fmaf(fz[e], sc[e], sh[e]) > 0.f) ? fd[e] : 0.f;
                    part[e] += d;
                    part[EPV + e] += d * ((fz[e] - mu[e]) * is[e]);
                }
            }
        }
    }
    block_reduce_to_acc<2 * EPV>(m, part, EPV, C, acc, 2);
}

// pass 2: dz = scale_g * (dy' - c1 - xhat*c2) with scale_g = gamma*invstd (= forward scale);
// also accumulates sum(dz) per channel (the bias gradient of the conv in front of the BN).
template <typename T, bool RELU, bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(PoolSrc ps, const T* __restrict__ dy, int ldd, int doff,
                                                           const T* __restrict__ z, int ldz, int zoff,
                                                           T* __restrict__ dz, int ldo, int ooff,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const double* __restrict__ sums, double count, int frozen,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           long long M, int C, double* __restrict__ dbias_acc,
                                                           float* __restrict__ dbias, unsigned* __restrict__ done) {
    constexpr int EPV = TT<T>::EPV;
    ChanMap m; m.init<T>(C);
    float part[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) part[i] = 0.f;
    // pass 1's sums (S1, S2) finalised here instead of in a launch of their own: dgamma = S2, dbeta = S1, c1 = S1/n, c2 = S2/n
    // (the arithmetic of bn_bwd_finalize_kernel).  One thread per channel divides (fp64) and the workgroup shares the two
    // coefficients through LDS -- every thread dividing for its eight channels cost 16 us per launch; workgroup 0 writes the
    // two gradients.  The LDS area is the block reduction's, which is not in use yet.
    extern __shared__ float coef[];                      // [2][C]
    for (int c = threadIdx.x; c < C; c += 256) {
        const double s1 = sums[c * 2], s2 = sums[c * 2 + 1];
        coef[c] = frozen ? 0.f : (float)(s1 / count);
        coef[C + c] = frozen ? 0.f : (float)(s2 / count);
        if (blockIdx.x == 0) { dgamma[c] = (float)s2; dbeta[c] = (float)s1; }
    }
    __syncthreads();
    if (m.active) {
        float sc[EPV], sh[EPV], mu[EPV], is[EPV], k1[EPV], k2[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const int c = m.cg * EPV + e;
            sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
            k1[e] = coef[c]; k2[e] = coef[C + c];
        }
        const long long stride = (long long)gridDim.x * m.PL;
        for (long long p0 = (long long)blockIdx.x * m.PL + m.pl; p0 < M; p0 += stride * STREAM_U) {
            uint4 vz[STREAM_U];
            DyRaw<T> vd[STREAM_U];
#pragma unroll
            for (int u = 0; u < STREAM_U; ++u) {
                const long long p = p0 + u * stride;
                if (p < M) {
                    vz[u] = *(const uint4*)(z + p * ldz + zoff + m.cg * EPV);
                    vd[u] = fetch_dy<T, POOL>(dy, ldd, doff, ps, p, m.cg, C);
                }
            }
#pragma unroll
            for (int u = 0; u < STREAM_U; ++u) {
                const long long p = p0 + u * stride;
                if (p >= M) break;
                float fz[EPV], fd[EPV];
                Vec16<T>::unpack(vz[u], fz);
                finish_dy<T, POOL>(vd[u], fd);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float d = (!RELU || fmaf(fz[e], sc[e], sh[e]) > 0.f) ? fd[e] : 0.f;
                    const float xh = (fz[e] - mu[e]) * is[e];
                    fd[e] = sc[e] * (d - k1[e] - xh * k2[e]);
                }
                const uint4 o = Vec16<T>::pack(fd);
                *(uint4*)(dz + p * ldo + ooff + m.cg * EPV) = o;
                if (dbias_acc != nullptr) {
                    float fr[EPV];
                    Vec16<T>::unpack(o, fr);          // sum what the consumers of dz will actually read
#pragma unroll
                    for (int e = 0; e < EPV; ++e) part[e] += fr[e];
                }
            }
        }
    }
    if (dbias_acc != nullptr) {
        __syncthreads();                                 // everyone has read its coefficients out of the shared area
        block_reduce_to_acc<EPV, true>(m, part, EPV, C, dbias_acc, 1);
        // The LAST workgroup to get here converts the finished fp64 sums to the fp32 gradient (a launch of its own before:
        // 17 per training step).  Only ATOMICS carry the data (device-scope adds above, device-scope loads below), so no cache
        // has to be written back; what is needed is that every add has been PERFORMED before thread 0 takes the ticket: the
        // adds are returning atomics whose results the threads wait for, then the barrier.  Measured alternatives: an
        // agent-scope fence here writes the XCD's whole L2 back -- dz was just streamed through it -- once per workgroup
        // (+3 ms per training step); a workgroup-scope fence after no-return adds is not enough (the last workgroup read
        // sums that other workgroups' queued adds had not reached yet: tests/test_gpu_realwidth.py caught it).
        __syncthreads();
        __shared__ unsigned ticket;
        if (threadIdx.x == 0) ticket = atomicAdd(done, 1u);
        __syncthreads();
        if (ticket == gridDim.x - 1) {
            for (int c = threadIdx.x; c < C; c += 256)
                dbias[c] = (float)__hip_atomic_load(&dbias_acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// plain ReLU backward (no norm): dz = dy * [y > 0]   (RepUNet: ReLU after the branch sum)
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dy, int ldd, int doff,
                                                       const T* __restrict__ y, int ldy, int yoff,
                                                       T* __restrict__ dz, int ldo, int ooff, long long M, int C) {
    constexpr int EPV = TT<T>::EPV;
    ChanMap m; m.init<T>(C);
    if (!m.active) return;
    for (long long p = (long long)blockIdx.x * m.PL + m.pl; p < M; p += (long long)gridDim.x * m.PL) {
        float fy[EPV], fd[EPV];
        Vec16<T>::unpack(*(const uint4*)(y + p * ldy + yoff + m.cg * EPV), fy);
        Vec16<T>::unpack(*(const uint4*)(dy + p * ldd + doff + m.cg * EPV), fd);
#pragma unroll
        for (int e = 0; e < EPV; ++e) fd[e] = fy[e] > 0.f ? fd[e] : 0.f;
        *(uint4*)(dz + p * ldo + ooff + m.cg * EPV) = Vec16<T>::pack(fd);
    }
}

// per-channel sum over pixels -> fp64 acc[C] (bias gradient of ConvTranspose2d and friends)
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ x, int ldx, int xoff, long long M, int C,
                                                          double* __restrict__ acc) {
    constexpr int EPV = TT<T>::EPV;
    ChanMap m; m.init<T>(C);
    float part[EPV];
#pragma unroll
    for (int i = 0; i < EPV; ++i) part[i] = 0.f;
    if (m.active) {
        for (long long p = (long long)blockIdx.x * m.PL + m.pl; p < M; p += (long long)gridDim.x * m.PL) {
            float f[EPV];
            Vec16<T>::unpack(*(const uint4*)(x + p * ldx + xoff + m.cg * EPV), f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) part[e] += f[e];
        }
    }
    block_reduce_to_acc<EPV>(m, part, EPV, C, acc, 1);
}

// ---------------------------------------------------------------------------------------------
// 2x2 / stride-2 max-pool with argmax.  code = 2*dh + dw of the winner; ATen CPU rule
// (max_pool2d_with_indices): scan the window row-major from -inf, take v if (v > max) || isnan(v).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, int xoff,
                                                          T* __restrict__ y, uint8_t* __restrict__ code,
                                                          int N, int H, int W, int C) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV, Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)N * Ho * Wo * G;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const int cg = (int)(v % G);
        long long p = v / G;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const T* base = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * ldx + xoff + cg * EPV;
        float best[EPV];
        unsigned cd[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { best[e] = -INFINITY; cd[e] = 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[EPV];
            Vec16<T>::unpack(*(const uint4*)(base + ((long long)(k >> 1) * W + (k & 1)) * ldx), f);
#pragma unroll
            for (int e = 0; e < EPV; ++e)
                if (f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; cd[e] = k; }
        }
        const long long o = (((long long)n * Ho + ho) * Wo + wo) * C + cg * EPV;
        *(uint4*)(y + o) = Vec16<T>::pack(best);
        if (EPV == 8) {
            uint2 pk;
            pk.x = cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24);
            pk.y = cd[4 % EPV] | (cd[5 % EPV] << 8) | (cd[6 % EPV] << 16) | (cd[7 % EPV] << 24);
            *(uint2*)(code + o) = pk;
        } else {
            *(uint32_t*)(code + o) = cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24);
        }
    }
}

// y = relu(z*scale + shift) AND its 2x2 max-pool in one pass (encoder levels, models/unet.py:45-47): the pool
// compares the values as stored (rounded to T), with the rule of maxpool_fwd_kernel, so y, pooled and code are
// exactly what affine_act followed by maxpool2x2_fwd produce; y is not read back.
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void affine_act_pool_kernel(const T* __restrict__ z, int ldz, int zoff,
                                                              T* __restrict__ y, int ldy, int yoff,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              T* __restrict__ pooled, uint8_t* __restrict__ code,
                                                              int N, int H, int W, int C) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV, Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)N * Ho * Wo * G;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const int cg = (int)(v % G);
        long long p = v / G;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float sc[EPV], sh[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { sc[e] = scale[cg * EPV + e]; sh[e] = shift[cg * EPV + e]; }
        const long long pix0 = ((long long)n * H + 2 * ho) * W + 2 * wo;
        float best[EPV];
        unsigned cd[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) { best[e] = -INFINITY; cd[e] = 0; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long pix = pix0 + (long long)(k >> 1) * W + (k & 1);
            float f[EPV];
            Vec16<T>::unpack(*(const uint4*)(z + pix * ldz + zoff + cg * EPV), f);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                f[e] = fmaf(f[e], sc[e], sh[e]);
                if (RELU) f[e] = fmaxf(f[e], 0.f);
            }
            const uint4 o = Vec16<T>::pack(f);
            *(uint4*)(y + pix * ldy + yoff + cg * EPV) = o;
            Vec16<T>::unpack(o, f);                       // the pool sees what was stored
#pragma unroll
            for (int e = 0; e < EPV; ++e)
                if (f[e] > best[e] || f[e] != f[e]) { best[e] = f[e]; cd[e] = k; }
        }
        const long long o = (((long long)n * Ho + ho) * Wo + wo) * C + cg * EPV;
        *(uint4*)(pooled + o) = Vec16<T>::pack(best);
        if (EPV == 8) {
            uint2 pk;
            pk.x = cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24);
            pk.y = cd[4 % EPV] | (cd[5 % EPV] << 8) | (cd[6 % EPV] << 16) | (cd[7 % EPV] << 24);
            *(uint2*)(code + o) = pk;
        } else {
            *(uint32_t*)(code + o) = cd[0] | (cd[1] << 8) | (cd[2] << 16) | (cd[3] << 24);
        }
    }
}

// dx[n,2ho+dh,2wo+dw,c] = (skip ? dskip[...] : 0) + (code == 2dh+dw ? dy[n,ho,wo,c] : 0)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ code,
                                                          const T* __restrict__ dskip, int lds_, int soff,
                                                          T* __restrict__ dx, int N, int H, int W, int C) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV, Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)N * Ho * Wo * G;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const int cg = (int)(v % G);
        long long p = v / G;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const long long o = (((long long)n * Ho + ho) * Wo + wo) * C + cg * EPV;
        float g[EPV];
        Vec16<T>::unpack(*(const uint4*)(dy + o), g);
        unsigned cd[8];
        if (EPV == 8) {
            const uint2 pk = *(const uint2*)(code + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { cd[e] = (pk.x >> (8 * e)) & 0xff; cd[4 + e] = (pk.y >> (8 * e)) & 0xff; }
        } else {
            const uint32_t pk = *(const uint32_t*)(code + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) cd[e] = (pk >> (8 * e)) & 0xff;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long pix = ((long long)n * H + 2 * ho + (k >> 1)) * W + 2 * wo + (k & 1);
            float f[EPV];
            if (dskip != nullptr) Vec16<T>::unpack(*(const uint4*)(dskip + pix * lds_ + soff + cg * EPV), f);
            else {
#pragma unroll
                for (int e = 0; e < EPV; ++e) f[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[e] += (cd[e] == (unsigned)k) ? g[e] : 0.f;
            *(uint4*)(dx + pix * C + cg * EPV) = Vec16<T>::pack(f);
        }
    }
}

// Odd H or W (nn.MaxPool2d floors): the last row / column belongs to no window, its gradient is the skip's alone.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_edge_kernel(const T* __restrict__ dskip, int lds_, int soff,
                                                               T* __restrict__ dx, int N, int H, int W, int C) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV;
    const int nrow = (H & 1) ? W : 0, ncol = (W & 1) ? (H - (H & 1)) : 0;       // last row; last column without the corner
    const long long total = (long long)N * (nrow + ncol) * G;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long long)gridDim.x * 256) {
        const int cg = (int)(v % G);
        long long p = v / G;
        const int e = (int)(p % (nrow + ncol));
        const int n = (int)(p / (nrow + ncol));
        const int hh = e < nrow ? H - 1 : e - nrow, ww = e < nrow ? e : W - 1;
        const long long pix = ((long long)n * H + hh) * W + ww;
        uint4 g = make_uint4(0u, 0u, 0u, 0u);
        if (dskip != nullptr) g = *(const uint4*)(dskip + pix * lds_ + soff + cg * EPV);
        *(uint4*)(dx + pix * C + cg * EPV) = g;
    }
}

// ---------------------------------------------------------------------------------------------
static inline int stream_grid_cap() {
    static const int v = [] { const char* e = getenv("UNET_AMD_BN_GRID"); const int g = e ? atoi(e) : 0; return g > 0 ? g : STREAM_GRID_CAP; }();
    return v;
}
static inline int grid_for(long long M, int PL) {
    long long g = (M + PL - 1) / PL;
    if (g > stream_grid_cap()) g = stream_grid_cap();
    if (g < 1) g = 1;
    return (int)g;
}
static inline int chan_check(int dtype, int C, const char* what) {
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    if (dtype != UNET_F32 && dtype != UNET_BF16) { unet_set_error("%s: bad dtype %d", what, dtype); return -1; }
    if (C <= 0 || C % 8 != 0 || C / epv > 256) { unet_set_error("%s: C=%d must be a multiple of 8 and at most %d", what, C, 256 * epv); return -1; }
    return 0;
}

extern "C" int unet_bn_finalize(const float* stat_rows, int nrows, int C, long long count,
                                const float* gamma, const float* beta,
                                float* running_mean, float* running_var, long long* num_batches_tracked,
                                float momentum, float eps,
                                float* scale, float* shift, float* mean, float* invstd,
                                double* acc_ws /* 2*C doubles */, void* stream) {
    UNET_CHECK(nrows > 0 && C > 0 && count > 0, "bn_finalize: empty problem");
    hipStream_t s = (hipStream_t)stream;
    if (nrows <= 4096) {
        hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, s, stat_rows, nrows, C, (double)count, gamma, beta,
                           running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, mean, invstd);
        UNET_LAUNCH_CHECK();
        return 0;
    }
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * 2 * C, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    int S = (nrows + 63) / 64; if (S > 64) S = 64; if (S < 1) S = 1;
    hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3((C + 31) / 32, S), dim3(256), 0, s, stat_rows, nrows, C, acc_ws);
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc_ws, (double)count, C, gamma, beta,
                       running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, mean, invstd);
    UNET_LAUNCH_CHECK();
    return 0;
}

// out[c] = sum_r rows[r*ld + off + c*stride]: column sums of fp32 partial rows (the ConvTranspose2d bias gradient from the
// statistics rows of the split dgrad, models/unet.py:35 autograd).  4 columns x 64 row lanes per workgroup, four loads in
// flight, partials added in lane order: no atomics, no workspace, deterministic.
__global__ __launch_bounds__(256) void rows_colsum_kernel(const float* __restrict__ rows, int nrows, int ld, int off, int C, int stride,
                                                          float* __restrict__ out) {
    const int ch = threadIdx.x & 3, pl = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + ch;
    double s = 0.0;
    if (c < C) {
        const float* col = rows + off + (size_t)c * stride;
        int r = pl;
        for (; r + 192 < nrows; r += 256) {
            const float v0 = col[(size_t)r * ld], v1 = col[(size_t)(r + 64) * ld];
            const float v2 = col[(size_t)(r + 128) * ld], v3 = col[(size_t)(r + 192) * ld];
            s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
        }
        for (; r < nrows; r += 64) s += (double)col[(size_t)r * ld];
    }
    __shared__ double red[64][4];
    red[pl][ch] = s;
    __syncthreads();
    if (pl == 0 && c < C) {
        double a = 0.0;
        for (int p = 0; p < 64; ++p) a += red[p][ch];
        out[c] = (float)a;
    }
}

extern "C" int unet_rows_colsum(const float* rows, int nrows, int ld, int off, int C, int stride, float* out, void* stream) {
    UNET_CHECK(rows != nullptr && out != nullptr, "rows_colsum: null argument");
    UNET_CHECK(nrows > 0 && C > 0 && stride > 0 && off >= 0 && ld > 0 && off + (long long)(C - 1) * stride < ld,
               "rows_colsum: columns outside the row (nrows=%d ld=%d off=%d C=%d stride=%d)", nrows, ld, off, C, stride);
    hipLaunchKernelGGL(rows_colsum_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, nrows, ld, off, C, stride, out);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift,
                                   float* mean /* nullable */, float* invstd /* nullable */, void* stream) {
    UNET_CHECK(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "bn_eval_coeffs: null argument");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift, mean, invstd);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_affine_act(int dtype, int relu, const void* z, int ldz, int zoff, void* y, int ldy, int yoff,
                               const float* scale, const float* shift, long long M, int C, void* stream) {
    if (chan_check(dtype, C, "affine_act")) return -1;
    UNET_CHECK(ldz % 8 == 0 && zoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0, "affine_act: strides/offsets must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    const int grid = grid_for(M, PL);
#define LAUNCH_AA(T, R) hipLaunchKernelGGL((affine_act_kernel<T, R>), dim3(grid), dim3(256), 0, s, (const T*)z, ldz, zoff, (T*)y, ldy, yoff, scale, shift, M, C)
    if (dtype == UNET_BF16) { if (relu) LAUNCH_AA(__bf16, true); else LAUNCH_AA(__bf16, false); }
    else { if (relu) LAUNCH_AA(float, true); else LAUNCH_AA(float, false); }
#undef LAUNCH_AA
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_affine_act_pool(int dtype, int relu, const void* z, int ldz, int zoff, void* y, int ldy, int yoff,
                                    const float* scale, const float* shift, void* pooled, uint8_t* code,
                                    int N, int H, int W, int C, void* stream) {
    if (chan_check(dtype, C, "affine_act_pool")) return -1;
    UNET_CHECK(ldz % 8 == 0 && zoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0, "affine_act_pool: strides/offsets must be multiples of 8");
    UNET_CHECK(H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0 && N > 0, "affine_act_pool: H and W must be even (got %dx%d)", H, W);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / epv);
    const int grid = grid_for(total, 256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_AP(T, R) hipLaunchKernelGGL((affine_act_pool_kernel<T, R>), dim3(grid), dim3(256), 0, s, (const T*)z, ldz, zoff, (T*)y, ldy, yoff, scale, shift, (T*)pooled, code, N, H, W, C)
    if (dtype == UNET_BF16) { if (relu) LAUNCH_AP(__bf16, true); else LAUNCH_AP(__bf16, false); }
    else { if (relu) LAUNCH_AP(float, true); else LAUNCH_AP(float, false); }
#undef LAUNCH_AP
    UNET_LAUNCH_CHECK();
    return 0;
}

template <bool POOL>
static int bn_relu_bwd_impl(PoolSrc ps, int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                            void* dz, int ldo, int ooff,
                            const float* scale, const float* shift, const float* mean, const float* invstd,
                            float* dgamma, float* dbeta, float* dbias, float* c1c2_ws, double* acc_ws,
                            long long M, int C, void* stream, const float* sum_rows = nullptr, int nrows = 0) {
    if (chan_check(dtype, C, "bn_relu_bwd")) return -1;
    UNET_CHECK(scale && shift && mean && invstd, "bn_relu_bwd: scale / shift / mean / invstd must not be NULL");
    UNET_CHECK(dy && z && dz && dgamma && dbeta && acc_ws, "bn_relu_bwd: null argument");
    const int frozen = (relu >> 1) & 1, prezeroed = (relu >> 2) & 1;
    relu &= 1;
    UNET_CHECK(ldd % 8 == 0 && doff % 8 == 0 && ldz % 8 == 0 && zoff % 8 == 0 && ldo % 8 == 0 && ooff % 8 == 0,
               "bn_relu_bwd: strides/offsets must be multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    const int grid = grid_for(M, PL);
    // 3*C fp64 sums + the apply pass's completion counter; relu bit 2: the caller hands over an area that is already zero
    // (unet_amd clears ONE arena per backward pass instead of one memset per layer)
    if (!prezeroed && hipMemsetAsync(acc_ws, 0, sizeof(double) * (3 * C + 2), s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    const size_t lds1 = reduce_lds_bytes(C / epv, 2 * epv);
    size_t lds2 = reduce_lds_bytes(C / epv, epv);
    if (lds2 < (size_t)2 * C * sizeof(float)) lds2 = (size_t)2 * C * sizeof(float);      // the apply pass parks c1 / c2 there first
    (void)c1c2_ws;            // the finalize that filled it now lives in the apply pass
    double* dbacc = dbias ? acc_ws + 2 * C : nullptr;
#define LAUNCH_R(T, R) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, R, POOL>), dim3(grid), dim3(256), lds1, s, ps, (const T*)dy, ldd, doff, (const T*)z, ldz, zoff, scale, shift, mean, invstd, M, C, acc_ws)
#define LAUNCH_A(T, R) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R, POOL>), dim3(grid), dim3(256), lds2, s, ps, (const T*)dy, ldd, doff, (const T*)z, ldz, zoff, (T*)dz, ldo, ooff, scale, shift, mean, invstd, acc_ws, (double)M, frozen, dgamma, dbeta, M, C, dbacc, dbias, (unsigned*)(acc_ws + 3 * C))
    if (sum_rows != nullptr) {
        // the first pass's sums arrive as per-tile rows from the producing dgrad's epilogue (unet_conv_dgrad_bnred)
        int S = (nrows + 63) / 64; if (S > 64) S = 64; if (S < 1) S = 1;
        hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3((C + 31) / 32, S), dim3(256), 0, s, sum_rows, nrows, C, acc_ws);
    } else if (dtype == UNET_BF16) { if (relu) LAUNCH_R(__bf16, true); else LAUNCH_R(__bf16, false); }
    else { if (relu) LAUNCH_R(float, true); else LAUNCH_R(float, false); }
    UNET_LAUNCH_CHECK();
    if (dtype == UNET_BF16) { if (relu) LAUNCH_A(__bf16, true); else LAUNCH_A(__bf16, false); }
    else { if (relu) LAUNCH_A(float, true); else LAUNCH_A(float, false); }
    UNET_LAUNCH_CHECK();
#undef LAUNCH_R
#undef LAUNCH_A
    return 0;
}

extern "C" int unet_bn_relu_bwd(int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                                void* dz, int ldo, int ooff,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                float* dgamma, float* dbeta, float* dbias /* nullable */,
                                float* c1c2_ws /* unused */, double* acc_ws /* 3*C + 2 doubles */,
                                long long M, int C, void* stream) {
    return bn_relu_bwd_impl<false>(PoolSrc{nullptr, nullptr, 0, 0}, dtype, relu, dy, ldd, doff, z, ldz, zoff, dz, ldo, ooff,
                                   scale, shift, mean, invstd, dgamma, dbeta, dbias, c1c2_ws, acc_ws, M, C, stream);
}

// unet_bn_relu_bwd whose first pass has already run inside the dgrad that produced dy (unet_conv_dgrad_bnred): sum_rows
// [nrows][C][2] hold per-tile (sum dy', sum dy'*xhat); they are added up in fp64 and the second pass runs as usual.
extern "C" int unet_bn_relu_bwd_rows(int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                                     void* dz, int ldo, int ooff,
                                     const float* scale, const float* shift, const float* mean, const float* invstd,
                                     float* dgamma, float* dbeta, float* dbias /* nullable */,
                                     const float* sum_rows, int nrows, double* acc_ws /* 3*C + 2 doubles */,
                                     long long M, int C, void* stream) {
    UNET_CHECK(sum_rows != nullptr && nrows > 0, "bn_relu_bwd_rows: no rows");
    return bn_relu_bwd_impl<false>(PoolSrc{nullptr, nullptr, 0, 0}, dtype, relu, dy, ldd, doff, z, ldz, zoff, dz, ldo, ooff,
                                   scale, shift, mean, invstd, dgamma, dbeta, dbias, nullptr, acc_ws, M, C, stream, sum_rows, nrows);
}

// ---------------------------------------------------------------------------------------------
// unet_bn_relu_bwd for callers that cannot allocate (code under HIP graph capture, tight serving / fine-tuning loops,
// harnesses that forbid hipMalloc): the two workspaces of unet_bn_relu_bwd (3*C fp64 accumulators + 2*C fp32
// coefficients) live in a file-scope device array instead of caller memory.  Every GPU holds its own copy of the array
// at its own address until the process ends (NOALLOC_SCRATCH_BYTES = 64 KiB).  The array's name is a host-side stand-in,
// never a device address: the address is fetched once per device with hipGetSymbolAddress and only that pointer is
// handed to memsets and kernels.
#define NOALLOC_MAX_C 2048
#define NOALLOC_ACC_BYTES ((NOALLOC_MAX_C * 3 + 2) * sizeof(double))
#define NOALLOC_SCRATCH_BYTES (NOALLOC_ACC_BYTES + NOALLOC_MAX_C * 2 * sizeof(float))
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

#define NOALLOC_MAX_DEVICES 64
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[NOALLOC_MAX_DEVICES];      // per device: that GPU's address of `scratch` (nullptr = not asked yet)

static int noalloc_scratch(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NOALLOC_MAX_DEVICES) {
        unet_set_error("bn_relu_bwd_noalloc: cannot identify the current device (%d)", dev);
        return -2;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));     // with `dev` current: dev's copy
        if (e != hipSuccess || ptr == nullptr) {
            unet_set_error("bn_relu_bwd_noalloc: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return -2;
        }
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int unet_bn_relu_bwd_noalloc(int dtype, int relu, const void* dy, int ldd, int doff, const void* z, int ldz, int zoff,
                                        void* dz, int ldo, int ooff,
                                        const float* scale, const float* shift, const float* mean, const float* invstd,
                                        float* dgamma, float* dbeta, float* dbias /* nullable */,
                                        long long M, int C, void* stream) {
    UNET_CHECK(C > 0 && C <= NOALLOC_MAX_C, "bn_relu_bwd_noalloc: C=%d exceeds the built-in scratch (at most %d channels); "
               "use unet_bn_relu_bwd with caller workspaces", C, NOALLOC_MAX_C);
    void* ptr = nullptr;
    if (int rc = noalloc_scratch(&ptr)) return rc;
    relu &= 3;                                                                 // the scratch keeps the last call's contents: never "already zero"
    double* acc_ws = (double*)ptr;                                             // 3*C + 2 doubles, cleared by the shared implementation
    float* c1c2_ws = (float*)((unsigned char*)ptr + NOALLOC_ACC_BYTES);        // 2*C floats, fully rewritten before they are read
    return bn_relu_bwd_impl<false>(PoolSrc{nullptr, nullptr, 0, 0}, dtype, relu, dy, ldd, doff, z, ldz, zoff, dz, ldo, ooff,
                                   scale, shift, mean, invstd, dgamma, dbeta, dbias, c1c2_ws, acc_ws, M, C, stream);
}

extern "C" int unet_bn_relu_bwd_pool(int dtype, int relu, const void* dpool, const uint8_t* code, const void* dskip, int lds, int soff,
                                     const void* z, int ldz, int zoff, void* dz, int ldo, int ooff,
                                     const float* scale, const float* shift, const float* mean, const float* invstd,
                                     float* dgamma, float* dbeta, float* dbias /* nullable */,
                                     float* c1c2_ws /* unused */, double* acc_ws /* 3*C + 2 doubles */,
                                     int N, int H, int W, int C, void* stream) {
    UNET_CHECK(dpool && code && dskip, "bn_relu_bwd_pool: null argument");
    UNET_CHECK(N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "bn_relu_bwd_pool: H and W must be even (got %dx%d)", H, W);
    UNET_CHECK((long long)N * H * W < 0x7fffffffLL, "bn_relu_bwd_pool: too many pixels");
    return bn_relu_bwd_impl<true>(PoolSrc{dpool, code, H, W}, dtype, relu, dskip, lds, soff, z, ldz, zoff, dz, ldo, ooff,
                                  scale, shift, mean, invstd, dgamma, dbeta, dbias, c1c2_ws, acc_ws, (long long)N * H * W, C, stream);
}

extern "C" int unet_relu_bwd(int dtype, const void* dy, int ldd, int doff, const void* y, int ldy, int yoff,
                             void* dz, int ldo, int ooff, long long M, int C, void* stream) {
    if (chan_check(dtype, C, "relu_bwd")) return -1;
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    const int grid = grid_for(M, PL);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16)
        hipLaunchKernelGGL((relu_bwd_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)dy, ldd, doff, (const __bf16*)y, ldy, yoff, (__bf16*)dz, ldo, ooff, M, C);
    else
        hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dy, ldd, doff, (const float*)y, ldy, yoff, (float*)dz, ldo, ooff, M, C);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_channel_sum(int dtype, const void* x, int ldx, int xoff, long long M, int C, float* out, int accumulate,
                                double* acc_ws /* C doubles */, void* stream) {
    if (chan_check(dtype, C, "channel_sum")) return -1;
    hipStream_t s = (hipStream_t)stream;
    const int epv = dtype == UNET_BF16 ? 8 : 4, PL = 256 / (C / epv);
    const int grid = grid_for(M, PL);
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * C, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    const size_t lds = reduce_lds_bytes(C / epv, epv);
    if (dtype == UNET_BF16)
        hipLaunchKernelGGL((channel_sum_kernel<__bf16>), dim3(grid), dim3(256), lds, s, (const __bf16*)x, ldx, xoff, M, C, acc_ws);
    else
        hipLaunchKernelGGL((channel_sum_kernel<float>), dim3(grid), dim3(256), lds, s, (const float*)x, ldx, xoff, M, C, acc_ws);
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(acc_to_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc_ws, out, C, accumulate);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_maxpool2x2_fwd(int dtype, const void* x, int ldx, int xoff, void* y, uint8_t* code,
                                   int N, int H, int W, int C, void* stream) {
    if (chan_check(dtype, C, "maxpool_fwd")) return -1;
    UNET_CHECK(H > 1 && W > 1, "maxpool: H and W must be at least 2 (got %dx%d)", H, W);      // odd sizes: floor, as nn.MaxPool2d(2, 2)
    UNET_CHECK(ldx % 8 == 0 && xoff % 8 == 0, "maxpool: stride/offset must be multiples of 8");
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / epv);
    const int grid = grid_for(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16)
        hipLaunchKernelGGL((maxpool_fwd_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)x, ldx, xoff, (__bf16*)y, code, N, H, W, C);
    else
        hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, ldx, xoff, (float*)y, code, N, H, W, C);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_maxpool2x2_bwd(int dtype, const void* dy, const uint8_t* code, const void* dskip, int lds_, int soff,
                                   void* dx, int N, int H, int W, int C, void* stream) {
    if (chan_check(dtype, C, "maxpool_bwd")) return -1;
    UNET_CHECK(H > 1 && W > 1, "maxpool: H and W must be at least 2 (got %dx%d)", H, W);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / epv);
    const int grid = grid_for(total, 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UNET_BF16)
        hipLaunchKernelGGL((maxpool_bwd_kernel<__bf16>), dim3(grid), dim3(256), 0, s, (const __bf16*)dy, code, (const __bf16*)dskip, lds_, soff, (__bf16*)dx, N, H, W, C);
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)dy, code, (const float*)dskip, lds_, soff, (float*)dx, N, H, W, C);
    UNET_LAUNCH_CHECK();
    if ((H | W) & 1) {
        const int nedge = ((H & 1) ? W : 0) + ((W & 1) ? (H - (H & 1)) : 0);
        const int ge = grid_for((long long)N * nedge * (C / epv), 256);
        if (dtype == UNET_BF16)
            hipLaunchKernelGGL((maxpool_bwd_edge_kernel<__bf16>), dim3(ge), dim3(256), 0, s, (const __bf16*)dskip, lds_, soff, (__bf16*)dx, N, H, W, C);
        else
            hipLaunchKernelGGL((maxpool_bwd_edge_kernel<float>), dim3(ge), dim3(256), 0, s, (const float*)dskip, lds_, soff, (float*)dx, N, H, W, C);
        UNET_LAUNCH_CHECK();
    }
    return 0;
}
