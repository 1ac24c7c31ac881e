// Shared by the bandwidth-bound NHWC kernels (norm_pool.hip, edge_fused.hip, conv_edge.hip): the per-channel thread
// mapping, the block reduction into fp64 accumulators, the BN-backward finalize, and the Cin = 1 sliding 3x3 window.
#pragma once
#include "common.h"

#define MAX_GRID 2048
// Workgroups of a streaming (HBM-bound, grid-stride) kernel: 4 per CU.  More resident workgroups LOWER the achieved HBM rate
// on this part (8 per CU: 4.97 TB/s, 3-4 per CU: 5.25-5.44 on a 2.1 GB read+write pass); UNET_AMD_BN_GRID overrides (experiments).
#define STREAM_GRID_CAP 1024

struct ChanMap {
    int G, PL, pl, cg;
    bool active;
    template <typename T> __device__ __forceinline__ void init(int C) {
        G = C / TT<T>::EPV;
        PL = 256 / G;
        pl = threadIdx.x / G;
        cg = threadIdx.x - pl * G;
        active = pl < PL;
    }
};

// Block-level reduction of per-thread [NV] partials over the PL pixel lanes, then fp64 atomics.
// Power-of-two channel-group counts (every U-Net width) first reduce inside the wave with shuffles and need only
// 4 x min(G,64) x NV floats of LDS (2 KB at C=64): small enough for these HBM-bound kernels to stay resident next to
// a weight-gradient kernel that holds 151 KB of the CU's LDS (engine.wgrad_async).  reduce_lds_bytes() is the size.
static inline size_t reduce_lds_bytes(int G, int NV) {
    const bool pow2 = (G & (G - 1)) == 0;
    return pow2 ? (size_t)4 * (G < 64 ? G : 64) * NV * sizeof(float) : (size_t)256 * NV * sizeof(float);
}

// WAIT = true: the adds are RETURNING atomics and the thread holds on to their results, i.e. it does not leave this function
// before every one of its adds has been performed at the device's coherence point (a no-return atomic is only queued).  For
// callers whose workgroup afterwards takes a completion ticket that lets another workgroup read the sums.
template <int NV, bool WAIT = false>
__device__ __forceinline__ void block_reduce_to_acc(const ChanMap& m, const float (&part)[NV], int epv, int C,
                                                    double* acc, int acc_stride /* values per channel */) {
    extern __shared__ float red_lds[];
    const int G = m.G;
    if ((G & (G - 1)) == 0) {
        // thread t holds channel group t % G; inside a wave equal groups sit at lane strides G, 2G, ...
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = part[i];
        for (int st = G; st < 64; st <<= 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], st);
        }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, SL = G < 64 ? G : 64;
        if (lane < SL) {
#pragma unroll
            for (int i = 0; i < NV; ++i) red_lds[(w * SL + lane) * NV + i] = v[i];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < G * NV; i += 256) {
            const int cg = i / NV, r = i - cg * NV;
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const int off = (cg - ww * 64) & (G - 1);          // lane of wave ww that held group cg, if any
                if (off < 64) sum += red_lds[(ww * SL + (off & (SL - 1))) * NV + r];
            }
            const int k = r / epv, e = r - k * epv;
            const int c = cg * epv + e;
            if (c < C) {
                if constexpr (WAIT) { const double old = atomicAdd(&acc[c * acc_stride + k], (double)sum); asm volatile("" :: "v"(old)); }
                else atomicAdd(&acc[c * acc_stride + k], (double)sum);
            }
        }
        return;
    }
    // general G: layout in LDS [pl][cg][NV]
    if (m.active) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red_lds[(m.pl * m.G + m.cg) * NV + i] = part[i];
    }
    __syncthreads();
    const int per_block = m.G * NV;                  // values after reducing over pl
    for (int i = threadIdx.x; i < per_block; i += 256) {
        float s = 0.f;
        for (int p = 0; p < m.PL; ++p) s += red_lds[p * per_block + i];
        // i = cg*NV + (k*epv + e): value k of channel cg*epv+e
        const int cg = i / NV, r = i - cg * NV;
        const int k = r / epv, e = r - k * epv;
        const int c = cg * epv + e;
        if (c < C) {
            if constexpr (WAIT) { const double old = atomicAdd(&acc[c * acc_stride + k], (double)s); asm volatile("" :: "v"(old)); }
            else atomicAdd(&acc[c * acc_stride + k], (double)s);
        }
    }
}

// pass 1 finalize: dgamma = S2, dbeta = S1, c1 = S1/n, c2 = S2/n.  frozen (eval-mode BN: mean / invstd are the
// running statistics, constants of the batch): the statistics carry no gradient, so c1 = c2 = 0 and pass 2 reduces to
// dz = scale * dy'.
static __global__ __attribute__((unused)) void bn_bwd_finalize_kernel(const double* __restrict__ acc, double count, int C, int frozen,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ c1, float* __restrict__ c2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = acc[c * 2], s2 = acc[c * 2 + 1];
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    c1[c] = frozen ? 0.f : (float)(s1 / count);
    c2[c] = frozen ? 0.f : (float)(s2 / count);
}

static __global__ __attribute__((unused)) void acc_to_f32_kernel(const double* __restrict__ acc, float* __restrict__ out, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = accumulate ? out[i] + (float)acc[i] : (float)acc[i];
}


// Cin == 1, 3x3: the three inputs of row hh around column wq (zero padding), for the sliding-window kernels that walk
// DOWN a column strip of FIRST_R rows (a pixel costs 3 input loads instead of 9).
constexpr int FIRST_R = 16;
__device__ __forceinline__ void first_row3(const float* __restrict__ xi, int hh, int wq, int H, int W, float* r) {
    const bool rowok = (unsigned)hh < (unsigned)H;
    const float* xr = xi + (long long)hh * W;
    r[0] = (rowok && wq >= 1) ? xr[wq - 1] : 0.f;
    r[1] = rowok ? xr[wq] : 0.f;
    r[2] = (rowok && wq + 1 < W) ? xr[wq + 1] : 0.f;
}

// storage rounding of the compute dtype (what a consumer of a stored T tensor would read back)
template <typename T> __device__ __forceinline__ float round_to(float v) { return to_f32(from_f32<T>(v)); }
