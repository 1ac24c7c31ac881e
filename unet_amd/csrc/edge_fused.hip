// The two ENDS of the U-Net fused into their BatchNorm passes (gfx950).  Both layers have next to no arithmetic -- the
// first conv is 9 MACs per output (in_channels = 1), the final conv 2 MACs per input -- so their tensors are cheaper to
// recompute inside the neighbouring bandwidth-bound pass than to store and read back:
//
//   first DoubleConv conv (models/unet.py:30 with in_channels = 1, then BatchNorm2d + ReLU, :11-12)
//     forward : statistics-only conv pass (unet_conv_first_fwd with z = NULL), then ONE pass x -> y = relu(bn(conv(x)));
//               the pre-norm tensor z0 (1.07 GB at 32 x 512^2 x 64 bf16) is never written or read;
//     backward: BN backward with z0 recomputed from x in both passes, and the conv's weight gradient accumulated in
//               the second pass from the dz it has in registers: dz0 is never written either (the first conv has no
//               input gradient to produce).  6.4 GB -> 2.2 GB of traffic for this layer's backward.
//   last DoubleConv BatchNorm + ReLU followed by final_conv 1x1 (models/unet.py:14-15,39,62)
//     forward : ONE pass z -> logits = W relu(bn(z)) + b; the last activation y is never written or read;
//     backward: BN backward whose incoming gradient dy = W^T dlogits is recomputed per element from the 2-8 logit
//               gradients of the pixel (8 B/pixel instead of 128), with final_conv's dW / db accumulated in the first
//               pass from the recomputed y: neither y nor dy exists in memory.  14 -> 6.3 B/element.
//
// Numerics are those of the unfused kernel sequence: every tensor that sequence stores in the compute dtype T (z0, y,
// dy, dz0) is rounded to T here at the same point (round_to<T>), so the two paths agree to the last bit except for the
// order of fp64 atomic adds.
#include "bn_shared.h"
#include "lds_dma.h"
#include <cstdlib>

// =================================================================================================
// first layer
// =================================================================================================

// y = act(scale * round_T(conv3x3(x) + bias) + shift), Cin = 1; same column-strip walk as conv_first1_fwd_kernel
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void first1_bn_act_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            T* __restrict__ y, int ldy, int yoff, int H, int W, int Cout) {
    constexpr int EPV = TT<T>::EPV;
    const int G = Cout / EPV, PL = 256 / G;
    const int pl = threadIdx.x / G, cg = threadIdx.x - pl * G;
    if (pl >= PL) return;
    float wr[9][EPV], b[EPV], sc[EPV], sh[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        const int c = cg * EPV + e;
        b[e] = bias != nullptr ? bias[c] : 0.f;
        sc[e] = scale[c]; sh[e] = shift[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][e] = w[c * 9 + t];
    }
    const int tiles_w = (W + PL - 1) / PL, tiles_h = (H + FIRST_R - 1) / FIRST_R;
    const float* xi = x + (long long)blockIdx.y * H * W;
    T* yi = y + (long long)blockIdx.y * H * W * ldy + yoff;
    for (int tile = blockIdx.x; tile < tiles_w * tiles_h; tile += gridDim.x) {
        const int th = tile / tiles_w, tw = tile - th * tiles_w;
        const int h0 = th * FIRST_R, wq = tw * PL + pl;
        if (wq >= W) continue;
        const int h1 = min(h0 + FIRST_R, H);
        float ra[3], rb[3], rc[3], rn[3];
        first_row3(xi, h0 - 1, wq, H, W, ra);
        first_row3(xi, h0, wq, H, W, rb);
        first_row3(xi, h0 + 1, wq, H, W, rc);
        for (int h = h0; h < h1; ++h) {
            first_row3(xi, h + 2, wq, H, W, rn);
            float o[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float v = b[e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    v = fmaf(ra[k], wr[k][e], v);
                    v = fmaf(rb[k], wr[3 + k][e], v);
                    v = fmaf(rc[k], wr[6 + k][e], v);
                }
                v = fmaf(round_to<T>(v), sc[e], sh[e]);
                o[e] = RELU ? fmaxf(v, 0.f) : v;
            }
            *(uint4*)(yi + ((long long)h * W + wq) * ldy + cg * EPV) = Vec16<T>::pack(o);
#pragma unroll
            for (int k = 0; k < 3; ++k) { ra[k] = rb[k]; rb[k] = rc[k]; rc[k] = rn[k]; }
        }
    }
}

// BN+ReLU backward of the first layer with z0 = round_T(conv(x)) recomputed.
//   PASS 1: S1 = sum dy', S2 = sum dy' * xhat                      -> acc[C][2]
//   PASS 2: dz = round_T(scale * (dy' - c1 - xhat*c2)); sum dz     -> dbias acc[C]
//           dW[co][tap] = sum_p dz[p][co] * x[p + tap]             -> wacc[C][9]   (dz is not stored)
//   PASS 3: pass 1 plus A[co][tap] = sum_p dy'[p][co] * x[p + tap] -> wacc[C][9]: everything the SINGLE-PASS form needs
//           from the gradient (unet_first_bn_relu_bwd below); passes 1 + 2 remain as the reference form for the tests.
template <typename T, bool RELU, int PASS>
__global__ __launch_bounds__(256) void first1_bn_bwd_kernel(const T* __restrict__ dy, int ldd, int doff,
                                                            const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ c1, const float* __restrict__ c2,
                                                            int N, int H, int W, int Cout,
                                                            double* __restrict__ acc, double* __restrict__ wacc) {
    constexpr int EPV = TT<T>::EPV;
    ChanMap m; m.init<T>(Cout);
    float part[2 * EPV];
#pragma unroll
    for (int i = 0; i < 2 * EPV; ++i) part[i] = 0.f;
    float pw[PASS >= 2 ? 9 : 1][EPV];
#pragma unroll
    for (int t = 0; t < (PASS >= 2 ? 9 : 1); ++t)
#pragma unroll
        for (int e = 0; e < EPV; ++e) pw[t][e] = 0.f;
    if (m.active) {
        float wr[9][EPV], b[EPV], sc[EPV], sh[EPV], mu[EPV], is[EPV], k1[EPV], k2[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const int c = m.cg * EPV + e;
            b[e] = bias != nullptr ? bias[c] : 0.f;
            sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
            k1[e] = PASS == 2 ? c1[c] : 0.f; k2[e] = PASS == 2 ? c2[c] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) wr[t][e] = w[c * 9 + t];
        }
        const int tiles_w = (W + m.PL - 1) / m.PL, tiles_h = (H + FIRST_R - 1) / FIRST_R;
        const int per_img = tiles_w * tiles_h;
        for (long long tile = blockIdx.x; tile < (long long)per_img * N; tile += gridDim.x) {
            const int n = (int)(tile / per_img), tt = (int)(tile - (long long)n * per_img);
            const int th = tt / tiles_w, tw = tt - th * tiles_w;
            const int h0 = th * FIRST_R, wq = tw * m.PL + m.pl;
            if (wq >= W) continue;
            const int h1 = min(h0 + FIRST_R, H);
            const float* xi = x + (long long)n * H * W;
            const T* di = dy + (long long)n * H * W * ldd + doff + m.cg * EPV;
            float ra[3], rb[3], rc[3], rn[3];
            first_row3(xi, h0 - 1, wq, H, W, ra);
            first_row3(xi, h0, wq, H, W, rb);
            first_row3(xi, h0 + 1, wq, H, W, rc);
            uint4 gq = *(const uint4*)(di + ((long long)h0 * W + wq) * ldd);
            for (int h = h0; h < h1; ++h) {
                first_row3(xi, h + 2, wq, H, W, rn);          // next row's inputs and gradient in flight under the FMAs
                const uint4 gn = h + 1 < h1 ? *(const uint4*)(di + ((long long)(h + 1) * W + wq) * ldd) : gq;
                float fd[EPV];
                Vec16<T>::unpack(gq, fd);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float v = b[e];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        v = fmaf(ra[k], wr[k][e], v);
                        v = fmaf(rb[k], wr[3 + k][e], v);
                        v = fmaf(rc[k], wr[6 + k][e], v);
                    }
                    const float z = round_to<T>(v);
                    const float d = (!RELU || fmaf(z, sc[e], sh[e]) > 0.f) ? fd[e] : 0.f;
                    const float xh = (z - mu[e]) * is[e];
                    if constexpr (PASS == 1 || PASS == 3) {
                        part[e] += d;
                        part[EPV + e] += d * xh;
                        if constexpr (PASS == 3) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                pw[k][e] = fmaf(d, ra[k], pw[k][e]);
                                pw[3 + k][e] = fmaf(d, rb[k], pw[3 + k][e]);
                                pw[6 + k][e] = fmaf(d, rc[k], pw[6 + k][e]);
                            }
                        }
                    } else {
                        const float dz = round_to<T>(sc[e] * (d - k1[e] - xh * k2[e]));
                        part[e] += dz;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            pw[k][e] = fmaf(dz, ra[k], pw[k][e]);
                            pw[3 + k][e] = fmaf(dz, rb[k], pw[3 + k][e]);
                            pw[6 + k][e] = fmaf(dz, rc[k], pw[6 + k][e]);
                        }
                    }
                }
                gq = gn;
#pragma unroll
                for (int k = 0; k < 3; ++k) { ra[k] = rb[k]; rb[k] = rc[k]; rc[k] = rn[k]; }
            }
        }
    }
    if constexpr (PASS == 1 || PASS == 3) block_reduce_to_acc<2 * EPV>(m, part, EPV, Cout, acc, 2);
    if constexpr (PASS == 2) {
        float p1[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) p1[e] = part[e];
        block_reduce_to_acc<EPV>(m, p1, EPV, Cout, acc, 1);
    }
    if constexpr (PASS >= 2) {
        // weight-gradient partials, one tap at a time through the same reduction (acc row stride 9: wacc[c*9 + t])
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            __syncthreads();
            float pt[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) pt[e] = pw[t][e];
            block_reduce_to_acc<EPV>(m, pt, EPV, Cout, wacc + t, 9);
        }
    }
}

// One dword per lane to LDS through a buffer descriptor (lane-linear, 256 bytes per wave instruction); a lane whose voff is
// beyond num_records writes ZERO -- the conv's zero padding for the 1-channel image.
__device__ __forceinline__ void lds_dma4_buf(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

// The single-pass backward (PASS 3 of first1_bn_bwd_kernel: S1, S2 and A[co][tap] = sum dy' * x[p + tap]) with its operands
// streamed through a wave-private LDS RING by LDS-DMA.
//
// Why: the arithmetic needs ~240 registers (72 weights + 72 weight-gradient partials per lane), i.e. two waves per SIMD, and a
// register prefetch of the gradient one row ahead left ONE 1-KiB request per wave in flight: a row took one HBM latency
// (~2 500 cycles against ~1 000 of arithmetic for a SIMD's two waves) and the pass ran at 1.7 TB/s.  The image rows cannot
// simply be prefetched deeper either -- vmcnt returns in order, so a shallow x load queues behind every older gradient load.
// Here nothing on the loop's path is a register load: each step's gradient row (64 lanes x 16 B) and image row (the wave's
// pixel columns + 2, one dword per lane, borders zero-filled by the descriptor) are DMA'd FR_D steps ahead into the wave's
// ring, counted with vmcnt by hand, and read back with ds_read when their step comes.  No workgroup barrier in the loop: every
// wave owns its columns and its ring.
//
// Stream: a tile is FR_ROWS image rows of the workgroup's PL columns; it takes rows + 2 steps, step j carrying image row
// h0 - 1 + j and gradient row h0 + j - 2 (steps 0 and 1 re-request row h0: 3 % of the gradient traffic), so that the 3-row
// window is complete when the gradient row arrives and a tile boundary is just another step.
#ifndef FR_DEPTH
#define FR_DEPTH 8
#endif
constexpr int FR_D = FR_DEPTH;                 // ring depth (steps in flight per wave)
constexpr int FR_SLOT = 1024 + 256;            // bytes per step and wave: gradient row + image row
constexpr int FR_ROWS = 64;                    // image rows per tile
static inline size_t first_ring_lds_bytes() { return (size_t)4 * FR_D * FR_SLOT; }

template <typename T, bool RELU>
__global__ __launch_bounds__(256) void first1_bn_bwd_ring_kernel(const T* __restrict__ dy, int ldd, int doff,
                                                                 const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 int N, int H, int W, int Cout, unsigned xbytes,
                                                                 double* __restrict__ acc, double* __restrict__ wacc) {
    constexpr int EPV = TT<T>::EPV;
    extern __shared__ float red_lds[];
    ChanMap m; m.init<T>(Cout);
    float part[2 * EPV];
#pragma unroll
    for (int i = 0; i < 2 * EPV; ++i) part[i] = 0.f;
    float pw[9][EPV];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < EPV; ++e) pw[t][e] = 0.f;
    float wr[9][EPV], b[EPV], sc[EPV], sh[EPV], mu[EPV], is[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        const int c = (m.active ? m.cg : 0) * EPV + e;
        b[e] = bias != nullptr ? bias[c] : 0.f;
        sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][e] = w[c * 9 + t];
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pb = (wave * 64) / m.G;                              // first pixel column (of the tile's PL) this wave owns
    const int npx = min((wave * 64 + 63) / m.G, m.PL - 1) - pb + 1; // and how many: the image row needs npx + 2 dwords
    const int mp = m.pl - pb;                                      // this lane's pixel inside the wave's run
    const int tiles_w = (W + m.PL - 1) / m.PL, tiles_h = (H + FR_ROWS - 1) / FR_ROWS;
    const int per_img = tiles_w * tiles_h;
    const int total = per_img * N;
    const i32x4_t xrsrc = make_rsrc(x, xbytes);
    const unsigned ring0 = lds_addr(red_lds) + (unsigned)wave * (FR_D * FR_SLOT);
    const char* ringp = (const char*)red_lds + wave * (FR_D * FR_SLOT);
    const long long rowbytes = (long long)W * ldd * (long long)sizeof(T);

    // ---- request cursor: the (tile, step) sequence below, FR_D steps ahead.  Plain locals and macros on purpose: state captured
    // by a lambda stays in scratch memory next to the asm statements' "memory" clobbers, and scratch loads are vmcnt traffic ----
    int ptile = blockIdx.x, pj = 0, pslots = 0, phh = 0;
    const char* pdy = (const char*)(dy + doff);                    // always a valid address (dummy requests once the stream is dry)
    unsigned pxoff = 0;
    bool pxcol = false;
// timing-only experiment builds (results wrong by construction): -DFR_X_NODMA drops both requests, -DFR_X_NOXDMA the image
// row's, -DFR_X_NOCOMP the arithmetic
#if defined(FR_X_NODMA)
#define FR_DMA_DY(a, b) asm volatile("" :: "v"(a), "s"(b))
#define FR_DMA_X(r, o, d) asm volatile("" :: "s"(r), "v"(o), "s"(d))
#define FR_VMCNT 0
#elif defined(FR_X_NOXDMA)
#define FR_DMA_DY(a, b) lds_dma16(a, b)
#define FR_DMA_X(r, o, d) asm volatile("" :: "s"(r), "v"(o), "s"(d))
#define FR_VMCNT (FR_D - 1)
#else
#define FR_DMA_DY(a, b) lds_dma16(a, b)
#define FR_DMA_X(r, o, d) lds_dma4_buf(r, o, d)
#define FR_VMCNT (2 * (FR_D - 1))
#endif
#define FR_OPEN()                                                                                                   \
    do {                                                                                                            \
        if (ptile >= total) { pslots = 0; break; }                                                                  \
        const int n_ = ptile / per_img, tt_ = ptile - n_ * per_img;                                                 \
        const int th_ = tt_ / tiles_w, tw_ = tt_ - th_ * tiles_w;                                                   \
        const int h0_ = th_ * FR_ROWS, w0_ = tw_ * m.PL;                                                            \
        pslots = min(h0_ + FR_ROWS, H) - h0_ + 2;                                                                   \
        pj = 0;                                                                                                     \
        phh = h0_ - 1;                                                                                              \
        const int wq_ = min(w0_ + (m.active ? m.pl : 0), W - 1);   /* lanes without a column request a valid one */ \
        pdy = (const char*)(dy + (((long long)n_ * H + h0_) * W + wq_) * ldd + doff + (m.active ? m.cg : 0) * EPV); \
        const int col_ = w0_ + pb - 1 + lane;                                                                       \
        pxcol = lane < npx + 2 && col_ >= 0 && col_ < W;                                                            \
        pxoff = (unsigned)((((long long)n_ * H + (h0_ - 1)) * W + col_) * 4);                                       \
    } while (0)
#define FR_REQUEST(slot_)                                                                                           \
    do {                                                                                                            \
        const unsigned dst_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring0 + (slot_) * FR_SLOT));           \
        FR_DMA_DY(pdy, dst_);                                                                                       \
        FR_DMA_X(xrsrc, (pxcol && (unsigned)phh < (unsigned)H) ? pxoff : LDS_DMA_OOB, dst_ + 1024);                 \
        if (pslots != 0) {                                                                                          \
            ++pj; ++phh; pxoff += (unsigned)W * 4u;                                                                 \
            if (pj > 2 && pj < pslots) pdy += rowbytes;            /* step j carries gradient row h0 + max(j - 2, 0) */ \
            if (pj == pslots) { ptile += gridDim.x; FR_OPEN(); }                                                    \
        }                                                                                                           \
    } while (0)
    FR_OPEN();
#pragma unroll 1
    for (int k = 0; k < FR_D; ++k) FR_REQUEST((unsigned)k);

    unsigned cnt = 0;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int tt = tile % per_img;
        const int th = tt / tiles_w, tw = tt - th * tiles_w;
        const int h0 = th * FR_ROWS;
        const int slots = min(h0 + FR_ROWS, H) - h0 + 2;
        const bool mine = m.active && tw * m.PL + m.pl < W;
        float ra[3] = {0.f, 0.f, 0.f}, rb[3] = {0.f, 0.f, 0.f}, rc[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < slots; ++j) {
            // the oldest step's two requests have landed once at most 2 * (FR_D - 1) are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FR_VMCNT) : "memory");
            const unsigned slot = cnt & (FR_D - 1);
            const char* sp = ringp + slot * FR_SLOT;
            const uint4 gq = *(const uint4*)(sp + lane * 16);
            const float x0 = *(const float*)(sp + 1024 + mp * 4);
            const float x1 = *(const float*)(sp + 1024 + mp * 4 + 4);
            const float x2 = *(const float*)(sp + 1024 + mp * 4 + 8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // in registers before the slot is requested again
            FR_REQUEST(slot);
            ++cnt;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ra[k] = rb[k]; rb[k] = rc[k]; }
            rc[0] = x0; rc[1] = x1; rc[2] = x2;
            if (j < 2 || !mine) continue;
            float fd[EPV];
            Vec16<T>::unpack(gq, fd);
#ifdef FR_X_NOCOMP
            part[0] += fd[0] + rc[0] + ra[1] + rb[2];
            continue;
#endif
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float v = b[e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    v = fmaf(ra[k], wr[k][e], v);
                    v = fmaf(rb[k], wr[3 + k][e], v);
                    v = fmaf(rc[k], wr[6 + k][e], v);
                }
                const float z = round_to<T>(v);
                const float d = (!RELU || fmaf(z, sc[e], sh[e]) > 0.f) ? fd[e] : 0.f;
                const float xh = (z - mu[e]) * is[e];
                part[e] += d;
                part[EPV + e] += d * xh;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pw[k][e] = fmaf(d, ra[k], pw[k][e]);
                    pw[3 + k][e] = fmaf(d, rb[k], pw[3 + k][e]);
                    pw[6 + k][e] = fmaf(d, rc[k], pw[6 + k][e]);
                }
            }
        }
    }
    // the dummy requests of the last FR_D steps are still landing in the ring, and the reductions reuse its LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    block_reduce_to_acc<2 * EPV>(m, part, EPV, Cout, acc, 2);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        float pt[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) pt[e] = pw[t][e];
        block_reduce_to_acc<EPV>(m, pt, EPV, Cout, wacc + t, 9);
    }
#undef FR_OPEN
#undef FR_REQUEST
#undef FR_DMA_DY
#undef FR_DMA_X
#undef FR_VMCNT
}

// The fused forward pass (first1_bn_act_kernel's arithmetic) on the same ring: only the image row travels through LDS, the output
// leaves by ordinary 16-byte stores.  What this buys is on the STORE side: vmcnt counts stores too and returns in order, so
// the register-prefetch form, which waits for the next image row every step, also waited for the previous step's store to be
// acknowledged -- one 1-KiB store in flight per wave, 3.3 TB/s.  Here the wait is counted by hand: the row of step s is older than
// the FA_D - 1 requests issued after it whatever the stores do (steps 0 and 1 of a tile and waves without columns store nothing),
// so vmcnt(FA_D - 1) is always sufficient, and in the steady state it leaves about FA_D / 2 stores per wave in flight.
constexpr int FA_D = 16;
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void first1_bn_act_ring_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 T* __restrict__ y, int ldy, int yoff, int N, int H, int W, int Cout,
                                                                 unsigned xbytes) {
    constexpr int EPV = TT<T>::EPV;
    constexpr int XSLOT = 256;
    extern __shared__ float red_lds[];
    ChanMap m; m.init<T>(Cout);
    float wr[9][EPV], b[EPV], sc[EPV], sh[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        const int c = (m.active ? m.cg : 0) * EPV + e;
        b[e] = bias != nullptr ? bias[c] : 0.f;
        sc[e] = scale[c]; sh[e] = shift[c];
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][e] = w[c * 9 + t];
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pb = (wave * 64) / m.G;
    const int npx = min((wave * 64 + 63) / m.G, m.PL - 1) - pb + 1;
    const int mp = m.pl - pb;
    const int tiles_w = (W + m.PL - 1) / m.PL, tiles_h = (H + FR_ROWS - 1) / FR_ROWS;
    const int per_img = tiles_w * tiles_h;
    const int total = per_img * N;
    const i32x4_t xrsrc = make_rsrc(x, xbytes);
    const unsigned ring0 = lds_addr(red_lds) + (unsigned)wave * (FA_D * XSLOT);
    const char* ringp = (const char*)red_lds + wave * (FA_D * XSLOT);

    int ptile = blockIdx.x, pj = 0, pslots = 0, phh = 0;
    unsigned pxoff = 0;
    bool pxcol = false;
#define FA_OPEN()                                                                                                   \
    do {                                                                                                            \
        if (ptile >= total) { pslots = 0; break; }                                                                  \
        const int n_ = ptile / per_img, tt_ = ptile - n_ * per_img;                                                 \
        const int th_ = tt_ / tiles_w, tw_ = tt_ - th_ * tiles_w;                                                   \
        const int h0_ = th_ * FR_ROWS, w0_ = tw_ * m.PL;                                                            \
        pslots = min(h0_ + FR_ROWS, H) - h0_ + 2;                                                                   \
        pj = 0;                                                                                                     \
        phh = h0_ - 1;                                                                                              \
        const int col_ = w0_ + pb - 1 + lane;                                                                       \
        pxcol = lane < npx + 2 && col_ >= 0 && col_ < W;                                                            \
        pxoff = (unsigned)((((long long)n_ * H + (h0_ - 1)) * W + col_) * 4);                                       \
    } while (0)
#define FA_REQUEST(slot_)                                                                                           \
    do {                                                                                                            \
        const unsigned dst_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring0 + (slot_) * XSLOT));             \
        lds_dma4_buf(xrsrc, (pslots != 0 && pxcol && (unsigned)phh < (unsigned)H) ? pxoff : LDS_DMA_OOB, dst_);     \
        if (pslots != 0) {                                                                                          \
            ++pj; ++phh; pxoff += (unsigned)W * 4u;                                                                 \
            if (pj == pslots) { ptile += gridDim.x; FA_OPEN(); }                                                    \
        }                                                                                                           \
    } while (0)
    FA_OPEN();
#pragma unroll 1
    for (int k = 0; k < FA_D; ++k) FA_REQUEST((unsigned)k);

    unsigned cnt = 0;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / per_img, tt = tile - n * per_img;
        const int th = tt / tiles_w, tw = tt - th * tiles_w;
        const int h0 = th * FR_ROWS;
        const int slots = min(h0 + FR_ROWS, H) - h0 + 2;
        const int wq = tw * m.PL + m.pl;
        const bool mine = m.active && wq < W;
        T* yp = y + (((long long)n * H + h0) * W + (mine ? wq : 0)) * ldy + yoff + (m.active ? m.cg : 0) * EPV;
        float ra[3] = {0.f, 0.f, 0.f}, rb[3] = {0.f, 0.f, 0.f}, rc[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < slots; ++j) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FA_D - 1) : "memory");
            const unsigned slot = cnt & (FA_D - 1);
            const char* sp = ringp + slot * XSLOT;
            const float x0 = *(const float*)(sp + mp * 4);
            const float x1 = *(const float*)(sp + mp * 4 + 4);
            const float x2 = *(const float*)(sp + mp * 4 + 8);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FA_REQUEST(slot);
            ++cnt;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ra[k] = rb[k]; rb[k] = rc[k]; }
            rc[0] = x0; rc[1] = x1; rc[2] = x2;
            float o[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float v = b[e];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    v = fmaf(ra[k], wr[k][e], v);
                    v = fmaf(rb[k], wr[3 + k][e], v);
                    v = fmaf(rc[k], wr[6 + k][e], v);
                }
                v = fmaf(round_to<T>(v), sc[e], sh[e]);
                o[e] = RELU ? fmaxf(v, 0.f) : v;
            }
            if (j >= 2 && mine) *(uint4*)(yp + (long long)(j - 2) * W * ldy) = Vec16<T>::pack(o);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef FA_OPEN
#undef FA_REQUEST
}

static __global__ void first1_store_kernel(const double* __restrict__ dbacc, const double* __restrict__ wacc, int C,
                                           float* __restrict__ dbias, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C && dbias != nullptr) dbias[i] = (float)dbacc[i];
    if (i < C * 9) dw[i] = (float)wacc[i];
}

// Patch moments of the 1-channel input: s[t] = sum_p u_p[t], R[t][t'] = sum_p u_p[t] u_p[t'] (t <= t'), n = pixels, with
// u_p the zero-padded 3x3 patch of pixel p.  mom[55] fp64, accumulated with one atomic per value and workgroup.
__global__ __launch_bounds__(256) void first1_moments_kernel(const float* __restrict__ x, int N, int H, int W, double* __restrict__ mom) {
    constexpr int PX = 4;                      // consecutive pixels of a row per thread: 18 loads for 4 patches
    __shared__ double part[4][55];
    const int Wg = (W + PX - 1) / PX;
    const long long groups = (long long)N * H * Wg;
    // fp64 accumulators: a product of two floats is exact in double, so the only rounding is the fp64 accumulation.  The
    // forward statistics (first1_stats_from_moments_kernel) take a variance out of these sums by cancellation; the kernel is
    // bound by its loads, not by the 54 double FMAs per pixel.
    double s[9], r[45], cnt = 0.0;
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] = 0.0;
#pragma unroll
    for (int t = 0; t < 45; ++t) r[t] = 0.0;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < groups; q += (long long)gridDim.x * 256) {
        const int wg = (int)(q % Wg);
        const long long t1 = q / Wg;
        const int h = (int)(t1 % H), n = (int)(t1 / H);
        const int w0 = wg * PX;
        const float* xi = x + (long long)n * H * W;
        float win[3][PX + 2];
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
            const int hh = h + dr - 1;
            const bool rowok = (unsigned)hh < (unsigned)H;
            const float* xr = xi + (long long)hh * W;
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) {
                const int ww = w0 + j - 1;
                win[dr][j] = (rowok && (unsigned)ww < (unsigned)W) ? xr[ww] : 0.f;
            }
        }
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            if (w0 + px >= W) break;
            cnt += 1.0;
            double u[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) u[t] = (double)win[t / 3][px + t % 3];
            int k = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                s[t] += u[t];
#pragma unroll
                for (int t2 = t; t2 < 9; ++t2) { r[k] = fma(u[t], u[t2], r[k]); ++k; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 55; ++i) {
        double v = i < 9 ? s[i] : (i < 54 ? r[i < 54 ? i - 9 : 0] : cnt);
        for (int st = 32; st >= 1; st >>= 1) v += __shfl_xor(v, st);
        if (lane == 0) part[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 55)
        atomicAdd(&mom[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// BatchNorm batch statistics of the first conv from the patch moments: z = w.u + b is linear in the pixel's 3x3 patch u, so
//   sum_p z = w.s + n b,   sum_p z^2 = w^T R w + 2 b (w.s) + n b^2
// -- 9 + 45 accumulators per pixel (first1_moments_kernel) instead of all Cout channels of conv3x3(x) computed only to be summed
// (VALU-bound: 257 us against 73 us at 32x512x512, 64 channels).  fp64 throughout; the two sums leave as TWO fp32 rows (value and
// the rounding remainder) so that the fp64 row sum of unet_bn_finalize sees ~48 bits of them.  The statistics are those of the
// UNROUNDED conv output; the bf16 path normalises round_bf16(z), whose rounding is unbiased and averages out over the pixels.
__global__ __launch_bounds__(256) void first1_stats_from_moments_kernel(const double* __restrict__ mom, const float* __restrict__ w,
                                                                        const float* __restrict__ bias, int C, float* __restrict__ rows) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = mom[54], b = bias != nullptr ? (double)bias[c] : 0.0;
    double wc[9], ws = 0.0, q = 0.0;
#pragma unroll
    for (int t = 0; t < 9; ++t) { wc[t] = (double)w[c * 9 + t]; ws += wc[t] * mom[t]; }
    int k = 9;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int t2 = t; t2 < 9; ++t2) { q += (t2 == t ? 1.0 : 2.0) * wc[t] * wc[t2] * mom[k]; ++k; }
    const double S1 = ws + n * b, S2 = q + 2.0 * b * ws + n * b * b;
    const float h1 = (float)S1, h2 = (float)S2;
    rows[c * 2] = h1; rows[c * 2 + 1] = h2;
    rows[(C + c) * 2] = (float)(S1 - (double)h1); rows[(C + c) * 2 + 1] = (float)(S2 - (double)h2);
}

// Everything the first layer's backward returns, from pass 3's sums and the input's patch moments.  With d = dy' (ReLU-masked),
// z = w.u + b, xhat = (z - mu)*is and dz = sc*(d - c1 - xhat*c2) (c1 = S1/n, c2 = S2/n; both 0 with frozen statistics):
//   dW[c][t]  = sum_p dz u[t] = sc*(A[t] - c1 s[t]) - sc*c2*is*(B[t] - mu s[t]),   B[t] = sum_p z u[t] = sum_t' w[t'] R[t'][t] + b s[t]
//   dbias[c]  = sum_p dz      = sc*(S1 - n c1) - sc*c2*is*(w.s + n b - n mu)        (rounding noise in train mode, as in the reference)
// so the second pass over dy (which existed only to multiply dz with the patches) is not needed.  B uses the unrounded z; in bf16
// mode the stored-type rounding of z (2^-9 relative, unbiased) averages out of a sum over all pixels.  fp64 throughout.
__global__ __launch_bounds__(256) void first1_bwd_finalize_kernel(const double* __restrict__ acc, const double* __restrict__ wacc,
                                                                  const double* __restrict__ mom, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const float* __restrict__ scale,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  int frozen, int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dbias, float* __restrict__ dw) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = mom[54];
    const double S1 = acc[c * 2], S2 = acc[c * 2 + 1];
    dgamma[c] = (float)S2;
    dbeta[c] = (float)S1;
    const double c1 = frozen ? 0.0 : (double)(float)(S1 / n), c2 = frozen ? 0.0 : (double)(float)(S2 / n);   // the fp32 coefficients the two-pass form applies
    const double sc = scale[c], is = invstd[c], mu = mean[c], b = bias != nullptr ? (double)bias[c] : 0.0;
    double wc[9], ws = 0.0;
#pragma unroll
    for (int t = 0; t < 9; ++t) { wc[t] = (double)w[c * 9 + t]; ws += wc[t] * mom[t]; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        double B = b * mom[t];
#pragma unroll
        for (int t2 = 0; t2 < 9; ++t2) {
            const int lo = t2 < t ? t2 : t, hi = t2 < t ? t : t2;
            B += wc[t2] * mom[9 + lo * 9 - lo * (lo - 1) / 2 + (hi - lo)];     // row lo of the upper triangle starts at 9 + sum_{i<lo} (9 - i)
        }
        dw[c * 9 + t] = (float)(sc * (wacc[c * 9 + t] - c1 * mom[t]) - sc * c2 * is * (B - mu * mom[t]));
    }
    if (dbias != nullptr) dbias[c] = (float)(sc * (S1 - n * c1) - sc * c2 * is * (ws + n * b - n * mu));
}

// =================================================================================================
// last layer
// =================================================================================================

// logits[n][k][hw] = bias[k] + sum_c w[k][c] * round_T(act(z*scale + shift)); y (optional) receives the activation.
// LP lanes (a power of two >= C/EPV) cooperate on one pixel.
template <typename T, int K, bool RELU>
__global__ __launch_bounds__(256) void affine_act_out_kernel(const T* __restrict__ z, int ldz, int zoff,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ logits, T* __restrict__ y, int ldy, int yoff,
                                                             long long M, long long HW, int C, int LP) {
    constexpr int EPV = TT<T>::EPV;
    const int G = C / EPV;
    const int sub = threadIdx.x & (LP - 1);
    const int ppb = 256 / LP;
    const int pslot = threadIdx.x / LP;
    float wr[K][EPV], sc[EPV], sh[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
        sc[e] = sub < G ? scale[sub * EPV + e] : 0.f;
        sh[e] = sub < G ? shift[sub * EPV + e] : 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) wr[k][e] = sub < G ? w[k * C + sub * EPV + e] : 0.f;
    }
    float bk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) bk[k] = bias != nullptr ? bias[k] : 0.f;
    // AO_U CONSECUTIVE pixels per thread and trip: all requests of the trip go out before the first value is used, and lane
    // sub == 0 writes its pixels' logits of a class as one 16-byte store (a wave's 8 pixel slots: 128 contiguous bytes per class).
    // One pixel per trip meant one 16-byte request per thread in flight and two 32-byte stores per wave behind every load,
    // which the next trip's load then waited for (vmcnt is in order): 4.0 TB/s.  Every lane of a wave runs the same trip
    // count so the shuffles below see all lanes.
    constexpr int AO_U = 4;
    const long long span = (long long)ppb * AO_U;                  // pixels per workgroup and trip
    const long long iters = (M + (long long)gridDim.x * span - 1) / ((long long)gridDim.x * span);
    // a thread's 4 pixels never straddle two images when 4 divides HW; the 16-byte store also needs an aligned base
    const bool vec_ok = (HW % AO_U) == 0 && ((size_t)logits & 15) == 0;
    for (long long it = 0; it < iters; ++it) {
        const long long pbase = (it * gridDim.x + blockIdx.x) * span + (long long)pslot * AO_U;
        uint4 vz[AO_U];
#pragma unroll
        for (int u = 0; u < AO_U; ++u)
            if (pbase + u < M && sub < G) vz[u] = *(const uint4*)(z + (pbase + u) * ldz + zoff + sub * EPV);
        float d[K][AO_U];
#pragma unroll
        for (int u = 0; u < AO_U; ++u) {
            const long long p = pbase + u;
            float f[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) f[e] = 0.f;
            if (p < M && sub < G) {
                Vec16<T>::unpack(vz[u], f);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float v = fmaf(f[e], sc[e], sh[e]);
                    f[e] = RELU ? fmaxf(v, 0.f) : v;
                }
                const uint4 o = Vec16<T>::pack(f);
                if (y != nullptr) *(uint4*)(y + p * ldy + yoff + sub * EPV) = o;
                Vec16<T>::unpack(o, f);                   // the 1x1 conv sees what a stored activation would hold
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < EPV; ++e) s = fmaf(f[e], wr[k][e], s);
                for (int o = 1; o < LP; o <<= 1) s += __shfl_xor(s, o);
                d[k][u] = s + bk[k];
            }
        }
        if (sub == 0 && pbase < M) {
            if (vec_ok && pbase + AO_U <= M) {
                const long long n = pbase / HW, hw = pbase - n * HW;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    *(float4*)(logits + (n * K + k) * HW + hw) = make_float4(d[k][0], d[k][1], d[k][2], d[k][3]);
            } else {
#pragma unroll
                for (int u = 0; u < AO_U; ++u) {
                    const long long p = pbase + u;
                    if (p < M) {
                        const long long n = p / HW, hw = p - n * HW;
#pragma unroll
                        for (int k = 0; k < K; ++k) logits[(n * K + k) * HW + hw] = d[k][u];
                    }
                }
            }
        }
    }
}

// BN+ReLU backward of the last DoubleConv layer with dy[p][c] = round_T(sum_k dlogits[n][k][hw] * w[k][c]).
//   PASS 1: S1, S2 -> acc[C][2];  dW[k][c] = sum_p dl[k][p] * y[p][c] -> oacc[k*C + c];  db[k] = sum_p dl[k][p] -> oacc[K*C + k]
//   PASS 2: dz = scale * (dy' - c1 - xhat*c2) stored; sum of the stored dz -> dbias acc[C]
template <typename T, int K, bool RELU, int PASS>
__global__ __launch_bounds__(256) void bn_bwd_out_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                         const T* __restrict__ z, int ldz, int zoff,
                                                         T* __restrict__ dz, int ldo, int ooff,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         const float* __restrict__ c1, const float* __restrict__ c2,
                                                         long long M, long long HW, int C,
                                                         double* __restrict__ acc, double* __restrict__ oacc) {
    constexpr int EPV = TT<T>::EPV;
    constexpr int KW_ = PASS == 1 ? K : 1;
    extern __shared__ float red_lds[];
    ChanMap m; m.init<T>(C);
    float part[2 * EPV], pw[KW_][EPV], pb[KW_];
#pragma unroll
    for (int i = 0; i < 2 * EPV; ++i) part[i] = 0.f;
#pragma unroll
    for (int k = 0; k < KW_; ++k) {
        pb[k] = 0.f;
#pragma unroll
        for (int e = 0; e < EPV; ++e) pw[k][e] = 0.f;
    }
    if (m.active) {
        float wr[K][EPV], sc[EPV], sh[EPV], mu[EPV], is[EPV], k1[EPV], k2[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            const int c = m.cg * EPV + e;
            sc[e] = scale[c]; sh[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
            k1[e] = PASS == 2 ? c1[c] : 0.f; k2[e] = PASS == 2 ? c2[c] : 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) wr[k][e] = w[k * C + c];
        }
        // OUT_U pixels per trip, every request of the trip issued before the first value is used (one pixel per trip left one
        // 16-byte request per thread in flight: 3.2 TB/s in pass 1); the image index comes from a 32-bit division where the
        // tensor allows it
        constexpr int OUT_U = PASS == 1 ? 4 : 1;            // pass 2 streams at 5 TB/s as it is; 4 pixels there cost it half its occupancy
        const long long stride = (long long)gridDim.x * m.PL;
        const bool small = M < (1ll << 31);
        for (long long p0 = (long long)blockIdx.x * m.PL + m.pl; p0 < M; p0 += stride * OUT_U) {
            uint4 vz[OUT_U];
            float gg[OUT_U][K];
#pragma unroll
            for (int u = 0; u < OUT_U; ++u) {
                const long long p = p0 + u * stride;
                if (p < M) {
                    vz[u] = *(const uint4*)(z + p * ldz + zoff + m.cg * EPV);
                    const long long n = small ? (long long)((unsigned)p / (unsigned)HW) : p / HW;
                    const float* dp = dl + p + n * (K - 1) * HW;           // dl[(n*K + k)*HW + hw] with hw = p - n*HW
#pragma unroll
                    for (int k = 0; k < K; ++k) gg[u][k] = dp[k * HW];
                }
            }
#pragma unroll
            for (int u = 0; u < OUT_U; ++u) {
                const long long p = p0 + u * stride;
                if (p >= M) break;
                float fz[EPV], g[K];
                Vec16<T>::unpack(vz[u], fz);
#pragma unroll
                for (int k = 0; k < K; ++k) g[k] = gg[u][k];
                float o[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    float dyv = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) dyv = fmaf(g[k], wr[k][e], dyv);
                    dyv = round_to<T>(dyv);
                    const float a = fmaf(fz[e], sc[e], sh[e]);
                    const float d = (!RELU || a > 0.f) ? dyv : 0.f;
                    const float xh = (fz[e] - mu[e]) * is[e];
                    if constexpr (PASS == 1) {
                        part[e] += d;
                        part[EPV + e] += d * xh;
                        const float yv = round_to<T>(RELU ? fmaxf(a, 0.f) : a);
#pragma unroll
                        for (int k = 0; k < K; ++k) pw[k][e] = fmaf(g[k], yv, pw[k][e]);
                    } else {
                        o[e] = sc[e] * (d - k1[e] - xh * k2[e]);
                    }
                }
                if constexpr (PASS == 1) {
#pragma unroll
                    for (int k = 0; k < K; ++k) pb[k] += g[k];
                } else {
                    const uint4 ov = Vec16<T>::pack(o);
                    *(uint4*)(dz + p * ldo + ooff + m.cg * EPV) = ov;
                    float fr[EPV];
                    Vec16<T>::unpack(ov, fr);              // sum what the consumers of dz will actually read
#pragma unroll
                    for (int e = 0; e < EPV; ++e) part[e] += fr[e];
                }
            }
        }
    }
    if constexpr (PASS == 1) {
        block_reduce_to_acc<2 * EPV>(m, part, EPV, C, acc, 2);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            __syncthreads();
            float pt[EPV];
#pragma unroll
            for (int e = 0; e < EPV; ++e) pt[e] = pw[k][e];
            block_reduce_to_acc<EPV>(m, pt, EPV, C, oacc + (size_t)k * C, 1);
        }
        // db: every channel group of a pixel lane carries the same copy; take the cg == 0 threads'
        __syncthreads();
        if (m.active && m.cg == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) red_lds[m.pl * K + k] = pb[k];
        }
        __syncthreads();
        if (threadIdx.x < K) {
            float s = 0.f;
            for (int p = 0; p < m.PL; ++p) s += red_lds[p * K + threadIdx.x];
            atomicAdd(&oacc[(size_t)K * C + threadIdx.x], (double)s);
        }
    } else {
        float p1[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) p1[e] = part[e];
        block_reduce_to_acc<EPV>(m, p1, EPV, C, acc, 1);
    }
}

static __global__ void out_store_kernel(const double* __restrict__ oacc, int n, float* __restrict__ dw, int ndw, float* __restrict__ db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ndw) dw[i] = (float)oacc[i];
    else if (i < n) db[i - ndw] = (float)oacc[i];
}

// =================================================================================================
static inline int ef_cap() {
    static const int v = [] { const char* e = getenv("UNET_AMD_BN_GRID"); const int g = e ? atoi(e) : 0; return g > 0 ? g : STREAM_GRID_CAP; }();
    return v;
}
static inline int ef_grid(long long M, int PL) {
    long long g = (M + PL - 1) / PL;
    if (g > ef_cap()) g = ef_cap();
    if (g < 1) g = 1;
    return (int)g;
}
static inline int ef_check(int dtype, int C, const char* what) {
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    if (dtype != UNET_F32 && dtype != UNET_BF16) { unet_set_error("%s: bad dtype %d", what, dtype); return -1; }
    if (C <= 0 || C % 8 != 0 || C / epv > 256) { unet_set_error("%s: C=%d must be a multiple of 8 and at most %d", what, C, 256 * epv); return -1; }
    return 0;
}
static int first_gx2(int dtype, int N, int H, int W, int Cout) {
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    const int G = Cout / epv;
    int cap = MAX_GRID / N; if (cap < 1) cap = 1;
    long long g = ((long long)H * W + 256 / G - 1) / (256 / G);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int unet_first_bn_act(int dtype, int relu, const float* x, const float* w, const float* bias,
                                 const float* scale, const float* shift, void* y, int ldy, int yoff,
                                 int N, int H, int W, int Cout, void* stream) {
    if (ef_check(dtype, Cout, "first_bn_act")) return -1;
    UNET_CHECK(x && w && scale && shift && y, "first_bn_act: null argument");
    UNET_CHECK(N > 0 && H > 0 && W > 0 && ldy % 8 == 0 && yoff % 8 == 0 && yoff + Cout <= ldy, "first_bn_act: bad shape");
    hipStream_t s = (hipStream_t)stream;
    // the image rows through the LDS ring (first1_bn_act_ring_kernel) under the same conditions as the backward;
    // UNET_AMD_FIRST_FWD_RING=0: the register-prefetch kernel (A/B runs)
    static const bool ring_off = [] { const char* e = getenv("UNET_AMD_FIRST_FWD_RING"); return e && atoi(e) == 0; }();
    const int G = Cout / (dtype == UNET_BF16 ? 8 : 4);
    const unsigned long long xbytes = (unsigned long long)N * H * W * 4ull;
    const long long rt = (long long)N * ((H + FR_ROWS - 1) / FR_ROWS) * ((W + 256 / G - 1) / (256 / G));
    if (!ring_off && G >= 2 && xbytes < 0xFFFFFF00ull && rt < (1ll << 30)) {
        const int rgrid = (int)(rt < 1024 ? rt : 1024);            // four workgroups per CU
#define LA(T, R) hipLaunchKernelGGL((first1_bn_act_ring_kernel<T, R>), dim3(rgrid), dim3(256), (size_t)4 * FA_D * 256, s, x, w, bias, scale, shift, (T*)y, ldy, yoff, N, H, W, Cout, (unsigned)xbytes)
        if (dtype == UNET_BF16) { if (relu & 1) LA(__bf16, true); else LA(__bf16, false); }
        else { if (relu & 1) LA(float, true); else LA(float, false); }
#undef LA
        UNET_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid(first_gx2(dtype, N, H, W, Cout), N);
#define LA(T, R) hipLaunchKernelGGL((first1_bn_act_kernel<T, R>), grid, dim3(256), 0, s, x, w, bias, scale, shift, (T*)y, ldy, yoff, H, W, Cout)
    if (dtype == UNET_BF16) { if (relu & 1) LA(__bf16, true); else LA(__bf16, false); }
    else { if (relu & 1) LA(float, true); else LA(float, false); }
#undef LA
    UNET_LAUNCH_CHECK();
    return 0;
}

static int launch_first_moments(const float* x, int N, int H, int W, double* mom, hipStream_t s) {
    const long long groups = (long long)N * H * ((W + 3) / 4);
    long long mg = (groups + 255) / 256;
    // 55 fp64 atomics per workgroup: keep them few -- but one workgroup per CU (256) left the pass latency-bound (104 us for
    // 33 MB; 512: 82 us, 1024: 110, 2048: 120 -- the atomics take over); UNET_AMD_MOMENTS_GRID overrides (experiments)
    static const long long cap = [] { const char* e = getenv("UNET_AMD_MOMENTS_GRID"); const long v = e ? atol(e) : 0; return (long long)(v > 0 ? v : 512); }();
    if (mg > cap) mg = cap;
    hipLaunchKernelGGL(first1_moments_kernel, dim3((unsigned)mg), dim3(256), 0, s, x, N, H, W, mom);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_first_stats_moments(const float* x, const float* w, const float* bias, int N, int H, int W, int Cout,
                                        double* mom /* 64 doubles, written */, float* stats /* [2][Cout][2] */, void* stream) {
    UNET_CHECK(x && w && mom && stats && N > 0 && H > 0 && W > 0 && Cout > 0, "first_stats_moments: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(mom, 0, sizeof(double) * 64, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    if (int rc = launch_first_moments(x, N, H, W, mom, s)) return rc;
    hipLaunchKernelGGL(first1_stats_from_moments_kernel, dim3((Cout + 255) / 256), dim3(256), 0, s, mom, w, bias, Cout, stats);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_first_bn_relu_bwd(int dtype, int relu, const void* dy, int ldd, int doff,
                                      const float* x, const float* w, const float* bias,
                                      const float* scale, const float* shift, const float* mean, const float* invstd,
                                      float* dgamma, float* dbeta, float* dbias /* nullable */, float* dw /* [Cout][1][3][3] */,
                                      float* c1c2_ws /* 2*Cout floats */, double* acc_ws /* 12*Cout + 64 doubles */,
                                      const double* moments /* nullable: the 64 doubles unet_first_stats_moments wrote for this x */,
                                      int N, int H, int W, int Cout, void* stream) {
    if (ef_check(dtype, Cout, "first_bn_relu_bwd")) return -1;
    UNET_CHECK(dy && x && w && scale && shift && mean && invstd && dgamma && dbeta && dw && c1c2_ws && acc_ws, "first_bn_relu_bwd: null argument");
    UNET_CHECK(N > 0 && H > 0 && W > 0 && ldd % 8 == 0 && doff % 8 == 0 && doff + Cout <= ldd, "first_bn_relu_bwd: bad shape");
    const int frozen = (relu >> 1) & 1;
    relu &= 1;
    hipStream_t s = (hipStream_t)stream;
    const int epv = dtype == UNET_BF16 ? 8 : 4, G = Cout / epv;
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * (12 * Cout + 64), s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    double* dbacc = acc_ws + 2 * Cout;
    double* wacc = acc_ws + 3 * Cout;
    double* mom = acc_ws + 12 * Cout;
    float* c1 = c1c2_ws; float* c2 = c1c2_ws + Cout;
    const long long tiles = (long long)N * ((H + FIRST_R - 1) / FIRST_R) * ((W + 256 / G - 1) / (256 / G));
    const int grid = (int)(tiles < MAX_GRID ? tiles : MAX_GRID);
    const size_t lds1 = reduce_lds_bytes(G, 2 * epv), lds2 = reduce_lds_bytes(G, epv);
    // UNET_AMD_FIRST_BWD_2PASS=1: the two-pass form (dz recomputed and multiplied with the patches in a second pass), for A/B runs
    static const bool two_pass = [] { const char* e = getenv("UNET_AMD_FIRST_BWD_2PASS"); return e && atoi(e) != 0; }();
#define LP(T, R, P, L, A1, A2, A3, A4) hipLaunchKernelGGL((first1_bn_bwd_kernel<T, R, P>), dim3(grid), dim3(256), L, s, (const T*)dy, ldd, doff, x, w, bias, scale, shift, mean, invstd, A1, A2, N, H, W, Cout, A3, A4)
#define LPD(P, L, A1, A2, A3, A4) do { if (dtype == UNET_BF16) { if (relu) LP(__bf16, true, P, L, A1, A2, A3, A4); else LP(__bf16, false, P, L, A1, A2, A3, A4); } \
                                       else { if (relu) LP(float, true, P, L, A1, A2, A3, A4); else LP(float, false, P, L, A1, A2, A3, A4); } } while (0)
    if (!two_pass) {
        if (moments == nullptr) { if (int rc = launch_first_moments(x, N, H, W, mom, s)) return rc; }
        const double* mom_use = moments != nullptr ? moments : mom;
        // operands through the LDS ring (first1_bn_bwd_ring_kernel) whenever the image fits a 32-bit buffer descriptor and a wave's
        // pixel run + 2 fits its 64 lanes (G >= 2); UNET_AMD_FIRST_BWD_RING=0: the register-prefetch kernel (A/B runs)
        static const bool ring_off = [] { const char* e = getenv("UNET_AMD_FIRST_BWD_RING"); return e && atoi(e) == 0; }();
        const unsigned long long xbytes = (unsigned long long)N * H * W * 4ull;
        if (!ring_off && G >= 2 && xbytes < 0xFFFFFF00ull && (long long)N * ((H + FR_ROWS - 1) / FR_ROWS) * ((W + 256 / G - 1) / (256 / G)) < (1ll << 30)) {
            const long long rt = (long long)N * ((H + FR_ROWS - 1) / FR_ROWS) * ((W + 256 / G - 1) / (256 / G));
            const int rgrid = (int)(rt < 512 ? rt : 512);          // two workgroups per CU are resident (register budget): one wave of them
            size_t rl = first_ring_lds_bytes(); if (rl < lds1) rl = lds1;
#define LR(T, R) hipLaunchKernelGGL((first1_bn_bwd_ring_kernel<T, R>), dim3(rgrid), dim3(256), rl, s, (const T*)dy, ldd, doff, x, w, bias, scale, shift, mean, invstd, N, H, W, Cout, (unsigned)xbytes, acc_ws, wacc)
            if (dtype == UNET_BF16) { if (relu) LR(__bf16, true); else LR(__bf16, false); }
            else { if (relu) LR(float, true); else LR(float, false); }
#undef LR
        } else {
            LPD(3, lds1, (const float*)nullptr, (const float*)nullptr, acc_ws, wacc);
        }
        UNET_LAUNCH_CHECK();
        hipLaunchKernelGGL(first1_bwd_finalize_kernel, dim3((Cout + 255) / 256), dim3(256), 0, s, acc_ws, wacc, mom_use, w, bias, scale, mean, invstd,
                           frozen, Cout, dgamma, dbeta, dbias, dw);
        UNET_LAUNCH_CHECK();
        return 0;
    }
    LPD(1, lds1, (const float*)nullptr, (const float*)nullptr, acc_ws, (double*)nullptr);
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((Cout + 255) / 256), dim3(256), 0, s, acc_ws, (double)N * H * W, Cout, frozen, dgamma, dbeta, c1, c2);
    UNET_LAUNCH_CHECK();
    LPD(2, lds2, (const float*)c1, (const float*)c2, dbacc, wacc);
    UNET_LAUNCH_CHECK();
#undef LP
#undef LPD
    hipLaunchKernelGGL(first1_store_kernel, dim3((Cout * 9 + 255) / 256), dim3(256), 0, s, dbacc, wacc, Cout, dbias, dw);
    UNET_LAUNCH_CHECK();
    return 0;
}

static int lanes_per_pixel2(int G) { int lp = 1; while (lp < G) lp <<= 1; return lp; }

extern "C" int unet_affine_act_out(int dtype, int relu, const void* z, int ldz, int zoff, const float* scale, const float* shift,
                                   const float* w, const float* bias, float* logits, void* y /* nullable */, int ldy, int yoff,
                                   long long M, long long HW, int C, int K, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    const int epv = dtype == UNET_BF16 ? 8 : 4;
    UNET_CHECK(C % 8 == 0 && C / epv <= 64, "affine_act_out: C=%d must be a multiple of 8 and at most %d", C, 64 * epv);
    UNET_CHECK(K >= 1 && K <= 4, "affine_act_out: 1..4 classes (got %d); use unet_affine_act + unet_conv_out_fwd", K);
    UNET_CHECK(z && scale && shift && w && logits && HW > 0 && M % HW == 0, "affine_act_out: bad arguments");
    UNET_CHECK(ldz % 8 == 0 && zoff % 8 == 0 && (y == nullptr || (ldy % 8 == 0 && yoff % 8 == 0)), "affine_act_out: strides/offsets must be multiples of 8");
    const int LP = lanes_per_pixel2(C / epv);
    // one request per thread per trip and a shuffle reduction behind it: this pass wants MORE resident workgroups than the
    // read+write streaming passes (8 per CU: 290 us, 4 per CU: 377 us at 32 x 512^2 x 64)
    long long gl = (M + 4 * (256 / LP) - 1) / (4 * (256 / LP));    // AO_U = 4 pixels per thread and trip
    const int grid = (int)(gl > MAX_GRID ? MAX_GRID : (gl < 1 ? 1 : gl));
    hipStream_t s = (hipStream_t)stream;
#define LO(T, KK, R) hipLaunchKernelGGL((affine_act_out_kernel<T, KK, R>), dim3(grid), dim3(256), 0, s, (const T*)z, ldz, zoff, scale, shift, w, bias, logits, (T*)y, ldy, yoff, M, HW, C, LP)
#define LOK(KK) case KK: if (dtype == UNET_BF16) { if (relu & 1) LO(__bf16, KK, true); else LO(__bf16, KK, false); } \
                         else { if (relu & 1) LO(float, KK, true); else LO(float, KK, false); } break;
    switch (K) { LOK(1) LOK(2) LOK(3) LOK(4) }
#undef LO
#undef LOK
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_bn_relu_bwd_out(int dtype, int relu, const float* dlogits, const float* w, int K, long long HW,
                                    const void* z, int ldz, int zoff, void* dz, int ldo, int ooff,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, float* dbias /* nullable */, float* dw /* [K][C] */, float* db /* [K] */,
                                    float* c1c2_ws /* 2*C floats */, double* acc_ws /* 3*C + K*C + K doubles */,
                                    long long M, int C, void* stream) {
    if (ef_check(dtype, C, "bn_relu_bwd_out")) return -1;
    UNET_CHECK(K >= 1 && K <= 4, "bn_relu_bwd_out: 1..4 classes (got %d); use unet_conv_out_bwd + unet_bn_relu_bwd", K);
    UNET_CHECK(dlogits && w && z && dz && scale && shift && mean && invstd && dgamma && dbeta && dw && db && c1c2_ws && acc_ws,
               "bn_relu_bwd_out: null argument");
    UNET_CHECK(HW > 0 && M % HW == 0 && ldz % 8 == 0 && zoff % 8 == 0 && ldo % 8 == 0 && ooff % 8 == 0, "bn_relu_bwd_out: bad shape");
    const int frozen = (relu >> 1) & 1;
    relu &= 1;
    hipStream_t s = (hipStream_t)stream;
    const int epv = dtype == UNET_BF16 ? 8 : 4, G = C / epv, PL = 256 / G;
    const int nacc = 3 * C + K * C + K;
    if (hipMemsetAsync(acc_ws, 0, sizeof(double) * nacc, s) != hipSuccess) { unet_set_error("memset failed"); return -2; }
    double* dbacc = acc_ws + 2 * C;
    double* oacc = acc_ws + 3 * C;
    float* c1 = c1c2_ws; float* c2 = c1c2_ws + C;
    const int grid = ef_grid(M, PL);
    size_t lds1 = reduce_lds_bytes(G, 2 * epv);
    if (lds1 < (size_t)PL * K * sizeof(float)) lds1 = (size_t)PL * K * sizeof(float);
    const size_t lds2 = reduce_lds_bytes(G, epv);
#define L1(T, KK, R) hipLaunchKernelGGL((bn_bwd_out_kernel<T, KK, R, 1>), dim3(grid), dim3(256), lds1, s, dlogits, w, (const T*)z, ldz, zoff, (T*)dz, ldo, ooff, scale, shift, mean, invstd, nullptr, nullptr, M, HW, C, acc_ws, oacc)
#define L2(T, KK, R) hipLaunchKernelGGL((bn_bwd_out_kernel<T, KK, R, 2>), dim3(grid), dim3(256), lds2, s, dlogits, w, (const T*)z, ldz, zoff, (T*)dz, ldo, ooff, scale, shift, mean, invstd, c1, c2, M, HW, C, dbacc, nullptr)
#define LK(LL, KK) case KK: if (dtype == UNET_BF16) { if (relu) LL(__bf16, KK, true); else LL(__bf16, KK, false); } \
                            else { if (relu) LL(float, KK, true); else LL(float, KK, false); } break;
    switch (K) { LK(L1, 1) LK(L1, 2) LK(L1, 3) LK(L1, 4) }
    UNET_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, acc_ws, (double)M, C, frozen, dgamma, dbeta, c1, c2);
    UNET_LAUNCH_CHECK();
    switch (K) { LK(L2, 1) LK(L2, 2) LK(L2, 3) LK(L2, 4) }
    UNET_LAUNCH_CHECK();
#undef L1
#undef L2
#undef LK
    hipLaunchKernelGGL(out_store_kernel, dim3((K * C + K + 255) / 256), dim3(256), 0, s, oacc, K * C + K, dw, K * C, db);
    UNET_LAUNCH_CHECK();
    if (dbias != nullptr) {
        hipLaunchKernelGGL(acc_to_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, dbacc, dbias, C, 0);
        UNET_LAUNCH_CHECK();
    }
    return 0;
}
