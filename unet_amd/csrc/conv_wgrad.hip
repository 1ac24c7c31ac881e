// Weight-gradient of the implicit-GEMM convolutions (gfx950), split-K over pixel tiles.
//
//   dW[tap][co][ci] = sum over output pixels m of  dz[m][co] * x[SH*m + tap][ci]
//
// i.e. a GEMM with M = Cout, N = Cin (per tap) and K = N*Hout*Wout pixels.  This file is the exact-f32 path (fp32 mode, the
// parity mode of the shipped `amp: false` configs): one dword per lane and v_mfma_f32_16x16x4_f32; the bf16 weight gradients
// are conv_wgrad2.hip's.  It also holds the split-K reduce kernels and the C entry point of both.
//
// Workgroup (4 waves, two per CU): a block of 32 output channels x 32 input channels x all taps of the launch, accumulated
// in registers over a contiguous range of 8x32 pixel tiles staged in LDS as 128-byte pixel records; the next tile's global
// loads sit in registers while the current one is multiplied; the partial block goes to slab[split][tap][co][ci] (plain
// stores) and a second kernel sums the slabs in a fixed order (bitwise reproducible) into the OIHW gradient.
//
//
// Serves: nn.Conv2d weight grads (reference models/unet.py:10,13; reparm_unet.py:55-61) and the
// nn.ConvTranspose2d(k=2,s=2) weight grad (models/unet.py:35) through the [N,2H,W,2C] view.
#include "common.h"

#include "wgrad_args.h"

// bf16 path (conv_wgrad2.hip)
int wgrad2_splits(int kh, int kw, int sh, int N, int Hout, int Wout, int Cin, int Cout);
int dispatch_wgrad2(int kh, int kw, int sh, const WgradArgs& a, hipStream_t s, int nsplit);

// 128-byte-record LDS image: record `pix`, byte b (b<128).  The XOR keeps the eight 32-byte pieces
// that one 32-lane half of a transposing read touches on distinct banks.
__device__ __forceinline__ int rec128_off(int pix, int b) {
    return pix * 128 + (b ^ ((((pix >> 1) & 1) << 6) | (((pix >> 3) & 1) << 5)));
}

// The x halo image swizzles by the record's COLUMN inside its halo row (not by its linear index): a lane's byte offset for
// (kernel column kw, K-step column c4) is then the same in every halo row, and the row enters the read as an immediate.
__device__ __forceinline__ int recx_off(int pix, int col, int b) {
    return pix * 128 + (b ^ ((((col >> 1) & 1) << 6) | (((col >> 3) & 1) << 5)));
}

template <typename T> struct WG;
template <> struct WG<__bf16> { static constexpr int WT = 2; static constexpr int CT = 64; };   // wave tile 2x2 MFMA tiles, block 64x64
template <> struct WG<float>  { static constexpr int WT = 1; static constexpr int CT = 32; };   // wave tile 1x1, block 32x32

template <typename T, int KH, int KW, int SH, int TG>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {      // two workgroups per CU (LDS: 76 KB each): at most 256 registers per lane
    static_assert(sizeof(T) == 4, "the exact-f32 path; bf16 weight gradients are conv_wgrad2.hip's");
    constexpr int TH = 8, TW = 32;
    constexpr int HH = (TH - 1) * SH + KH, HWD = TW - 1 + KW;
    constexpr int EPV = TT<T>::EPV;
    constexpr int CT = WG<T>::CT, WT = WG<T>::WT;
    constexpr int NTAPS = KH * KW;
    constexpr int DZ_BYTES = TH * TW * 128, HALO_PIX = HH * HWD;
    constexpr int DZ_UNITS = TH * TW * 8, HALO_UNITS = HALO_PIX * 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_dz = smem;
    char* s_x = smem + DZ_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int split = blockIdx.x;
    const int co0 = blockIdx.y * CT, ci0 = blockIdx.z * CT;
    const int wco = (wave >> 1) * (16 * WT), wci = (wave & 1) * (16 * WT);   // wave's corner inside the block tile

    f32x4 acc[TG][WT][WT];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* __restrict__ dzp = (const T*)a.dz;
    const T* __restrict__ xp = (const T*)a.x;

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.tiles_total);

    // per-lane LDS byte offsets of the fragment reads (f32 path): dz record (c4 + g) of a tile row, x record (kw + c4 + g) of a halo row
    int aoff[TW / 4], boff[KW][TW / 4];
#pragma unroll
    for (int c = 0; c < TW / 4; ++c) {
        aoff[c] = rec128_off(4 * c + g, (wco + l15) * 4);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) boff[kw][c] = recx_off(kw + 4 * c + g, kw + 4 * c + g, (wci + l15) * 4);
    }
    // REGISTER PREFETCH: tile t+1's global loads are issued before tile t's MFMAs and stored to LDS after them, so the load
    // latency hides under a tile of matrix work instead of standing between two of them (fp32 3x3: the staging of a
    // 32 x 32-channel block was as long as its MFMAs; 175 -> see DESIGN for the step).  8 + 11 uint4 per thread.
    constexpr int DZ_IT = DZ_UNITS / 256, X_IT = (HALO_UNITS + 255) / 256;
    static_assert(DZ_UNITS % 256 == 0, "dz tile divides among the threads");
    uint4 pdz[DZ_IT], px_[X_IT];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tw_i = t % a.tiles_w; t /= a.tiles_w;
        const int th_i = t % a.tiles_h;
        const int n = t / a.tiles_h;
        const int h0 = th_i * TH, w0 = tw_i * TW;
        {
            const T* base = dzp + (size_t)n * a.Hout * a.Wout * a.lddz + a.dzoff + co0;
#pragma unroll
            for (int it = 0; it < DZ_IT; ++it) {
                const int u = tid + it * 256;
                const int pix = u >> 3, q = u & 7;
                const int r = pix >> 5, c = pix & 31;
                const int ho = h0 + r, wo = w0 + c;
                const bool ok = ho < a.Hout && wo < a.Wout && (co0 + q * EPV) < a.Cout;
                pdz[it] = make_uint4(0, 0, 0, 0);
                if (ok) pdz[it] = *(const uint4*)(base + ((size_t)ho * a.Wout + wo) * a.lddz + q * EPV);
            }
        }
        {
            const T* base = xp + (size_t)n * a.Hin * a.Win * a.ldx + a.xoff + ci0;
#pragma unroll
            for (int it = 0; it < X_IT; ++it) {
                const int u = tid + it * 256;
                const int pix = u >> 3, q = u & 7;
                const int hr = pix / HWD, hc = pix - hr * HWD;
                const int row = h0 * SH + hr - a.pad_h, col = w0 + hc - a.pad_w;
                const bool ok = u < HALO_UNITS && row >= 0 && row < a.Hin && col >= 0 && col < a.Win && (ci0 + q * EPV) < a.Cin;
                px_[it] = make_uint4(0, 0, 0, 0);
                if (ok) px_[it] = *(const uint4*)(base + ((size_t)row * a.Win + col) * a.ldx + q * EPV);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < DZ_IT; ++it) {
            const int u = tid + it * 256;
            *(uint4*)(s_dz + rec128_off(u >> 3, (u & 7) * 16)) = pdz[it];
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int u = tid + it * 256;
            if (u < HALO_UNITS) *(uint4*)(s_x + recx_off(u >> 3, (u >> 3) % HWD, (u & 7) * 16)) = px_[it];
        }
    };
    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();   // previous tile's reads are done
        stash();
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);

        {
            // K-step = 4 pixels; A[row=l15][k=g] = dz[pix+g][co], B[k=g][col=l15] = x[pix'+g][ci].  Every LDS address is a
            // per-lane offset fixed for the kernel's life (aoff / boff, set up once) plus a compile-time immediate (the tile row
            // and the kernel row): the loop is fully unrolled and spends no vector instruction on addresses -- with them
            // (5 per MFMA) two waves per SIMD could not keep the matrix pipe fed.  The fragments of step k+1 are read before the
            // MFMAs of step k.
            // One tile row (8 K-steps) is unrolled at a time: unrolling all 64 lets the compiler hoist reads until it spills.
            constexpr int KS = TW / 4;
#pragma unroll 1
            for (int r = 0; r < TH; ++r) {
                const char* dzr = s_dz + r * (32 * 128);
                const char* xr = s_x + r * SH * (HWD * 128);
                auto rd = [&](int ci4, float& av, float (&bv)[TG]) {
                    av = *(const float*)(dzr + aoff[ci4]);
#pragma unroll
                    for (int t2 = 0; t2 < TG; ++t2) {
                        const int kh = t2 / KW, kw = t2 % KW;
                        bv[t2] = *(const float*)(xr + boff[kw][ci4] + kh * (HWD * 128));
                    }
                };
                float a0, a1, b0[TG], b1[TG];
                rd(0, a0, b0);
#pragma unroll
                for (int ks = 0; ks < KS; ks += 2) {
                    rd(ks + 1, a1, b1);
#pragma unroll
                    for (int t2 = 0; t2 < TG; ++t2) acc[t2][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0[t2], acc[t2][0][0], 0, 0, 0);
                    if (ks + 2 < KS) rd(ks + 2, a0, b0);
#pragma unroll
                    for (int t2 = 0; t2 < TG; ++t2) acc[t2][0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[t2], acc[t2][0][0], 0, 0, 0);
                }
            }
        }
    }

    // ---- store the partial block: D[row = co = 4g+reg][col = ci = l15] ----
    float* slab = a.slab + (size_t)split * NTAPS * a.Cout * a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < TG; ++t2)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int ci = ci0 + wci + 16 * j + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wco + 16 * i + 4 * g + r;
                    if (co < a.Cout && ci < a.Cin) slab[((size_t)t2 * a.Cout + co) * a.Cin + ci] = acc[t2][i][j][r];
                }
            }
}

// Sum split-K slabs in split order and scatter to the framework layout:
//   out[o][i][tap] (= OIHW for Conv2d, (Cin,Cout,kh,kw) for ConvTranspose2d) = sum_s slab[s][tap][o][i]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                    int nsplit, int ntaps, int O, int I, int accumulate,
                                    int out_taps, int out_tap0) {
    const size_t total = (size_t)ntaps * O * I;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * total + e];
        const int i = (int)(e % I);
        const size_t r = e / I;
        const int o = (int)(r % O);
        const int tap = (int)(r / O);
        const size_t dst = ((size_t)o * I + i) * out_taps + out_tap0 + tap;
        out[dst] = accumulate ? out[dst] + s : s;
    }
}

// The same with a thread owning one (o, i) pair and all NT taps of the launch: NT independent coalesced loads per split, and
// its NT results leave as 4*NT contiguous bytes.  Small layers have few (o, i) pairs but many splits (64 x 64 weights:
// 4096 pairs, 256 splits = 16 workgroups walking 256 slabs each), so KL lanes share a pair: lane kl sums splits
// kl, kl + KL, ... in order, lane 0 then adds the KL partials in lane order - a fixed order, bitwise reproducible.
// (NT = 9: the 3x3 gradients; 3 / 5 / 10: the k x 1, 1 x k branches and the row pieces of a 5x5, models/reparm_unet.py:55-61.)
template <int KL, int NT>
__global__ __launch_bounds__(256) void wgrad_reduce9_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                            int nsplit, int O, int I, int accumulate, int out_taps, int out_tap0) {
    constexpr int EL = 256 / KL;                         // (o, i) pairs per workgroup
    __shared__ float part[KL > 1 ? KL : 1][EL][NT];
    const size_t OI = (size_t)O * I, total = NT * OI;
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    for (size_t e0 = (size_t)blockIdx.x * EL; e0 < OI; e0 += (size_t)gridDim.x * EL) {
        const size_t e = e0 + el;
        float s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) s[t] = 0.f;
        if (e < OI) {
            for (int k = kl; k < nsplit; k += KL) {
                const float* p = slab + (size_t)k * total + e;
#pragma unroll
                for (int t = 0; t < NT; ++t) s[t] += p[(size_t)t * OI];
            }
        }
        if (KL > 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) part[kl][el][t] = s[t];
            __syncthreads();
            if (kl == 0) {
                for (int q = 1; q < KL; ++q)
#pragma unroll
                    for (int t = 0; t < NT; ++t) s[t] += part[q][el][t];
            }
            __syncthreads();
        }
        if (kl == 0 && e < OI) {
            float* d = out + e * out_taps + out_tap0;
#pragma unroll
            for (int t = 0; t < NT; ++t) d[t] = accumulate ? d[t] + s[t] : s[t];
        }
    }
}

template <int NT>
static void launch_reduce_taps(const float* slab, float* dw, int nsplit, int Cout, int Cin, int accumulate, int dw_taps, int dw_tap0, hipStream_t s) {
    const size_t OI = (size_t)Cout * Cin;
    if (OI < 262144 && nsplit >= 16) {                // fewer than 1024 workgroups of pairs: share each pair among 8 lanes
        const int b = (int)((OI + 31) / 32);
        hipLaunchKernelGGL((wgrad_reduce9_kernel<8, NT>), dim3(b), dim3(256), 0, s, slab, dw, nsplit, Cout, Cin, accumulate, dw_taps, dw_tap0);
    } else {
        int b = (int)((OI + 255) / 256);
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL((wgrad_reduce9_kernel<1, NT>), dim3(b), dim3(256), 0, s, slab, dw, nsplit, Cout, Cin, accumulate, dw_taps, dw_tap0);
    }
}

template <typename T, int KH, int KW, int SH>
static int launch_wgrad(const WgradArgs& a, hipStream_t stream, int nsplit) {
    constexpr int TG = KH * KW;
    constexpr int HH = 7 * SH + KH, HWD = 31 + KW;
    constexpr int SMEM = 8 * 32 * 128 + HH * HWD * 128;
    static DynLdsOnce lds_once;
    if (int rc = ensure_dyn_lds(lds_once, (const void*)conv_wgrad_kernel<T, KH, KW, SH, TG>, SMEM)) return rc;
    constexpr int CT = WG<T>::CT;
    dim3 grid((unsigned)nsplit, (unsigned)((a.Cout + CT - 1) / CT), (unsigned)((a.Cin + CT - 1) / CT));
    hipLaunchKernelGGL((conv_wgrad_kernel<T, KH, KW, SH, TG>), grid, dim3(256), SMEM, stream, a);
    UNET_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int dispatch_wgrad(int kh, int kw, int sh, const WgradArgs& a, hipStream_t s, int nsplit) {
    if (sh == 2) {
        UNET_CHECK(kh == 2 && kw == 1, "stride-2 form is the 2x1 convT view only");
        return launch_wgrad<T, 2, 1, 2>(a, s, nsplit);
    }
    if (kh == 3 && kw == 3) return launch_wgrad<T, 3, 3, 1>(a, s, nsplit);
    if (kh == 1 && kw == 1) return launch_wgrad<T, 1, 1, 1>(a, s, nsplit);
    if (kh == 3 && kw == 1) return launch_wgrad<T, 3, 1, 1>(a, s, nsplit);
    if (kh == 1 && kw == 3) return launch_wgrad<T, 1, 3, 1>(a, s, nsplit);
    if (kh == 5 && kw == 1) return launch_wgrad<T, 5, 1, 1>(a, s, nsplit);
    if (kh == 1 && kw == 5) return launch_wgrad<T, 1, 5, 1>(a, s, nsplit);
    unet_set_error("unsupported wgrad kernel %dx%d stride %d", kh, kw, sh);
    return -1;
}

// Number of split-K slabs unet_conv_wgrad will write for this problem (the caller sizes the
// workspace as nsplit * kh*kw * Cout * Cin floats).
extern "C" int unet_conv_wgrad_splits(int dtype, int kh, int kw, int stride_h, int N, int Hout, int Wout, int Cin, int Cout) {
    if (dtype == UNET_BF16) return wgrad2_splits(kh, kw, stride_h, N, Hout, Wout, Cin, Cout);
    const int ct = 32;
    const int tiles = N * ((Hout + 7) / 8) * ((Wout + 31) / 32);
    const int blocks = ((Cout + ct - 1) / ct) * ((Cin + ct - 1) / ct);
    int want = (1024 + blocks - 1) / blocks;          // aim for ~4 workgroups per CU
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    const int per = (tiles + want - 1) / want;
    return (tiles + per - 1) / per;
}

extern "C" int unet_conv_wgrad(int dtype, int kh, int kw, int stride_h,
                               const void* dz, const void* x, float* slab, float* dw, int accumulate,
                               int dw_taps, int dw_tap0,
                               int N, int Hin, int Win, int Cin, int ldx, int xoff,
                               int Hout, int Wout, int Cout, int lddz, int dzoff,
                               int pad_h, int pad_w, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    UNET_CHECK(dz != nullptr && x != nullptr && slab != nullptr, "conv_wgrad: null pointer argument");      // dw may be NULL: partials stay in the slab
    UNET_CHECK(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && xoff % 8 == 0 && lddz % 8 == 0 && dzoff % 8 == 0,
               "channel counts/strides/offsets must be multiples of 8");
    UNET_CHECK(xoff + Cin <= ldx && dzoff + Cout <= lddz, "channel slice exceeds pixel stride");
    // pad_h/pad_w may be any integers: a 5x5 weight grad is issued row-wise (bf16: 2x5 + 2x5 + 1x5 launches, fp32: five 1x5)
    // with pad_h = 2 - first row, each filling taps [dw_tap0, dw_tap0 + kh*kw) of the dw_taps-tap gradient
    UNET_CHECK(kh * kw <= 10, "wgrad handles at most 10 taps per launch; split larger kernels by rows");
    UNET_CHECK(!(kh == 2 && kw == 5) || dtype == UNET_BF16, "the 2x5 row pair is a bf16 instantiation");
    UNET_CHECK(dw_tap0 >= 0 && dw_tap0 + kh * kw <= dw_taps, "tap window outside the gradient tensor");
    UNET_CHECK(N > 0 && Hout > 0 && Wout > 0, "empty wgrad problem");
    UNET_CHECK((size_t)Hin * Win * ldx < (1ull << 31) && (size_t)Hout * Wout * lddz < (1ull << 31), "image exceeds 2^31 elements");
    WgradArgs a;
    a.dz = dz; a.x = x; a.slab = slab; a.tq = nullptr;
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.ldx = ldx; a.xoff = xoff;
    a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.lddz = lddz; a.dzoff = dzoff;
    a.pad_h = pad_h; a.pad_w = pad_w;
    a.tiles_h = (Hout + 7) / 8; a.tiles_w = (Wout + 31) / 32;
    a.tiles_total = N * a.tiles_h * a.tiles_w;
    const int nsplit = unet_conv_wgrad_splits(dtype, kh, kw, stride_h, N, Hout, Wout, Cin, Cout);
    a.tiles_per_split = (a.tiles_total + nsplit - 1) / nsplit;
    hipStream_t s = (hipStream_t)stream;
    int rc = dtype == UNET_BF16 ? dispatch_wgrad2(kh, kw, stride_h, a, s, nsplit)
                                : dispatch_wgrad<float>(kh, kw, stride_h, a, s, nsplit);
    if (rc) return rc;
    if (dw == nullptr) return 0;    // caller reduces the slabs itself (ConvTranspose2d layout)
    const size_t total = (size_t)kh * kw * Cout * Cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    switch (kh * kw) {
        case 9: launch_reduce_taps<9>(slab, dw, nsplit, Cout, Cin, accumulate, dw_taps, dw_tap0, s); break;
        case 3: launch_reduce_taps<3>(slab, dw, nsplit, Cout, Cin, accumulate, dw_taps, dw_tap0, s); break;
        case 5: launch_reduce_taps<5>(slab, dw, nsplit, Cout, Cin, accumulate, dw_taps, dw_tap0, s); break;
        case 10: launch_reduce_taps<10>(slab, dw, nsplit, Cout, Cin, accumulate, dw_taps, dw_tap0, s); break;
        default: hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, slab, dw, nsplit, kh * kw, Cout, Cin, accumulate, dw_taps, dw_tap0);
    }
    UNET_LAUNCH_CHECK();
    return 0;
}
