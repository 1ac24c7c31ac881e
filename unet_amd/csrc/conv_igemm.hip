// Implicit-GEMM convolution for gfx950: NHWC activations, [tap][Cout][Cin] packed weights.
//
// One kernel family covers every MFMA-shaped convolution of the U-Net hot path:
//   3x3 s1 p1          forward and dgrad of nn.Conv2d(k=3,pad=1)      (reference models/unet.py:10,13)
//   1x1 + 2x2 shuffle  forward of nn.ConvTranspose2d(k=2,s=2)         (reference models/unet.py:35,53)
//   2x1 stride(2,1)    dgrad of that ConvTranspose2d, on the [N,2H,W,2C] view of its output grad
//   kx1 / 1xk / kxk    RepUNet branches and the deployed 5x5 conv     (reference models/reparm_unet.py:55-61,139)
//
// Tiling (per 256-thread workgroup = 4 waves): 8x32 output pixels x 64 output channels.
// GEMM orientation: A = weights (rows = Cout), B = activations (cols = pixels), so each lane ends
// up with 4 consecutive output channels of one pixel (one 8-byte bf16 store).
// The input halo tile ((8-1)*SH+KH) x (31+KW) pixels x 64 bytes of channels is staged in LDS once
// per channel chunk and re-used by all KH*KW taps; weights are staged per tap group.  Both go
// global -> registers -> LDS (zero padding needs a predicate per pixel), with the next stage's
// global loads issued before the current stage's MFMAs (issue-early / write-late).
#include "common.h"

#include "conv_args.h"
#include <cstdlib>

template <typename T, int KH, int KW, int SH, int TG, int EPI>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int TH = 8, TW = 32, BN = 64;
    constexpr int HH = (TH - 1) * SH + KH, HWD = TW - 1 + KW, HWDP = (HWD + 7) & ~7;
    constexpr int EPV = TT<T>::EPV, CKE = TT<T>::CKE;
    constexpr int NTAPS = KH * KW, NTG = NTAPS / TG;
    static_assert(NTAPS % TG == 0, "tap groups must tile the kernel");
    constexpr int HUNITS = HH * HWD * 4, NHU = (HUNITS + 255) / 256;
    constexpr int HALO_BYTES = HH * HWDP * 64;
    constexpr int WTAP_BYTES = BN * 64;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_halo = smem;
    char* s_w = smem + HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;

    int t = blockIdx.x;
    const int tw_i = t % a.tiles_w; t /= a.tiles_w;
    const int th_i = t % a.tiles_h;
    const int n = t / a.tiles_h;
    const int h0 = th_i * TH, w0 = tw_i * TW;
    const int co0 = blockIdx.y * BN;

    const T* __restrict__ xin = (const T*)a.x + (size_t)n * a.Hin * a.Win * a.ldx + a.xoff;
    const T* __restrict__ wp = (const T*)a.w;

    // ---- per-thread staging plan -------------------------------------------------------------
    int hoff[NHU], hlds[NHU];
    unsigned hmask = 0;
#pragma unroll
    for (int i = 0; i < NHU; ++i) {
        const int u = tid + i * 256;
        const int P = u >> 2, q = u & 3;
        const int hr = P / HWD, hc = P - hr * HWD;
        const int row = h0 * SH + hr - a.pad_h, col = w0 + hc - a.pad_w;
        const bool ok = (u < HUNITS) && row >= 0 && row < a.Hin && col >= 0 && col < a.Win;
        hoff[i] = ok ? ((row * a.Win + col) * a.ldx + q * EPV) : 0;
        hlds[i] = (u < HUNITS) ? rec64_off(hr * HWDP + hc, q) : -1;
        if (ok) hmask |= 1u << i;
    }
    const int wrow = tid >> 2, wq = tid & 3;
    const bool wrow_ok = (co0 + wrow) < a.Cout;
    const int wlds = rec64_off(wrow, wq);
    const int wbase = (co0 + wrow) * a.Cin + wq * EPV;   // + tap*Cout*Cin + chunk*CKE

    // ---- per-lane fragment read bases ----------------------------------------------------------
    const int a_base = rec64_off(l15, g);                         // + ct*1024 + j*WTAP_BYTES
    int b_base[KW];
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
        const int col = kw + l15;
        b_base[kw] = (wave * 2 * SH * HWDP + col) * 64 + ((g ^ (((col >> 2) & 1) << 1)) << 4);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 hreg[NHU], wreg[TG];
    const int nchunks = (a.Cin + CKE - 1) / CKE;

    auto load_halo = [&](int c) {
        const int cbase = c * CKE;
#pragma unroll
        for (int i = 0; i < NHU; ++i) {
            const int q = (tid + i * 256) & 3;
            const bool ok = ((hmask >> i) & 1u) && (cbase + q * EPV < a.Cin);
            hreg[i] = ok ? *(const uint4*)(xin + hoff[i] + cbase) : make_uint4(0, 0, 0, 0);
        }
    };
    auto load_w = [&](int c, int tg) {
        const int cbase = c * CKE;
        const bool ok = wrow_ok && (cbase + wq * EPV < a.Cin);
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int tap = tg * TG + j;
            wreg[j] = ok ? *(const uint4*)(wp + (size_t)tap * a.Cout * a.Cin + wbase + cbase) : make_uint4(0, 0, 0, 0);
        }
    };

    load_halo(0);
    load_w(0, 0);

    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) {
            __syncthreads();     // every wave has finished reading the buffers being overwritten
            if (tg == 0) {
#pragma unroll
                for (int i = 0; i < NHU; ++i)
                    if (hlds[i] >= 0) *(uint4*)(s_halo + hlds[i]) = hreg[i];
            }
#pragma unroll
            for (int j = 0; j < TG; ++j) *(uint4*)(s_w + j * WTAP_BYTES + wlds) = wreg[j];
            __syncthreads();
            // issue the next stage's global loads; they land while the MFMAs below run
            if (tg + 1 < NTG) {
                load_w(c, tg + 1);
            } else if (c + 1 < nchunks) {
                load_w(c + 1, 0);
                load_halo(c + 1);
            }
#pragma unroll
            for (int j = 0; j < TG; ++j) {
                const int tap = tg * TG + j;
                const int kh = tap / KW, kw = tap % KW;
                uint4 af[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) af[ct] = *(const uint4*)(s_w + j * WTAP_BYTES + a_base + ct * 1024);
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const uint4 bf = *(const uint4*)(s_halo + b_base[kw] + ((rr * SH + kh) * HWDP + cc * 16) * 64);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) mma16<T>(acc[ct][rr * 2 + cc], af[ct], bf);
                    }
            }
        }
    }

    // ---- epilogue: + bias, per-channel sum / sum-of-squares partials, store ---------------------
    float s1[4][4], s2[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[ct][r] = s2[ct][r] = 0.f;

    T* __restrict__ yout = (T*)a.y;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int co = co0 + ct * 16 + 4 * g;
        const bool co_ok = co < a.Cout;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        int cq = 0, qi = 0, cout_real = a.Cout;
        if (EPI == 1) {
            cout_real = a.Cout >> 2;
            qi = co_ok ? co / cout_real : 0;
            cq = co - qi * cout_real;
        } else {
            cq = co;
        }
        if (a.bias != nullptr && co_ok) {
            const float4 bv = *(const float4*)(a.bias + cq);
            b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int ho = h0 + wave * 2 + rr, wo = w0 + cc * 16 + l15;
                const bool ok = co_ok && ho < a.Hout && wo < a.Wout;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[ct][rr * 2 + cc][r] + b4[r];
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s1[ct][r] += v[r]; s2[ct][r] += v[r] * v[r]; }
                    size_t idx;
                    if (EPI == 1)
                        idx = (((size_t)n * 2 * a.Hout + 2 * ho + (qi >> 1)) * (2 * a.Wout) + 2 * wo + (qi & 1)) * a.ldy + a.yoff + cq;
                    else
                        idx = (((size_t)n * a.Hout + ho) * a.Wout + wo) * a.ldy + a.yoff + cq;
                    T* dst = yout + idx;
                    if (EPI == 0 && a.csplit > 0 && cq >= a.csplit)
                        dst = (T*)a.y2 + (((size_t)n * a.Hout + ho) * a.Wout + wo) * a.ldy2 + a.yoff2 + (cq - a.csplit);
                    if (a.flags & 1) {
                        if (sizeof(T) == 2) {
                            const uint2 old = *(const uint2*)dst;
                            v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                            v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
                        } else {
                            const float4 old = *(const float4*)dst;
                            v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
                        }
                    }
                    if (a.flags & 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (sizeof(T) == 2) {
                        uint2 pk;
                        pk.x = Vec16<__bf16>::pk(v[0], v[1]);
                        pk.y = Vec16<__bf16>::pk(v[2], v[3]);
                        *(uint2*)dst = pk;
                    } else {
                        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
    }

    if (EPI == 0 && a.stats != nullptr) {
        __syncthreads();                       // all waves are done with the staging buffers
        float* red = (float*)smem;             // [wave][64 co][2]
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = wave_sum16(s1[ct][r]);
                const float t2 = wave_sum16(s2[ct][r]);
                if (l15 == 0) {
                    red[(wave * 64 + ct * 16 + 4 * g + r) * 2 + 0] = t1;
                    red[(wave * 64 + ct * 16 + 4 * g + r) * 2 + 1] = t2;
                }
            }
        __syncthreads();
        if (tid < 128) {
            const int ch = tid >> 1, k = tid & 1;
            const float s = red[(0 * 64 + ch) * 2 + k] + red[(1 * 64 + ch) * 2 + k] +
                            red[(2 * 64 + ch) * 2 + k] + red[(3 * 64 + ch) * 2 + k];
            if (co0 + ch < a.Cout) a.stats[((size_t)blockIdx.x * a.Cout + co0 + ch) * 2 + k] = s;
        }
    }
}

template <typename T, int KH, int KW, int SH, int TG, int EPI>
static int launch_conv(const ConvArgs& a, hipStream_t stream) {
    constexpr int TH = 8, TW = 32;
    constexpr int HH = (TH - 1) * SH + KH, HWD = TW - 1 + KW, HWDP = (HWD + 7) & ~7;
    constexpr int SMEM = HH * HWDP * 64 + TG * 64 * 64;
    static_assert(SMEM <= 65536, "dynamic LDS above 64 KiB needs hipFuncSetAttribute");
    dim3 grid((unsigned)(a.N * a.tiles_h * a.tiles_w), (unsigned)((a.Cout + 63) / 64));
    hipLaunchKernelGGL((conv_igemm_kernel<T, KH, KW, SH, TG, EPI>), grid, dim3(256), SMEM, stream, a);
    UNET_LAUNCH_CHECK();
    return 0;
}

// bf16 3x3 stride-1 path (conv_fwd2.hip)
int conv2_rows(int N, int Hout, int Wout, int Cin, int Cout, int flags, int csplit);
int dispatch_conv2(const ConvArgs& a, hipStream_t s);
bool conv2_bnred_ok(int Cin, int Cout, int Hout, int Wout);
int dispatch_conv2_bnred(const ConvArgs& a, hipStream_t s);
// bf16 5x5 stride-1 path on whole tiles (conv_fwd2.hip)
bool conv5_v2_ok(int Cin, int Cout, int Hout, int Wout, int flags, int csplit);
int conv5_rows(int N, int Hout, int Wout);
int dispatch_conv5(const ConvArgs& a, hipStream_t s);
int dispatch_convk(int kh, int kw, const ConvArgs& a, hipStream_t s);
static bool use_conv5(int dtype, int kh, int kw, int sh, int epi, int Cin, int Cout, int Hout, int Wout, int flags, int csplit) {
    const bool shape = (kh == 5 && kw == 5) || ((kh == 3 || kh == 5) && kw == 1) || (kh == 1 && (kw == 3 || kw == 5));
    return dtype == UNET_BF16 && shape && sh == 1 && epi == 0 && conv5_v2_ok(Cin, Cout, Hout, Wout, flags, csplit);
}
// bf16 ConvTranspose2d forward / dgrad on whole tiles (conv_t2.hip)
bool convT2_fwd_ok(const ConvArgs& a);
int convT2_fwd(const ConvArgs& a, hipStream_t s);
bool convT2_dgrad_ok(const ConvArgs& a);
int convT2_dgrad(const ConvArgs& a, hipStream_t s);
static bool use_conv2(int dtype, int kh, int kw, int sh, int epi) {
    return dtype == UNET_BF16 && kh == 3 && kw == 3 && sh == 1 && epi == 0;
}

template <typename T>
static int dispatch_conv(int kh, int kw, int sh, int epi, const ConvArgs& a, hipStream_t s) {
    if (epi == 1) {
        UNET_CHECK(kh == 1 && kw == 1 && sh == 1, "shuffle epilogue is 1x1 only");
        return launch_conv<T, 1, 1, 1, 1, 1>(a, s);
    }
    if (sh == 2) {
        UNET_CHECK(kh == 2 && kw == 1, "stride-2 form is the 2x1 convT-dgrad view only");
        return launch_conv<T, 2, 1, 2, 2, 0>(a, s);
    }
    if (kh == 3 && kw == 3) return launch_conv<T, 3, 3, 1, 9, 0>(a, s);
    if (kh == 1 && kw == 1) return launch_conv<T, 1, 1, 1, 1, 0>(a, s);
    if (kh == 5 && kw == 5) return launch_conv<T, 5, 5, 1, 5, 0>(a, s);
    if (kh == 3 && kw == 1) return launch_conv<T, 3, 1, 1, 3, 0>(a, s);
    if (kh == 1 && kw == 3) return launch_conv<T, 1, 3, 1, 3, 0>(a, s);
    if (kh == 5 && kw == 1) return launch_conv<T, 5, 1, 1, 5, 0>(a, s);
    if (kh == 1 && kw == 5) return launch_conv<T, 1, 5, 1, 5, 0>(a, s);
    unet_set_error("unsupported conv kernel %dx%d stride %d", kh, kw, sh);
    return -1;
}

// stats_count() tells the caller how many partial rows unet_conv_fwd writes: one per pixel tile.
extern "C" int unet_conv_tiles(int dtype, int kh, int kw, int stride_h, int epi, int N, int Hout, int Wout,
                               int Cin, int Cout, int flags, int csplit) {
    if (use_conv2(dtype, kh, kw, stride_h, epi)) return conv2_rows(N, Hout, Wout, Cin, Cout, flags, csplit);
    if (use_conv5(dtype, kh, kw, stride_h, epi, Cin, Cout, Hout, Wout, flags, csplit)) return conv5_rows(N, Hout, Wout);
    return N * ((Hout + 7) / 8) * ((Wout + 31) / 32);
}

extern "C" int unet_conv_fwd(int dtype, int kh, int kw, int stride_h, int epi,
                             const void* x, const void* w, const float* bias, void* y, float* stats,
                             int N, int Hin, int Win, int Cin, int ldx, int xoff,
                             int Hout, int Wout, int Cout, int ldy, int yoff,
                             int pad_h, int pad_w,
                             void* y2, int ldy2, int yoff2, int csplit, int flags, void* stream) {
    UNET_CHECK(dtype == UNET_F32 || dtype == UNET_BF16, "bad dtype %d", dtype);
    if (csplit > 0) UNET_CHECK(y2 != nullptr && csplit % 8 == 0 && ldy2 % 8 == 0 && yoff2 % 8 == 0 && epi == 0 && csplit < Cout,
                               "bad split-output arguments");
    UNET_CHECK(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "empty conv problem");
    UNET_CHECK(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && xoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0,
               "channel counts/strides/offsets must be multiples of 8 (Cin=%d Cout=%d ldx=%d xoff=%d ldy=%d yoff=%d)",
               Cin, Cout, ldx, xoff, ldy, yoff);
    UNET_CHECK(xoff + Cin <= ldx, "input channel slice exceeds pixel stride");
    UNET_CHECK((size_t)Hin * Win * ldx < (1ull << 31), "one input image exceeds 2^31 elements");
    if (epi == 1) UNET_CHECK(Cout % 32 == 0, "shuffle epilogue needs Cout = 4*C with C %% 8 == 0");
    // the output grid the tiles cover must be what the kernel geometry produces
    const int eh = (Hin + 2 * pad_h - kh) / stride_h + 1, ew = Win + 2 * pad_w - kw + 1;
    UNET_CHECK(eh == Hout && ew == Wout, "output %dx%d does not match conv geometry %dx%d", Hout, Wout, eh, ew);
    UNET_CHECK(x != nullptr && w != nullptr && y != nullptr, "conv_fwd: null pointer argument");
    // statistics are those of the conv's own result; an accumulating launch (flags bit 0, the summed branch dgrads) has none to offer
    UNET_CHECK((flags & 1) == 0 || stats == nullptr, "conv_fwd: statistics and accumulation (flags bit 0) exclude each other");
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stats;
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.ldx = ldx; a.xoff = xoff;
    a.Hout = Hout; a.Wout = Wout; a.Cout = Cout; a.ldy = ldy; a.yoff = yoff;
    a.pad_h = pad_h; a.pad_w = pad_w;
    a.tiles_h = (Hout + 7) / 8; a.tiles_w = (Wout + 31) / 32;
    a.y2 = y2; a.ldy2 = ldy2; a.yoff2 = yoff2; a.csplit = csplit; a.flags = flags; a.tq = nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (use_conv2(dtype, kh, kw, stride_h, epi)) {
        UNET_CHECK((size_t)9 * Cout * Cin * 2 < (1ull << 31), "packed weights exceed 2^31 bytes");
        return dispatch_conv2(a, s);
    }
    if (use_conv5(dtype, kh, kw, stride_h, epi, Cin, Cout, Hout, Wout, flags, csplit)) {
        UNET_CHECK((size_t)25 * Cout * Cin * 2 < (1ull << 31), "packed weights exceed 2^31 bytes");
        return (kh == 5 && kw == 5) ? dispatch_conv5(a, s) : dispatch_convk(kh, kw, a, s);
    }
    if (dtype == UNET_BF16 && epi == 1 && kh == 1 && kw == 1 && stride_h == 1 && convT2_fwd_ok(a)) return convT2_fwd(a, s);
    if (dtype == UNET_BF16 && epi == 0 && kh == 2 && kw == 1 && stride_h == 2 && convT2_dgrad_ok(a)) return convT2_dgrad(a, s);
    return dtype == UNET_BF16 ? dispatch_conv<__bf16>(kh, kw, stride_h, epi, a, s)
                              : dispatch_conv<float>(kh, kw, stride_h, epi, a, s);
}

// 3x3 stride-1 dgrad of the SECOND conv of a DoubleConv (models/unet.py:13 under autograd) with the first pass of the BatchNorm
// backward in front of it (models/unet.py:11-12) riding on the epilogue: dy = conv3x3(dz, flipped weights) is written as usual,
// and sum_rows [N*(H/16)*(W/32)][Cout][2] receive, per 16x32 pixel tile, sum dy' and sum dy'*xhat with dy' = dy where
// z*scale + shift > 0 and xhat = (z - mean)*invstd -- the two sums unet_bn_relu_bwd's first pass takes from dy and z in a pass of
// its own.  unet_bn_relu_bwd_rows consumes the rows.  bf16, whole 16x32 tiles and 64-channel blocks only: ask _ok first.
extern "C" int unet_conv_dgrad_bnred_ok(int dtype, int H, int W, int Cin, int Cout) {
    return dtype == UNET_BF16 && conv2_bnred_ok(Cin, Cout, H, W) ? 1 : 0;
}
extern "C" int unet_conv_dgrad_bnred(int dtype, const void* dz, int ldx, int xoff, const void* w_dgrad, void* dy, int ldy, int yoff,
                                     float* sum_rows, const void* z, int ldz, int zoff,
                                     const float* scale, const float* shift, const float* mean, const float* invstd,
                                     int N, int H, int W, int Cin, int Cout, void* stream) {
    UNET_CHECK(unet_conv_dgrad_bnred_ok(dtype, H, W, Cin, Cout), "conv_dgrad_bnred: bf16, H %% 16 == 0, W %% 32 == 0, Cin %% 32 == 0, Cout %% 64 == 0 "
               "only (got %dx%d, %d -> %d channels)", H, W, Cin, Cout);
    UNET_CHECK(dz && w_dgrad && dy && sum_rows && z && scale && shift && mean && invstd, "conv_dgrad_bnred: null argument");
    UNET_CHECK(N > 0 && ldx % 8 == 0 && xoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0 && ldz % 4 == 0 && zoff % 4 == 0 &&
               xoff + Cin <= ldx && yoff + Cout <= ldy && zoff + Cout <= ldz, "conv_dgrad_bnred: bad strides / offsets");
    UNET_CHECK((size_t)H * W * ldx < (1ull << 31) && (size_t)9 * Cout * Cin * 2 < (1ull << 31), "conv_dgrad_bnred: tensor too large");
    ConvArgs a;
    a.x = dz; a.w = w_dgrad; a.bias = nullptr; a.y = dy; a.stats = sum_rows;
    a.N = N; a.Hin = H; a.Win = W; a.Cin = Cin; a.ldx = ldx; a.xoff = xoff;
    a.Hout = H; a.Wout = W; a.Cout = Cout; a.ldy = ldy; a.yoff = yoff;
    a.pad_h = 1; a.pad_w = 1; a.tiles_h = H / 16; a.tiles_w = W / 32;
    a.y2 = nullptr; a.ldy2 = 0; a.yoff2 = 0; a.csplit = 0; a.flags = 0; a.tq = nullptr;
    a.bn_z = z; a.bn_ldz = ldz; a.bn_zoff = zoff; a.bn_scale = scale; a.bn_shift = shift; a.bn_mean = mean; a.bn_invstd = invstd;
    return dispatch_conv2_bnred(a, (hipStream_t)stream);
}
