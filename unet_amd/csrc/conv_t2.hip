// bf16 ConvTranspose2d(k=2, s=2) forward and input gradient, second generation (gfx950)
// (reference models/unet.py:35,53: the four decoder upsamplers).
//
// Both are one GEMM over the LOW-resolution pixel grid (h, w), out[px][n] = sum_k A[px][k] * Wt[n][k]:
//   forward : A = x[h, w, ci] (K = Ci); n = q*Co + co is written to y[2h+i, 2w+j, co] (+ bias[co]), q = 2i+j,
//             straight into the channel slice of the concat buffer (torch.cat of models/unet.py:59 is gone);
//   dgrad   : A = dy[2h+i, 2w+j, co] for K index (q, co) (K = 4*Co); n = ci is written to dx[h, w, ci].
// With 2..8 FLOP per byte of activations these launches are HBM-bound, so the kernel is built as a
// streaming pipeline: persistent 512-thread workgroups, one per CU, walk tiles of 16x32 pixels x 128
// outputs; each PHASE is one 32-channel K chunk whose operands (512 pixel records + 128 weight rows of
// 64 bytes) were written into one of THREE LDS stages by buffer_load ... lds two phases earlier; counted
// vmcnt waits let the younger requests fly across the barrier.  The epilogue transposes each 16-pixel
// group through a wave-private LDS tile and stores whole 128-byte lines.  Shapes that are not whole tiles
// stay on conv_igemm.hip.
//
// SPECIALISED REQUEST WAVES (as conv_wgrad2.hip): the workgroup has 12 waves, three per SIMD.  Waves 0-7 are the
// matrix waves (2 output halves x 4 row groups) and touch global memory only to store; waves 8-11 request every
// phase's 40 pieces (10 each), wait for them and meet the others at the phase barrier.  A phase is only 32 MFMAs per
// wave, so the 5 requests a matrix wave used to issue per phase (60-180 cycles each) cost about as much as its MFMAs.
#include "common.h"
#include "conv_args.h"
#include "lds_dma.h"

struct CT2Args {
    const void* src;      // forward: x (+xoff) [N,H,W,lds];  dgrad: dy [N,2H,2W,lds]
    const void* w;        // forward: [4*Co][Ci] (pack mode 2);  dgrad: [2][Ci][2*Co] (pack mode 3)
    const float* bias;    // forward only
    void* dst;            // forward: y [N,2H,2W,ldd] (+doff);  dgrad: dx [N,H,W,ldd] (+doff)
    int N, H, W;          // low-resolution grid
    int lds_, soff, ldd, doff;
    int Ci, Co;
    int tiles_h, tiles_w, ntile_n, total_tiles, nk;
    int nt;               // streaming stores
};

struct CT2Cfg {
    static constexpr int A_BYTES = 512 * 64, W_BYTES = 128 * 64, STAGE = A_BYTES + W_BYTES;
    static constexpr int OFF_BIAS = 3 * STAGE;
    static constexpr int OFF_SCR = OFF_BIAS + 2 * 1024;      // epilogue transposition tiles, 2 KiB per matrix wave
    static constexpr int SMEM = OFF_SCR + 8 * 2048;
    static constexpr int NREQ = 4;                           // request waves
};

__device__ __forceinline__ int ct2_swz(int idx) { return ((idx >> 2) & 1) << 1; }

template <bool DGRAD>
__global__ __launch_bounds__(768) void convT2_kernel(CT2Args a, FastDiv d_n, FastDiv d_tw, FastDiv d_th) {
    using C = CT2Cfg;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned sb = lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool requester = wave >= 8;
    const int Ws = DGRAD ? 2 * a.W : a.W, sy = DGRAD ? 2 : 1;
    const int kq = a.Co >> 5;                               // dgrad: K chunks per quadrant

    struct Tile { int nb, n, h0, w0; };
    auto decode = [&](int t) {
        Tile o;
        const int pt = (int)fdiv((unsigned)t, d_n); o.nb = t - pt * a.ntile_n;
        const int p1 = (int)fdiv((unsigned)pt, d_tw), tw_i = pt - p1 * a.tiles_w;
        o.n = (int)fdiv((unsigned)p1, d_th);
        o.h0 = (p1 - o.n * a.tiles_h) * 16; o.w0 = tw_i * 32;
        return o;
    };
    const int lblock = xcd_block_id();
    if (lblock >= a.total_tiles) return;

    if (requester) {
        // ---- request waves: piece p of a phase is 16 records (pixels, or weight rows); wave r takes p = r + 4i -------
        const int rw = wave - 8;
        const int dl = lane >> 2, ds = lane & 3;
        unsigned aoff[8], woff[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (rw + C::NREQ * i) * 16 + dl;     // pixel record 0..511 = row*32 + col
            const int hr = r >> 5, hc = r & 31;
            aoff[i] = (unsigned)((sy * hr * Ws + sy * hc) * a.lds_ + (ds ^ ct2_swz(hc)) * 8) * 2u;
        }
        const int wstride = DGRAD ? 2 * a.Co * 2 : a.Ci * 2;    // bytes between consecutive output rows of Wt
#pragma unroll
        for (int i = 0; i < 2; ++i)
            woff[i] = (unsigned)(((rw + C::NREQ * i) * 16 + dl) * wstride + (ds ^ ct2_swz(dl)) * 16);
        int q_tile = lblock, q_k = 0, q_seq = 0, q_stage = 0;
        bool q_valid = true;
        Tile q_t = decode(q_tile);
        int n_issued = 0, mark[3] = {0, 0, 0};
        auto push = [&]() {
            if (!q_valid) return;
            int qi = 0, qj = 0, c = q_k;
            if (DGRAD) { const int qq = q_k / kq; c = q_k - qq * kq; qi = qq >> 1; qj = qq & 1; }
            const long long pix = ((long long)(q_t.n * sy * a.H + sy * q_t.h0 + qi) * Ws + sy * q_t.w0 + qj);
            const i32x4_t ra = make_rsrc((const __bf16*)a.src + a.soff + pix * a.lds_ + c * 32, LDS_DMA_RECORDS);
            const long long wb = DGRAD ? ((long long)(qi * a.Ci + q_t.nb * 128) * (2 * a.Co) + qj * a.Co + c * 32)
                                       : ((long long)(q_t.nb * 128) * a.Ci + c * 32);
            const i32x4_t rwt = make_rsrc((const __bf16*)a.w + wb, LDS_DMA_RECORDS);
            const unsigned st = sb + q_stage * C::STAGE;
            if (!DGRAD && q_k == 0 && rw == 0) {
                // the tile's biases, ONE piece (a padded lane writes zeros, so two pieces must not overlap): lane L < 32
                // carries floats 4*(L & 15).. of output half L >> 4, whose 64 outputs are one quadrant's channels
                const int half = lane >> 4;
                const unsigned bo = lane < 32 ? (unsigned)(((q_t.nb * 128 + half * 64) % a.Co + (lane & 15) * 4) * 4) : LDS_DMA_OOB;
                lds_dma16_buf(make_rsrc(a.bias, (unsigned)a.Co * 4u), bo,
                              __builtin_amdgcn_readfirstlane(sb + C::OFF_BIAS + q_seq * 1024));
                ++n_issued;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                lds_dma16_buf(rwt, woff[i], __builtin_amdgcn_readfirstlane(st + C::A_BYTES + (rw + C::NREQ * i) * 1024));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                lds_dma16_buf(ra, aoff[i], __builtin_amdgcn_readfirstlane(st + (rw + C::NREQ * i) * 1024));
            n_issued += 10;
            if (q_stage == 0) mark[0] = n_issued; else if (q_stage == 1) mark[1] = n_issued; else mark[2] = n_issued;
            q_stage = q_stage == 2 ? 0 : q_stage + 1;
            if (++q_k == a.nk) {
                q_k = 0; q_tile += gridDim.x; q_seq ^= 1;
                q_valid = q_tile < a.total_tiles;
                if (q_valid) q_t = decode(q_tile);
            }
        };
        // all but `allowed` of this wave's youngest requests have landed (a push is 10 or 11 requests)
        auto wait_landed = [&](int need) {
            const int allowed = n_issued - need;
            if (allowed < 10) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (allowed == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        };
        push(); push(); push();
        wait_landed(mark[0]);
        __syncthreads();
        int s = 0;
        for (int tile_i = lblock; tile_i < a.total_tiles; tile_i += gridDim.x)
            for (int k = 0; k < a.nk; ++k) {
                const int sn = s == 2 ? 0 : s + 1;
                wait_landed(sn == 0 ? mark[0] : (sn == 1 ? mark[1] : mark[2]));      // the next chunk; younger ones keep flying
                __syncthreads();                                                     // ... and nobody reads stage s any more
                push();                                                              // refill it: three chunks ahead
                s = sn;
            }
        return;
    }

    // ---- matrix waves ------------------------------------------------------------------------------------
    const int g = lane >> 4, l15 = lane & 15;
    const int wc = wave >> 2, wr = wave & 3;                // 64 outputs x (4 rows x 32 pixels) per wave
    // fragment bases: A operand = weights (rows = outputs), B operand = pixels
    const int a_base = C::A_BYTES + (wc * 64 + l15) * 64 + ((g ^ ct2_swz(l15)) << 4);
    int b_base[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) b_base[cc] = ((wr * 4) * 32 + cc * 16 + l15) * 64 + ((g ^ ct2_swz(l15)) << 4);
    __syncthreads();                                        // the first chunk has landed

    int tile_i = lblock, tseq = 0, s = 0;
    Tile cur = decode(tile_i);
    for (;;) {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < a.nk; ++k) {
            const char* st = smem + s * C::STAGE;
            uint4 fa[4], fb[8];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fa[ct] = *(const uint4*)(st + a_base + ct * 1024);
#pragma unroll
            for (int pg = 0; pg < 8; ++pg) fb[pg] = *(const uint4*)(st + b_base[pg & 1] + (pg >> 1) * (32 * 64));
#pragma unroll
            for (int pg = 0; pg < 8; ++pg)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct][pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, fa[ct]), __builtin_bit_cast(bf16x8, fb[pg]), acc[ct][pg], 0, 0, 0);
            __syncthreads();                                // next chunk landed (request waves waited); stage s is free
            s = s == 2 ? 0 : s + 1;
        }
        {
            // ---- epilogue: + bias, bf16, transposition through this wave's 2 KiB LDS tile, whole 128-byte lines out ----
            char* scr = smem + C::OFF_SCR + wave * 2048;
            const int n_w = cur.nb * 128 + wc * 64;             // this wave's first output
            int qi = 0, qj = 0, chan = n_w;
            if (!DGRAD) { const int qq = n_w / a.Co; chan = n_w - qq * a.Co; qi = qq >> 1; qj = qq & 1; }
            const int osy = DGRAD ? 1 : 2, Wd = osy * a.W;
            const size_t pix0 = ((size_t)(cur.n * osy * a.H + osy * (cur.h0 + wr * 4) + qi)) * Wd + osy * cur.w0 + qj;
            char* ybase = (char*)a.dst + (pix0 * a.ldd + a.doff + chan) * 2;
            // lane-derived addresses from an opaque copy of the lane id (hipcc would otherwise hoist them out of the
            // tile loop and hold ~10 registers across the main loop), bias added into the accumulators once,
            // one v_cvt_pk_bf16_f32 per pair: the same epilogue as conv3x3_v2_kernel
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int ge = lane_e >> 4, l15e = lane_e & 15;
            const int rpx = lane_e >> 3, rch = lane_e & 7;
            const unsigned st_off = (unsigned)(osy * rpx * a.ldd + rch * 8) * 2u;
            const size_t row_stride = (size_t)osy * Wd * a.ldd * 2, half_stride = (size_t)osy * 8 * a.ldd * 2;
            if (!DGRAD) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const float4 b4 = *(const float4*)(smem + C::OFF_BIAS + tseq * 1024 + (wc * 64 + ct * 16 + 4 * ge) * 4);
#pragma unroll
                    for (int pg = 0; pg < 8; ++pg) {
                        acc[ct][pg][0] += b4.x; acc[ct][pg][1] += b4.y; acc[ct][pg][2] += b4.z; acc[ct][pg][3] += b4.w;
                    }
                }
            }
            char* wbase = scr + (l15e >> 3) * 1024 + (l15e & 7) * 128 + (ge & 1) * 8;
            const char* rbase = scr + rpx * 128 + ((rch ^ rpx) << 4);
#pragma unroll
            for (int pg = 0; pg < 8; ++pg) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    uint2 pk;
                    pk.x = cvt_pk_bf16(acc[ct][pg][0], acc[ct][pg][1]);
                    pk.y = cvt_pk_bf16(acc[ct][pg][2], acc[ct][pg][3]);
                    *(uint2*)(wbase + (((ct * 2 + (ge >> 1)) ^ (l15e & 7)) << 4)) = pk;
                }
                char* pp = ybase + (pg >> 1) * row_stride + (pg & 1) * 2 * half_stride + st_off;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint4 val = *(const uint4*)(rbase + h * 1024);
                    if (a.nt) {         // streaming stores for outputs far larger than the L2s (see conv_fwd2.hip)
                        typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(u32x4_nt{val.x, val.y, val.z, val.w}, (u32x4_nt*)(pp + h * half_stride));
                    } else {
                        *(uint4*)(pp + h * half_stride) = val;
                    }
                }
            }
        }
        tile_i += gridDim.x; tseq ^= 1;
        if (tile_i >= a.total_tiles) break;
        cur = decode(tile_i);
    }
}

template <bool DGRAD>
static int launch_convT2(CT2Args a, hipStream_t stream) {
    static DynLdsOnce lds_once;
    if (int rc = ensure_dyn_lds(lds_once, (const void*)convT2_kernel<DGRAD>, CT2Cfg::SMEM)) return rc;
    a.tiles_h = a.H / 16; a.tiles_w = a.W / 32;
    a.ntile_n = (DGRAD ? a.Ci : 4 * a.Co) / 128;
    a.total_tiles = a.N * a.tiles_h * a.tiles_w * a.ntile_n;
    a.nk = (DGRAD ? 4 * a.Co : a.Ci) / 32;
    a.nt = (size_t)a.N * a.H * a.W * (DGRAD ? a.Ci : 4 * a.Co) * 2 >= nt_threshold_bytes();
    const int mx = a.ntile_n > a.tiles_w ? (a.ntile_n > a.tiles_h ? a.ntile_n : a.tiles_h) : (a.tiles_w > a.tiles_h ? a.tiles_w : a.tiles_h);
    UNET_CHECK((unsigned long long)a.total_tiles * (unsigned)mx < (1ull << 32), "tile count too large for the index arithmetic");
    const int grid = a.total_tiles < 256 ? a.total_tiles : 256;
    hipLaunchKernelGGL((convT2_kernel<DGRAD>), dim3((unsigned)grid), dim3(768), CT2Cfg::SMEM, stream, a,
                       make_fastdiv((unsigned)a.ntile_n), make_fastdiv((unsigned)a.tiles_w), make_fastdiv((unsigned)a.tiles_h));
    UNET_LAUNCH_CHECK();
    return 0;
}

// Called from unet_conv_fwd (conv_igemm.hip) for the two ConvTranspose2d forms when the shapes are whole
// tiles: epi == 1 (1x1 conv + 2x2 shuffle) and the 2x1 stride-2 dgrad view.
bool convT2_fwd_ok(const ConvArgs& a) {      // a: x [N,H,W,Cin], Cout = 4*Co
    const int Co = a.Cout / 4;
    // Cin >= 96: with fewer than three K chunks per tile the bias piece of tile t+2 (same LDS slot as tile t's) would be
    // requested while tile t's epilogue may still be reading its own
    return (a.Cin & 31) == 0 && a.Cin >= 96 && (Co & 63) == 0 && (a.Hout & 15) == 0 && (a.Wout & 31) == 0 && a.flags == 0 && a.csplit == 0 &&
           (size_t)4 * a.Hout * a.Wout * a.ldy < (1ull << 31);
}
int convT2_fwd(const ConvArgs& a, hipStream_t s) {
    CT2Args t;
    t.src = a.x; t.w = a.w; t.bias = a.bias; t.dst = a.y;
    t.N = a.N; t.H = a.Hin; t.W = a.Win; t.lds_ = a.ldx; t.soff = a.xoff; t.ldd = a.ldy; t.doff = a.yoff;
    t.Ci = a.Cin; t.Co = a.Cout / 4;
    return launch_convT2<false>(t, s);
}
bool convT2_dgrad_ok(const ConvArgs& a) {    // a: view [N,2H,W,2*Co] dense, output [N,H,W,Ci]
    const int Co = a.Cin / 2;
    return a.ldx == a.Cin && a.xoff == 0 && (Co & 31) == 0 && (a.Cout & 127) == 0 && (a.Hout & 15) == 0 && (a.Wout & 31) == 0 &&
           a.flags == 0 && a.csplit == 0 && a.bias == nullptr && a.stats == nullptr && a.pad_h == 0 && a.pad_w == 0;
}
int convT2_dgrad(const ConvArgs& a, hipStream_t s) {
    CT2Args t;
    t.src = a.x; t.w = a.w; t.bias = nullptr; t.dst = a.y;
    t.N = a.N; t.H = a.Hout; t.W = a.Wout; t.lds_ = a.Cin / 2; t.soff = 0; t.ldd = a.ldy; t.doff = a.yoff;
    t.Ci = a.Cout; t.Co = a.Cin / 2;
    return launch_convT2<true>(t, s);
}
