// bf16 weight gradient, second generation (gfx950): LDS-DMA double-buffered split-K GEMM.
//
//   dW[tap][co][ci] = sum over output pixels m of  dz[m][co] * x[SH*m + tap][ci]
//
// Same math and slab layout as conv_wgrad.hip (which keeps the exact-f32 path); what changes is how
// the operands reach the matrix cores:
//   * one 512-thread workgroup per CU: 8 waves as 2 (co) x 4 (ci), each owning a (CT/2) x (CT/4) x taps
//     block of accumulators (72 registers for 3x3), so each SIMD holds two waves that hide each
//     other's LDS latency and there is room to request the next tap's fragments ahead of the MFMAs.
//   * the dz tile and the x halo tile are written straight into LDS by global_load_lds_dwordx4
//     (no staging registers, no ds_write), into one of two stages: tile t+1 is in flight while
//     tile t is multiplied; one barrier per tile.
//   * zero padding (image border, ragged tiles, channel tails): the lane's SOURCE address is pointed
//     at a 16-byte block of zeros, the LDS destination being lane-linear.
//   * LDS images are pixel-major records of CT channels; the bank swizzle is a function of the pixel
//     COLUMN only and is applied to the source address (DMA side) and to the transposing reads, so
//     every fragment address is a per-lane base plus a compile-time immediate.
//   * workgroups are ordered channel-block fastest, so the workgroups resident at one time read the
//     same pixel range and share it through L2 / Infinity Cache.
#include "common.h"
#include "wgrad_args.h"
#include "lds_dma.h"
#include "tile_queue.h"

__device__ __attribute__((aligned(64))) uint4 g_wgrad_zero[4];   // source of every padded 16-byte piece

// REC = bytes per pixel record (128: 64 channels, 256: 128 channels).  XOR, in bytes, of the record's
// 32-byte pieces: keeps the eight pieces that one 32-lane half of ds_read_b64_tr_b16 touches (pixel
// columns c..c+3 and c+8..c+11, same channel bytes) on distinct banks.
template <int REC> __device__ __forceinline__ int swz_bytes(int col) {
    if constexpr (REC == 128) return (((col >> 1) & 1) << 6) | (((col >> 3) & 1) << 5);
    else return ((col & 3) << 5) | (((col >> 3) & 1) << 7);
}

__device__ __forceinline__ uint4 tr_frag2(const char* lo_addr, const char* hi_addr) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_addr);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_addr);
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

template <int KH, int KW, int SH, int WT, int TH>
struct Wg2Cfg {
    static constexpr int TW = 32;
    static constexpr int CT = 32 * WT;                 // channels per block side
    static constexpr int REC = CT * 2;                 // bytes per pixel record
    static constexpr int SLOTS = REC / 16;
    static constexpr int RPP = 1024 / REC;             // records per DMA piece (one wave instruction = 1 KiB)
    static constexpr int HH = (TH - 1) * SH + KH, HWD = TW - 1 + KW;
    static constexpr int DZ_REC = TH * TW, X_REC = HH * HWD;
    static constexpr int DZ_PIECES = DZ_REC / RPP, X_PIECES = (X_REC + RPP - 1) / RPP;
    static constexpr int DZ_BYTES = DZ_PIECES * 1024, X_BYTES = X_PIECES * 1024;
    static constexpr int STAGE = DZ_BYTES + X_BYTES;
    static constexpr int NT = KH * KW;
    static constexpr int WM = WT, WN = WT / 2;         // wave tile in 16-channel units: (16*WM) co x (16*WN) ci
    static constexpr int SMEM = 2 * STAGE;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(WT % 2 == 0, "wave grid is 2 (co) x 4 (ci)");
    static_assert((HWD * REC) % 256 == 0, "row pitch must keep the bank phase");
};

// GEN = false: channel counts multiples of the block and the output a whole number of tiles (every U-Net
// layer): the DMA is one buffer_load ... lds per piece with a wave-uniform descriptor (tile origin) and
// per-lane offsets fixed for the kernel's life; image borders use the out-of-range offset (hardware
// writes zeros).  GEN = true: every bound checked per lane, zero block as the padded source.
// SPEC = true (fast path only): SPECIALISED REQUEST WAVES.  The workgroup has 12 waves, three per SIMD: waves 0-7 are the
// 2 x 4 grid of matrix waves and never touch global memory until they store; waves 8-11 do nothing but request tiles
// (a quarter of the pieces each), wait for them and meet the others at the tile barrier.  A request costs its wave
// 60-180 cycles of issue (it waits for the memory pipeline to take the one before); spent by a matrix wave, as in the
// 8-wave form, those cycles are MFMA cycles (the loop without its requests ran 1.5x faster); spent by a wave that has
// nothing else to do they are free.  Needs the kernel to fit 168 registers (three waves per SIMD).
// DYN = true (request-wave form only): the pixel tiles come from atomic queues, one per XCD and channel block (tile_queue.h).
// The workgroups of an XCD that share a channel block draw consecutive tiles of that XCD's range, so the channel blocks of
// an XCD still walk the same pixels at about the same time (shared through its L2, as in the static order); a dry queue
// steals from the fullest queue of the same channel block.  Request wave 8 draws the tile after next while the next one's
// DMA is in flight; the ids travel through four LDS words.
template <int KH, int KW, int SH, int WT, int TH, bool GEN, bool SPEC = false, bool DYN = false>
__global__ __launch_bounds__(SPEC ? 768 : 512) void conv_wgrad2_kernel(WgradArgs a, int nchan_ci, int nchan, FastDiv d_tw, FastDiv d_th) {
    static_assert(!SPEC || !GEN, "request waves use the fast DMA plan");
    static_assert(!DYN || SPEC, "tile queues are drawn by a request wave");
    using C = Wg2Cfg<KH, KW, SH, WT, TH>;
    constexpr int REC = C::REC, SLOTS = C::SLOTS, RPP = C::RPP, HWD = C::HWD, NT = C::NT, CT = C::CT;

    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    constexpr int WM = C::WM, WN = C::WN;

    // channel block fastest: resident workgroups share the pixel range
    const int lblock = xcd_block_id();
    const int chan = lblock % nchan, split = lblock / nchan;
    const int co0 = (chan / nchan_ci) * CT, ci0 = (chan % nchan_ci) * CT;
    const int wco = ((wave >> 2) & 1) * (16 * WM), wci = (wave & 3) * (16 * WN);
    constexpr int NREQ = SPEC ? 4 : 8;                       // waves that request tiles
    const bool requester = !SPEC || wave >= 8;
    const int rw = SPEC ? (wave & 3) : wave;                 // index among the requesting waves

    f32x4 acc[NT][WM][WN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment bases (bytes inside the dz / x image); rows add compile-time immediates
    int a_lo[WM], a_hi[WM], b_lo[KW][WN], b_hi[KW][WN];
    {
        const int q = l15 >> 2, p = l15 & 3;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int col = 8 * g + q, chb = (wco + 16 * i) * 2;
            a_lo[i] = col * REC + (chb ^ swz_bytes<REC>(col)) + 8 * p;
            a_hi[i] = (col + 4) * REC + (chb ^ swz_bytes<REC>(col + 4)) + 8 * p;
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = kw + 8 * g + q, chb = (wci + 16 * j) * 2;
                b_lo[kw][j] = C::DZ_BYTES + col * REC + (chb ^ swz_bytes<REC>(col)) + 8 * p;
                b_hi[kw][j] = C::DZ_BYTES + (col + 4) * REC + (chb ^ swz_bytes<REC>(col + 4)) + 8 * p;
            }
    }

    const __bf16* __restrict__ dzp = (const __bf16*)a.dz + a.dzoff + co0;
    const __bf16* __restrict__ xp = (const __bf16*)a.x + a.xoff + ci0;
    const char* zero = (const char*)g_wgrad_zero;
    const unsigned smem_base = lds_addr(smem);

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(t_begin + a.tiles_per_split, a.tiles_total);

    // ---- fast DMA plan -------------------------------------------------------------------------------
    constexpr int DZ_PW = (C::DZ_PIECES + NREQ - 1) / NREQ, X_PW = (C::X_PIECES + NREQ - 1) / NREQ;
    static_assert(X_PW <= 32, "piece mask is one word");
    unsigned dz_l[DZ_PW], x_l[X_PW];
    if (!GEN && requester) {
        const int rl = lane / SLOTS, s = lane % SLOTS;
#pragma unroll
        for (int i = 0; i < DZ_PW; ++i) {
            const int R = (rw + NREQ * i) * RPP + rl;
            const int r = R >> 5, c = R & 31;
            dz_l[i] = (unsigned)((r * a.Wout + c) * a.lddz + (s ^ (swz_bytes<REC>(c) >> 4)) * 8) * 2u;
        }
#pragma unroll
        for (int i = 0; i < X_PW; ++i) {
            const int R = (rw + NREQ * i) * RPP + rl;
            const int hr = R / HWD, hc = R - hr * HWD;
            x_l[i] = R < C::X_REC ? (unsigned)((hr * a.Win + hc) * a.ldx + (s ^ (swz_bytes<REC>(hc) >> 4)) * 8) * 2u : LDS_DMA_OOB;
        }
    }
    auto issue_fast = [&](int tile, int stage_off) {
        const int p1 = (int)fdiv((unsigned)tile, d_tw), tw_i = tile - p1 * a.tiles_w;
        const int n = (int)fdiv((unsigned)p1, d_th), th_i = p1 - n * a.tiles_h;
        const int h0 = th_i * TH, w0 = tw_i * C::TW;
        const int row0 = h0 * SH - a.pad_h, col0 = w0 - a.pad_w;
        const i32x4_t rdz = make_rsrc(dzp + ((long long)(n * a.Hout + h0) * a.Wout + w0) * a.lddz, LDS_DMA_RECORDS);
        const i32x4_t rx = make_rsrc(xp + ((long long)(n * a.Hin + row0) * a.Win + col0) * a.ldx, LDS_DMA_RECORDS);
        unsigned mask = ~0u;
        if (!(row0 >= 0 && row0 + C::HH <= a.Hin && col0 >= 0 && col0 + HWD <= a.Win)) {      // border tile (wave-uniform)
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            mask = 0;
#pragma unroll
            for (int i = 0; i < X_PW; ++i) {
                const int R = (rw + NREQ * i) * RPP + lane_ / SLOTS;
                const int hr = R / HWD, hc = R - hr * HWD;
                if ((unsigned)(row0 + hr) < (unsigned)a.Hin && (unsigned)(col0 + hc) < (unsigned)a.Win) mask |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < DZ_PW; ++i) {
            const int piece = rw + NREQ * i;
            if (piece < C::DZ_PIECES)
                lds_dma16_buf(rdz, dz_l[i], __builtin_amdgcn_readfirstlane(smem_base + stage_off + piece * 1024));
        }
#pragma unroll
        for (int i = 0; i < X_PW; ++i) {
            const int piece = rw + NREQ * i;
            if (piece < C::X_PIECES)
                lds_dma16_buf(rx, ((mask >> i) & 1u) ? x_l[i] : LDS_DMA_OOB,
                              __builtin_amdgcn_readfirstlane(smem_base + stage_off + C::DZ_BYTES + piece * 1024));
        }
    };
    // One tile's DMA, generic form: every wave issues its share of the 1-KiB pieces of both images.
    auto issue_gen = [&](int tile, int stage_off) {
        const int p1 = (int)fdiv((unsigned)tile, d_tw), tw_i = tile - p1 * a.tiles_w;
        const int n = (int)fdiv((unsigned)p1, d_th), th_i = p1 - n * a.tiles_h;
        const int h0 = th_i * TH, w0 = tw_i * C::TW;
        const int rl = lane / SLOTS, s = lane % SLOTS;
        {
            const char* base = (const char*)(dzp + (size_t)n * a.Hout * a.Wout * a.lddz);
#pragma unroll
            for (int i = 0; i < (C::DZ_PIECES + 7) / 8; ++i) {
                const int piece = wave + 8 * i;
                if (piece < C::DZ_PIECES) {
                    const int R = piece * RPP + rl;
                    const int r = R >> 5, c = R & 31;
                    const int q = s ^ (swz_bytes<REC>(c) >> 4);
                    const int ho = h0 + r, wo = w0 + c;
                    const bool ok = ho < a.Hout && wo < a.Wout && (co0 + q * 8) < a.Cout;
                    const char* src = ok ? base + ((unsigned)((ho * a.Wout + wo) * a.lddz + q * 8) * 2u) : zero;
                    lds_dma16(src, __builtin_amdgcn_readfirstlane(smem_base + stage_off + piece * 1024));
                }
            }
        }
        {
            const char* base = (const char*)(xp + (size_t)n * a.Hin * a.Win * a.ldx);
#pragma unroll
            for (int i = 0; i < (C::X_PIECES + 7) / 8; ++i) {
                const int piece = wave + 8 * i;
                if (piece < C::X_PIECES) {
                    const int R = piece * RPP + rl;
                    const int hr = R / HWD, hc = R - hr * HWD;
                    const int q = s ^ (swz_bytes<REC>(hc) >> 4);
                    const int row = h0 * SH + hr - a.pad_h, col = w0 + hc - a.pad_w;
                    const bool ok = R < C::X_REC && row >= 0 && row < a.Hin && col >= 0 && col < a.Win && (ci0 + q * 8) < a.Cin;
                    const char* src = ok ? base + ((unsigned)((row * a.Win + col) * a.ldx + q * 8) * 2u) : zero;
                    lds_dma16(src, __builtin_amdgcn_readfirstlane(smem_base + stage_off + C::DZ_BYTES + piece * 1024));
                }
            }
        }
    };

    auto issue = [&](int tile, int stage_off) {
#ifdef CV2_X_NODMA
        return;          // timing experiment (wrong results): the loop without its requests
#endif
        if constexpr (GEN) issue_gen(tile, stage_off); else issue_fast(tile, stage_off);
    };
    int cur = 0;
    int* ring = (int*)(smem + C::SMEM);                 // DYN: ids of tiles k .. k+2 of this workgroup, slot k & 3
    auto ring_at = [&](int k) { return __builtin_amdgcn_readfirstlane(ring[k & 3]); };
    if constexpr (DYN) {
        const int xcd = (int)blockIdx.x & 7;
        const int span = (((int)gridDim.x >> 3) / nchan) * a.tiles_per_split;          // tiles of one XCD's range
        auto q_end = [&](int x2) { return min((x2 + 1) * span, a.tiles_total); };
        auto q_tile = [&](int x2, unsigned m) {
            const long long t = (long long)x2 * span + m;
            return t < (long long)q_end(x2) ? (int)t : TQ_INVALID;
        };
        int victim = xcd;
        bool dry = false;
        auto fetch = [&]() -> int {                     // wave 8, all lanes
            if (dry) return TQ_INVALID;
            for (int tries = 0; tries < 4; ++tries) {
                unsigned m = 0;
                if (lane == 0) m = atomicAdd(&a.tq[victim * nchan + chan], 1u);
                m = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
                const int t = q_tile(victim, m);
                if (t != TQ_INVALID) return t;
                int rem = 0;                            // that queue is dry: the fullest queue of this channel block
                if (lane < 8) {
                    const unsigned nx = __hip_atomic_load(&a.tq[lane * nchan + chan], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    rem = q_end(lane) - lane * span - (int)nx;
                }
                int best = -1, bestrem = 0;
#pragma unroll
                for (int l = 0; l < 8; ++l) {
                    const int r = __builtin_amdgcn_readlane(rem, l);
                    if (r > bestrem) { bestrem = r; best = l; }
                }
                if (best < 0) break;
                victim = best;
            }
            dry = true;
            return TQ_INVALID;
        };
        if (wave == 8) {
            const int t0 = fetch();
            const int t1 = fetch();
            if (lane == 0) { ring[0] = t0; ring[1] = t1; }
        }
        __syncthreads();
        if (requester && ring_at(0) != TQ_INVALID) issue(ring_at(0), 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (requester) {
            for (int k = 0;; ++k) {
                if (ring_at(k) == TQ_INVALID) break;
                const int tn = ring_at(k + 1);
                if (tn != TQ_INVALID) issue(tn, (cur ^ 1) * C::STAGE);
                if (wave == 8) {
                    const int t2 = fetch();             // its round trip runs under the DMA just requested
                    if (lane == 0) ring[(k + 2) & 3] = t2;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                cur ^= 1;
            }
            if (wave == 8 && lane == 0) {               // the last workgroup to finish hands the slot back zeroed
                if (atomicAdd(&a.tq[TQ_DONE], 1u) == gridDim.x - 1) {
                    for (int i = 0; i < 8 * nchan; ++i) a.tq[i] = 0u;
                    a.tq[TQ_DONE] = 0u;
                }
            }
            return;
        }
    } else {
        if (t_begin < t_end && requester) issue(t_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (!DYN && SPEC && requester) {
        // request wave: tile t+1 goes out as soon as the barrier has freed its stage; the barrier that ends tile t
        // says both "t+1 has landed" (these waves wait before they arrive) and "nobody reads stage cur any more"
        for (int tile = t_begin; tile < t_end; ++tile) {
            if (tile + 1 < t_end) issue(tile + 1, (cur ^ 1) * C::STAGE);
#ifndef CV2_X_NOWAIT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __syncthreads();
            cur ^= 1;
        }
        return;
    }
    int kdyn = 0;
    bool go = DYN ? ring_at(0) != TQ_INVALID : t_begin < t_end;
    for (int tile = t_begin; go; ++tile) {
        const bool more = tile + 1 < t_end;
        const char* img = smem + cur * C::STAGE;
        auto load_a = [&](int r, uint4* fa) {
#pragma unroll
            for (int i = 0; i < WM; ++i) fa[i] = tr_frag2(img + r * (32 * REC) + a_lo[i], img + r * (32 * REC) + a_hi[i]);
        };
        auto load_b = [&](int r, int t2, uint4* fb) {
            const int kh = t2 / KW, kw = t2 % KW;
#pragma unroll
            for (int j = 0; j < WN; ++j)
                fb[j] = tr_frag2(img + (r * SH + kh) * (HWD * REC) + b_lo[kw][j], img + (r * SH + kh) * (HWD * REC) + b_hi[kw][j]);
        };
        // K-step = one tile row of 32 pixels.  Every fragment of row r+1 is requested before the MFMAs of
        // row r (two register sets), so an LDS round trip is covered by a whole row of matrix work.
        uint4 fa[2][WM];
        load_a(0, fa[0]);
        if constexpr (SH == 1 && (KH == 3 || (KH == 2 && KW == 5))) {
            // The x fragment of (row r, tap kh, kw) is halo row r + kh shifted by kw: consecutive rows share two
            // of their three halo rows.  Keep a ring of four halo rows in registers and read only the new one
            // per K-step (3 fragments instead of 9: the LDS read traffic of the loop drops from 22 to 10
            // transposed reads per 18 MFMAs).
            // The ring holds THREE rows: row r's slot is free once the kh = 0 taps of step r have been issued, and the
            // row the step after next needs (r + 3) is requested into it right there, under the step's other 12 MFMAs.
            // (KH = 2, the two-row pieces a 5x5 gradient is issued in: a ring of two rows.)
            uint4 hf[KH][KW][WN];
            auto load_h = [&](int row, uint4 (*dst)[WN]) {
#pragma unroll
                for (int kw = 0; kw < KW; ++kw)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        dst[kw][j] = tr_frag2(img + row * (HWD * REC) + b_lo[kw][j], img + row * (HWD * REC) + b_hi[kw][j]);
            };
#pragma unroll
            for (int k = 0; k < KH; ++k) load_h(k, hf[k]);
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                if (r + 1 < TH) load_a(r + 1, fa[(r + 1) & 1]);
                if (!SPEC && more && r == (wave >> 2)) issue(tile + 1, (cur ^ 1) * C::STAGE);
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[t2][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, fa[r & 1][i]), __builtin_bit_cast(bf16x8, hf[(r + t2 / KW) % KH][t2 % KW][j]),
                                acc[t2][i][j], 0, 0, 0);
                    if (t2 == KW - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (r + 1 < TH) load_h(r + KH, hf[r % KH]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
        uint4 fb[2][NT][WN];
#pragma unroll
        for (int t2 = 0; t2 < NT; ++t2) load_b(0, t2, fb[0][t2]);
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            if (r + 1 < TH) {
                load_a(r + 1, fa[(r + 1) & 1]);
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) load_b(r + 1, t2, fb[(r + 1) & 1][t2]);
            }
            // waves 0-3 and 4-7 issue the next tile's DMA one K-step apart, so on every SIMD one wave
            // does address arithmetic while its partner feeds the matrix pipe
            if (!SPEC && more && r == (wave >> 2)) issue(tile + 1, (cur ^ 1) * C::STAGE);
#pragma unroll
            for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[t2][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, fa[r & 1][i]), __builtin_bit_cast(bf16x8, fb[r & 1][t2][j]), acc[t2][i][j], 0, 0, 0);
        }
        }
        if (!SPEC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile has landed (this wave's pieces)
        __syncthreads();                                    // ... everyone's; and everyone is done reading `cur`
        cur ^= 1;
        go = DYN ? ring_at(++kdyn) != TQ_INVALID : tile + 1 < t_end;
    }

    // ---- store the partial block: D[row = co = 4g+reg][col = ci = l15] ----
    float* slab = a.slab + (size_t)split * NT * a.Cout * a.Cin;
#pragma unroll
    for (int t2 = 0; t2 < NT; ++t2)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int ci = ci0 + wci + 16 * j + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wco + 16 * i + 4 * g + r;
                    if (co < a.Cout && ci < a.Cin) slab[((size_t)t2 * a.Cout + co) * a.Cin + ci] = acc[t2][i][j][r];
                }
            }
}

template <int KH, int KW, int SH, int WT, int TH, bool GEN, bool SPEC = false, bool DYN = false>
static int launch_wgrad2g(WgradArgs a, hipStream_t stream, int nsplit) {
    using C = Wg2Cfg<KH, KW, SH, WT, TH>;
    constexpr int LDS = C::SMEM + (DYN ? 64 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static DynLdsOnce lds_once;
    if (int rc = ensure_dyn_lds(lds_once, (const void*)conv_wgrad2_kernel<KH, KW, SH, WT, TH, GEN, SPEC, DYN>, LDS)) return rc;
    a.tiles_h = (a.Hout + TH - 1) / TH; a.tiles_w = (a.Wout + 31) / 32;
    a.tiles_total = a.N * a.tiles_h * a.tiles_w;
    a.tiles_per_split = (a.tiles_total + nsplit - 1) / nsplit;
    const int nco = (a.Cout + C::CT - 1) / C::CT, nci = (a.Cin + C::CT - 1) / C::CT;
    UNET_CHECK((unsigned long long)a.tiles_total * (unsigned)(a.tiles_w > a.tiles_h ? a.tiles_w : a.tiles_h) < (1ull << 32),
               "tile count too large for the index arithmetic");
    if constexpr (!DYN && SPEC && C::SMEM + 64 <= 160 * 1024) {       // (the widest halos leave no LDS for the queue's four words)
        // tile queues (tile_queue.h): a whole number of workgroups per XCD and channel block, at most 16 channel blocks, and
        // enough tiles per workgroup for the choice to matter
        const int nchan = nco * nci, grid = nsplit * nchan;
        if (tileq_enabled() && nchan <= 16 && grid % (8 * nchan) == 0 && a.tiles_per_split >= (tileq_level() == 2 ? 2 : 3)
            && !tileq_capturing(stream)) {
            a.tq = tileq_next_slot();
            if (a.tq == nullptr) return -3;
            return launch_wgrad2g<KH, KW, SH, WT, TH, GEN, SPEC, true>(a, stream, nsplit);
        }
    }
    hipLaunchKernelGGL((conv_wgrad2_kernel<KH, KW, SH, WT, TH, GEN, SPEC, DYN>), dim3((unsigned)(nsplit * nco * nci)), dim3(SPEC ? 768 : 512), LDS, stream,
                       a, nci, nco * nci, make_fastdiv((unsigned)a.tiles_w), make_fastdiv((unsigned)a.tiles_h));
    UNET_LAUNCH_CHECK();
    return 0;
}

template <int KH, int KW, int SH, int WT, int TH, bool SPEC = false>
static int launch_wgrad2(const WgradArgs& a, hipStream_t stream, int nsplit) {
    constexpr int CT = 32 * WT;
    const bool gen = (a.Cout % CT) != 0 || (a.Cin % CT) != 0 || (a.Hout % TH) != 0 || (a.Wout & 31) != 0;
    return gen ? launch_wgrad2g<KH, KW, SH, WT, TH, true>(a, stream, nsplit) : launch_wgrad2g<KH, KW, SH, WT, TH, false, SPEC>(a, stream, nsplit);
}

// geometry of the bf16 path for (kh, kw, stride): channel tile and rows per K tile
void wgrad2_geometry(int kh, int kw, int sh, int* ct, int* th) {
    if (sh == 2) { *ct = 128; *th = 2; } else { *ct = 64; *th = 8; }
}

int wgrad2_splits(int kh, int kw, int sh, int N, int Hout, int Wout, int Cin, int Cout) {
    int ct, th;
    wgrad2_geometry(kh, kw, sh, &ct, &th);
    const int tiles = N * ((Hout + th - 1) / th) * ((Wout + 31) / 32);
    const int blocks = ((Cout + ct - 1) / ct) * ((Cin + ct - 1) / ct);
    int want = (256 + blocks - 1) / blocks;           // one 512-thread workgroup per CU
    if (want > tiles) want = tiles;
    if (want < 1) want = 1;
    const int per = (tiles + want - 1) / want;
    return (tiles + per - 1) / per;
}

int dispatch_wgrad2(int kh, int kw, int sh, const WgradArgs& a, hipStream_t s, int nsplit) {
    if (sh == 2) {
        UNET_CHECK(kh == 2 && kw == 1, "stride-2 form is the 2x1 convT view only");
        return launch_wgrad2<2, 1, 2, 4, 2, true>(a, s, nsplit);
    }
    if (kh == 3 && kw == 3) return launch_wgrad2<3, 3, 1, 2, 8, true>(a, s, nsplit);
    if (kh == 1 && kw == 1) return launch_wgrad2<1, 1, 1, 2, 8, true>(a, s, nsplit);
    if (kh == 3 && kw == 1) return launch_wgrad2<3, 1, 1, 2, 8, true>(a, s, nsplit);
    if (kh == 1 && kw == 3) return launch_wgrad2<1, 3, 1, 2, 8, true>(a, s, nsplit);
    if (kh == 5 && kw == 1) return launch_wgrad2<5, 1, 1, 2, 8, true>(a, s, nsplit);
    if (kh == 1 && kw == 5) return launch_wgrad2<1, 5, 1, 2, 8, true>(a, s, nsplit);
    if (kh == 2 && kw == 5) return launch_wgrad2<2, 5, 1, 2, 8, true>(a, s, nsplit);      // two rows of a 5x5 gradient per launch
    unet_set_error("unsupported wgrad kernel %dx%d stride %d", kh, kw, sh);
    return -1;
}
