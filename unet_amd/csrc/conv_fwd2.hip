// bf16 3x3 stride-1 implicit-GEMM convolution, second generation (gfx950): forward and dgrad of
// nn.Conv2d(k=3,pad=1) (reference models/unet.py:10,13), same contract as conv_igemm.hip's 3x3 path
// (bias, BN partial statistics, accumulate / ReLU flags, split destination).
//
// What changes against conv_igemm.hip is how operands reach the matrix cores:
//   * persistent 512-thread workgroups, one per CU, walk output tiles of 16x32 pixels x BCO channels
//     (BCO = 128: waves 2 (co) x 4 (row groups), 64 co x 128 px each; BCO = 64: 1 x 8, 64 co x 64 px).
//   * a tile is a sequence of PHASES (32-channel chunk, kernel row kh): the phase's three taps of
//     weights [3][BCO][32ch] and, once per chunk, the input halo tile [18][34][32ch] are written into
//     LDS by global_load_lds_dwordx4 (no staging registers, no ds_write) while the previous phase is
//     multiplied: two halo stages, two weight stages, one barrier per phase.  The pipeline runs across
//     tile boundaries, so the next tile's first operands load under the epilogue.
//   * zero padding (borders, ragged tiles, channel tails): the lane's SOURCE address points at a block
//     of zeros; the LDS destination is lane-linear.
//   * LDS images are 64-byte records (pixel or output channel x 32 input channels); the 16-byte slot is
//     XOR-swizzled by a function of the pixel COLUMN / channel index, applied to the DMA source and to
//     the ds_read_b128 fragment reads (conflict-free for every tap shift; rows add immediates).
#include "common.h"
#include "conv_args.h"
#include "lds_dma.h"
#include "tile_queue.h"
#include <climits>
#include <type_traits>
#include <cstdlib>


template <int WC, int KH = 3, int KW = 3>
struct Cv2Cfg {
    static_assert((KH == 3 && KW == 3) || WC == 1, "kernels other than 3x3 run on 64-channel tiles");
    static constexpr int BCO = 64 * WC;            // output channels per workgroup
    static constexpr int NWR = 8 / WC;             // wave row groups
    static constexpr int TH = 16, TW = 32;
    static constexpr int RW = TH / NWR;            // tile rows per wave (4 or 2)
    static constexpr int NPG = RW * 2;             // 16-pixel groups per wave
    static constexpr int HP = TW + KW - 1;         // halo row pitch (records): 32 / 34 / 36
    static constexpr int HREC = (TH + KH - 1) * HP;   // 612 records for 3x3, 720 for 5x5
    static constexpr int H_PIECES = (HREC + 127) / 128 * 8;      // 16 records per 1-KiB piece, padded to a multiple of 8: 40 / 48
    static constexpr int H_BYTES = H_PIECES * 1024;
    // taps per phase: one kernel column (kw) for 128-channel tiles and for 5x5; the whole 3x3 kernel for 64-channel tiles
    // (k x 1 and 1 x k, the RepUNet training branches of models/reparm_unet.py:55-61: the whole kernel, one phase)
    static constexpr int TPP = (WC == 2 || (KH == 5 && KW == 5)) ? KH : KH * KW;
    static constexpr int NPH = KH * KW / TPP;      // phases per 32-channel chunk
    static constexpr int W_PIECES = TPP * BCO / 16;
    static constexpr int W_BYTES = W_PIECES * 1024;
    static constexpr int OFF_H = 0, OFF_W = 2 * H_BYTES, OFF_RED = OFF_W + 2 * W_BYTES;
    static constexpr int RED_BYTES = NWR * BCO * 2 * 4;
    static constexpr int OFF_BIAS = OFF_RED + RED_BYTES;        // two 1-KiB slots: the tile's BCO fp32 biases
    // producer scheme (WC == 2): the per-lane source offsets of each producer wave's 10 halo pieces, [wave][piece][lane]
    static constexpr int OFF_HOFF = OFF_BIAS + 2 * 1024;
    // producer scheme: waves 0-3 issue every request.  Not for one-phase chunks: there a wave refills, after its own
    // epilogue, exactly the halo pieces it borrowed as its transposition tile, which only the symmetric split guarantees.
    static constexpr bool ASYM = WC == 2 || (KH == 5 && KW == 5);
    static constexpr int HOFF_BYTES = ASYM ? 4 * (H_PIECES / 4) * 256 : 0;
    static constexpr int SMEM = OFF_HOFF + HOFF_BYTES;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

// TIMING-ONLY VARIANTS (never in the shipped library; results are wrong by construction): make EXTRA=-DCV2_X_<NAME>
// LIB=../libunet_hip_x_<name>.so BUILD=build_x_<name>, compared on one box with tools/ab_variants.sh.
//   CV2_X_NODMA   no LDS-DMA request is issued (the loop multiplies whatever the LDS holds)
//   CV2_X_NOMFMA  fragments are read, no MFMA is issued
//   CV2_X_NOEPI   accumulators stay live, nothing is stored
//   CV2_X_NOBAR   the per-phase barrier is dropped
// (conv_wgrad2.hip: CV2_X_NODMA, CV2_X_NOWAIT = request waves do not wait for their requests.)  DESIGN 4.1 has the numbers.
#ifdef CV2_TRACE
// diagnostic build only (make TRACE=1): s_memtime stamps of block 0, waves 0 and 4
__device__ unsigned long long* g_cv2_trace = nullptr;
extern "C" int unet_debug_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cv2_trace), &p, sizeof(p)); }
#define STAMP(tag) do { if (trc && tn < 500) { trc[tn * 2] = __builtin_amdgcn_s_memtime(); trc[tn * 2 + 1] = (tag); ++tn; } } while (0)
#else
#define STAMP(tag) do { } while (0)
#endif

__device__ __forceinline__ int swz64(int idx) { return ((idx >> 2) & 1) << 1; }   // slot XOR of a 64-byte record

// GEN = false: the specialised kernel of the U-Net shapes (channel counts multiples of the tile, output
// a whole number of 16x32 tiles, plain store, split point on a 64-channel boundary); GEN = true: the
// same pipeline with every bound checked per lane and the accumulate / ReLU / arbitrary-split stores.
// SPEC = true (64-channel tiles of the specialised kernel, no deferred statistics): SPECIALISED REQUEST WAVES, as in
// conv_wgrad2.hip.  The workgroup has 12 waves, three per SIMD: waves 0-7 are the matrix waves, waves 8-11 issue every
// LDS-DMA request (19-20 pieces per chunk each), wait for them and meet the others at the barriers.
// KH x KW = 5x5: the 5x5 convolutions of a deployed RepUNet (models/reparm_unet.py:139, kernel_sizes [3,5] folded) and of its
// training branches, on 64-channel tiles with one kernel column (5 taps) per phase.
// 3x1 / 5x1 / 1x3 / 1x5: the vertical and horizontal branches of the AsymmConvBlock (models/reparm_unet.py:55-61), whole
// kernel per phase; a 1 x k kernel is k columns of one row each, its A fragments double-buffered per column.
// ACC = true (specialised kernel, no request waves): flags bit 0 ADDS the result to what the destination holds -- the branch
// dgrads of a RepUNet block (models/reparm_unet.py:62-72,110-122: six convolutions of one input) summed into one dx.
// DYN = true (128-channel tiles and the request-wave kernel): tiles are drawn from one atomic queue per XCD (tile_queue.h): the
// workgroups of an XCD take consecutive tiles of that XCD's eighth of the tile range, a dry queue steals from the fullest.
// A workgroup holds the ids of its current and its next tile (the DMA cursors run up to one tile ahead; with Cin >= 64 a
// tile has at least two chunks, so no cursor reaches tile k+2 before the last chunk of tile k): two are drawn at the start,
// then matrix wave 7 (it issues no DMA, so its memory counter is free) requests the id of tile k+2 at the top of tile k and
// collects it at the tile's last phase; the ids travel through four LDS words to the cursors and the request waves.
// BNR = true (the 3x3 dgrad of a DoubleConv's second conv, specialised kernel): the FIRST PASS OF THE BATCHNORM BACKWARD in front
// of it rides on the epilogue.  The output tile is the gradient dy of that BatchNorm + ReLU's output; the lane reads the matching
// values of the saved pre-norm tensor z (8 bytes per 16-pixel group and channel quad, the accumulator layout) and the statistics
// rows carry sum dy' and sum dy' * xhat (dy' = dy where z*scale + shift > 0) -- what bn_bwd_reduce_kernel computes in a pass of
// its own over dy and z (4 B/element; here 2 B/element, and dy is summed before it is rounded to bf16).
template <int WC, bool GEN, bool DST = false, bool SPEC = false, int KH = 3, int KW = 3, bool ACC = false, bool DYN = false, bool BNR = false>
__global__ __launch_bounds__(SPEC ? 768 : 512) void conv3x3_v2_kernel(ConvArgs a, int nco, int total_tiles, FastDiv d_nco, FastDiv d_tw, FastDiv d_th) {
    using C = Cv2Cfg<WC, KH, KW>;
    static_assert((KH == 3 && KW == 3) || (!GEN && !DST && !SPEC), "other than 3x3: the specialised kernel only");
    static_assert(!ACC || (!GEN && !DST && !SPEC), "accumulating epilogue: the specialised kernel without request waves");
    static_assert(!DYN || (!GEN && !DST && !ACC && KH == 3 && KW == 3 && (WC == 2 || SPEC)), "tile queues: 128-channel tiles or request waves");
    static_assert(!BNR || (!GEN && !DST && !ACC && !DYN && KH == 3 && KW == 3), "BatchNorm-backward sums: the plain specialised 3x3 kernel");
    constexpr int BCO = C::BCO, RW = C::RW, NPG = C::NPG, HP = C::HP, TPP = C::TPP, NPH = C::NPH;

    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned sb = lds_addr(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int wc = (WC == 2) ? (wave >> 2) : 0;
    const int wr = (WC == 2) ? (wave & 3) : wave;
    const int nchunks = (a.Cin + 31) >> 5;
#ifdef CV2_TRACE
    unsigned long long* trc = (g_cv2_trace && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0) ? g_cv2_trace + (wave >> 2) * 1024 : nullptr;
    int tn = 0;
    // in-kernel clock: shader-cycle counter against the 100 MHz real-time counter, around the whole kernel
    if (trc && wave == 0) { g_cv2_trace[2048] = __builtin_amdgcn_s_memtime(); g_cv2_trace[2049] = __builtin_amdgcn_s_memrealtime(); }
#endif

    // ---- per-lane fragment bases ------------------------------------------------------------------
    // A (weights): record = tapslot*BCO + wc*64 + ct*16 + l15, k-slot g
    const int a_base = C::OFF_W + (wc * 64 + l15) * 64 + ((g ^ swz64(l15)) << 4);
    // B (halo): record = (wr*RW + rr + kh)*HP + cc*16 + l15 + kw, k-slot g
    int b_base[KW][2];
#pragma unroll
    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const int col = cc * 16 + l15 + kw;
            b_base[kw][cc] = C::OFF_H + ((wr * RW) * HP + col) * 64 + ((g ^ swz64(col)) << 4);
        }

    // ---- DMA addressing ---------------------------------------------------------------------------
    // Every piece is one buffer_load ... lds whose descriptor base is wave-uniform (tile origin, chunk,
    // tap) and whose per-lane offset is a kernel-lifetime constant: hoff[i] for the lane's record/slot
    // of halo piece wave+8i, woff for its row/slot of a 16-row weight piece.  A padded lane gets the
    // out-of-range offset and the hardware writes zeros.  Per tile one bit mask says which of the
    // lane's five halo pieces lie inside the image; channel counts that are not multiples of the tile
    // (never in the U-Net) take the GENERIC path that recomputes everything per piece.
    // PRODUCER WAVES.  With 128-channel tiles (WC == 2) only waves 0-3 issue DMA, twice the pieces each; their
    // SIMD partners (waves 4-7) never do.  After a barrier the partner therefore has the matrix pipe to itself
    // while the producer issues, and the producer has it while the partner waits at the next barrier: the two
    // waves of a SIMD run in anti-phase by construction instead of stalling on DMA issue side by side.
    constexpr bool ASYM = C::ASYM;
    static_assert(!SPEC || (WC == 1 && !GEN), "request waves: 64-channel tiles of the specialised kernel");
    constexpr int NPROD = (ASYM || SPEC) ? 4 : 8;             // waves that issue DMA
    constexpr int HPW = C::H_PIECES / NPROD;                  // halo pieces per producer wave
    static_assert(C::H_PIECES % NPROD == 0, "halo pieces must divide among the producer waves");
    const bool producer = SPEC ? wave >= 8 : (!ASYM || wave < 4);
    const int pidx = SPEC ? (wave & 3) : wave;                // index among the waves that issue DMA
    const int dl = lane >> 2, ds = lane & 3;       // record within the piece, 16-byte slot
    const int wq = ds ^ swz64(dl);                 // weight source chunk (record index % 16 == dl)
    const unsigned woff = (unsigned)(dl * a.Cin + wq * 8) * 2u;
    // per-lane source offset of halo piece `piece`; kept in registers for the symmetric scheme (5 per lane),
    // recomputed per request for the producer scheme (10 per lane would not fit the register budget)
    auto h_off_calc = [&](int piece, int lane_) -> unsigned {
        const int r = piece * 16 + (lane_ >> 2);
        const int hr = r / HP, hc = r - hr * HP;
        return r < C::HREC ? (unsigned)((hr * a.Win + hc) * a.ldx + ((lane_ & 3) ^ swz64(hc)) * 8) * 2u : LDS_DMA_OOB;
    };
    constexpr int HREG = ASYM ? 1 : HPW;                      // symmetric: 5 per lane; request waves: 10 (no accumulators there)
    unsigned hoff[HREG];
#pragma unroll
    for (int i = 0; i < HREG; ++i) hoff[i] = (ASYM || !producer) ? 0u : h_off_calc(pidx + NPROD * i, lane);
    // Producer scheme: 10 offsets per lane do not fit the register budget, and recomputing them per piece costs ~15 VALU
    // instructions that have to squeeze between the SIMD partner's MFMAs (an MFMA holds the vector issue for 8 of its 16
    // cycles): the in-kernel trace showed 1,700 cycles of refill per phase for 9-10 pieces.  They are computed once into
    // LDS; a refill reads its 10 words back (wave-private rows: no barrier needed) and spends one select per piece.
    unsigned* hoff_lds = (unsigned*)(smem + C::OFF_HOFF) + (ASYM ? (wave & 3) * HPW * 64 : 0);
    if (ASYM && producer) {
#pragma unroll
        for (int i = 0; i < HPW; ++i) hoff_lds[i * 64 + lane] = h_off_calc(pidx + NPROD * i, lane);
    }
    constexpr bool generic = GEN;
    const char* xg = (const char*)((const __bf16*)a.x + a.xoff);
    const char* wg = (const char*)a.w;

    struct Tile { int cb, n, h0, w0, pt; };
    auto decode = [&](int t) {
        Tile o;
        o.pt = (int)fdiv((unsigned)t, d_nco); o.cb = t - o.pt * nco;
        const int p1 = (int)fdiv((unsigned)o.pt, d_tw), tw_i = o.pt - p1 * a.tiles_w;
        o.n = (int)fdiv((unsigned)p1, d_th);
        const int th_i = p1 - o.n * a.tiles_h;
        o.h0 = th_i * C::TH; o.w0 = tw_i * C::TW;
        return o;
    };
    // bit i: halo piece wave + NPROD*i of this lane lies inside the image for tile t
    auto halo_mask = [&](const Tile& t) {
        const int r0 = t.h0 - a.pad_h, c0 = t.w0 - a.pad_w;
        unsigned m = 0;
        if (r0 >= 0 && r0 + C::TH + KH - 1 <= a.Hin && c0 >= 0 && c0 + HP <= a.Win) return (1u << HPW) - 1u;     // interior tile (wave-uniform)
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
#pragma unroll
        for (int i = 0; i < HPW; ++i) {
            const int r = (pidx + NPROD * i) * 16 + (lane_ >> 2);
            const int hr = r / HP, hc = r - hr * HP;
            if ((unsigned)(r0 + hr) < (unsigned)a.Hin && (unsigned)(c0 + hc) < (unsigned)a.Win) m |= 1u << i;
        }
        return m;
    };
    // weights of one phase: taps of the phase, channels [c*32, c*32+32) of output-channel block cb; a wave's share is
    // WPW pieces (piece i of wave w = 16 rows of the [TPP][BCO] image starting at row (w + NPROD*i)*16)
    constexpr int WPW = (C::W_PIECES + NPROD - 1) / NPROD;
    auto issue_bias = [&](const Tile& t, int bslot) {
        // the tile's biases ride with its first weights (wave 0, one piece; lanes past the block read zeros)
        const int nb = min(BCO, a.Cout - t.cb * BCO) * 4;
        lds_dma16_buf(make_rsrc(a.bias + t.cb * BCO, (unsigned)nb), (unsigned)lane * 16u,
                      __builtin_amdgcn_readfirstlane(sb + C::OFF_BIAS + bslot * 1024));
        return 1;
    };
    auto issue_w_piece = [&](const Tile& t, int c, int ph, int wstage, int i) {
        const int p = pidx + NPROD * i;
        if (p >= C::W_PIECES) return 0;
        const int r0 = p * 16;
        const int j = r0 / BCO, cou = t.cb * BCO + (r0 % BCO);
        const int tap = (NPH == 1) ? j : j * KW + ph;                 // a one-column phase holds taps (kh = j, kw = ph)
        const long long offu = ((long long)(tap * a.Cout + cou) * a.Cin + c * 32) * 2;   // wave-uniform
        unsigned v = woff;
        if (generic) {
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            const int dl_ = lane_ >> 2, wq_ = (lane_ & 3) ^ swz64(dl_);
            v = ((cou + dl_) < a.Cout && (c * 32 + wq_ * 8) < a.Cin) ? woff : LDS_DMA_OOB;
        }
        lds_dma16_buf(make_rsrc(wg + offu, LDS_DMA_RECORDS), v,
                      __builtin_amdgcn_readfirstlane(sb + C::OFF_W + wstage * C::W_BYTES + p * 1024));
        return 1;
    };
    auto issue_w = [&](const Tile& t, int c, int ph, int wstage, int bslot, bool bias_only) {
        int cnt = 0;
        if (!producer) return cnt;
#ifdef CV2_X_NODMA
        return cnt;
#endif
        if (c == 0 && ph == 0 && a.bias != nullptr && pidx == 0) cnt += issue_bias(t, bslot);
        if (bias_only) return cnt;
#pragma unroll
        for (int i = 0; i < WPW; ++i) cnt += issue_w_piece(t, c, ph, wstage, i);
        return cnt;
    };
    // input halo of one chunk: rows h0-pad .. +17, cols w0-pad .. +33, channels [c*32, c*32+32)
    auto halo_rsrc = [&](const Tile& t, int c) {
        const int r0 = t.h0 - a.pad_h, c0 = t.w0 - a.pad_w;
        const long long offu = (((long long)t.n * a.Hin + r0) * a.Win + c0) * a.ldx * 2 + c * 64;          // wave-uniform
        return make_rsrc(xg + offu, LDS_DMA_RECORDS);
    };
    auto issue_h_piece = [&](const i32x4_t rsrc, int c, int hstage, unsigned mask, int i, unsigned off) {
        const int piece = pidx + NPROD * i;
        unsigned v = ((mask >> i) & 1u) ? off : LDS_DMA_OOB;
        if (generic) {
            int lane_ = lane;
            asm volatile("" : "+v"(lane_));
            const int r = piece * 16 + (lane_ >> 2);
            const int hr = r / HP, hc = r - hr * HP;
            if (c * 32 + ((lane_ & 3) ^ swz64(hc)) * 8 >= a.Cin) v = LDS_DMA_OOB;
        }
        lds_dma16_buf(rsrc, v, __builtin_amdgcn_readfirstlane(sb + C::OFF_H + hstage * C::H_BYTES + piece * 1024));
    };
    auto issue_h = [&](const Tile& t, int c, int hstage, unsigned mask) {
        if (!producer) return 0;
#ifdef CV2_X_NODMA
        return 0;
#endif
        const i32x4_t rsrc = halo_rsrc(t, c);
        unsigned ho[ASYM ? HPW : 1];
        if (ASYM) {
#pragma unroll
            for (int i = 0; i < HPW; ++i) ho[i] = hoff_lds[i * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < HPW; ++i) issue_h_piece(rsrc, c, hstage, mask, i, ASYM ? ho[i % (ASYM ? HPW : 1)] : hoff[i % HREG]);
        return HPW;
    };

    // ---- issue-side cursors: the next weight phase and the next halo chunk to request ------------------
    int* ring = (int*)(smem + C::SMEM);                 // DYN: ids of this workgroup's tiles k-1 .. k+2, slot k & 3
    auto ring_at = [&](int k) { return __builtin_amdgcn_readfirstlane(ring[k & 3]); };
    struct Cursor { int tile_i, c, ph, seq, sn; Tile t; bool valid; };
    auto cursor_at = [&](int tile_i) {
        Cursor q;
        q.tile_i = tile_i; q.c = 0; q.ph = 0; q.seq = 0; q.sn = 0;
        q.valid = tile_i < total_tiles;
        q.t = decode(q.valid ? tile_i : 0);
        return q;
    };
    auto next_tile = [&](Cursor& q) {
        q.c = 0; q.seq ^= 1;
        if constexpr (DYN) q.tile_i = ring_at(++q.sn); else q.tile_i += gridDim.x;
        q.valid = q.tile_i < total_tiles;
        if (q.valid) q.t = decode(q.tile_i);
    };
    if ((int)blockIdx.x >= total_tiles) return;
    // ---- DYN: the tile queue of this workgroup's XCD (wave 7 draws) ----------------------------------------------
    const int tq_span = (total_tiles + 7) >> 3;          // tiles of one XCD's range
    auto q_end = [&](int x2) { return min((x2 + 1) * tq_span, total_tiles); };
    auto q_tile = [&](int x2, unsigned m) {
        const long long t = (long long)x2 * tq_span + m;
        return t < (long long)q_end(x2) ? (int)t : TQ_INVALID;
    };
    int tq_victim = (int)blockIdx.x & 7, tq_pend_q = 0;
    bool tq_dry = false, tq_has_pend = false;
    unsigned tq_pend = 0;
    auto tq_request = [&]() {                            // wave 7: the atomic goes out, nobody waits
        tq_has_pend = !tq_dry;
        tq_pend_q = tq_victim;
        if (tq_has_pend && lane == 0) tq_pend = atomicAdd(&a.tq[tq_victim], 1u);
    };
    auto tq_collect = [&]() -> int {                     // wave 7: the requested id; a dry queue steals (synchronously, rare)
        if (!tq_has_pend) return TQ_INVALID;
        tq_has_pend = false;
        int t = q_tile(tq_pend_q, (unsigned)__builtin_amdgcn_readfirstlane((int)tq_pend));
        for (int tries = 0; t == TQ_INVALID && tries < 4; ++tries) {
            int rem = 0;
            if (lane < 8) {
                const unsigned nx = __hip_atomic_load(&a.tq[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                rem = q_end(lane) - lane * tq_span - (int)nx;
            }
            int best = -1, bestrem = 0;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const int r = __builtin_amdgcn_readlane(rem, l);
                if (r > bestrem) { bestrem = r; best = l; }
            }
            if (best < 0) break;
            tq_victim = best;
            unsigned m2 = 0;
            if (lane == 0) m2 = atomicAdd(&a.tq[best], 1u);
            t = q_tile(best, (unsigned)__builtin_amdgcn_readfirstlane((int)m2));
        }
        if (t == TQ_INVALID) tq_dry = true;
        return t;
    };
    auto tq_done = [&]() {                               // wave 7: the last workgroup to finish hands the slot back zeroed
        if (lane == 0 && atomicAdd(&a.tq[TQ_DONE], 1u) == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a.tq[i] = 0u;
            a.tq[TQ_DONE] = 0u;
        }
    };
    if constexpr (DYN) {
        if (wave == 7) {                                 // the first two tiles in one draw from the own queue
            unsigned m = 0;
            if (lane == 0) m = atomicAdd(&a.tq[tq_victim], 2u);
            m = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
            const int t0 = q_tile(tq_victim, m), t1 = q_tile(tq_victim, m + 1);
            if (t1 == TQ_INVALID) tq_dry = true;         // an id after an invalid one would never be computed: stop drawing
            if (lane == 0) { ring[0] = t0; ring[1] = t1; ring[2] = TQ_INVALID; ring[3] = TQ_INVALID; }
        }
        __syncthreads();
        if (ring_at(0) == TQ_INVALID) {
            if (wave == 7) tq_done();
            return;
        }
    }
    const int lblock = DYN ? ring_at(0) : xcd_block_id();
    Cursor wq_ = cursor_at(lblock);        // weights: advances phase by phase
    Cursor hq_ = wq_;                          // halo: advances chunk by chunk
    unsigned hmask = halo_mask(hq_.t);
    int wstage_i = 0, hstage_i = 0;
    // Bookkeeping for counted waits: this wave's running count of DMA instructions and, per stage, the
    // count right after the stage's last request.  vmcnt retires in order, so "stage X has landed" is
    // "all but (n_issued - mark[X]) of my youngest requests are done".
    int n_issued = 0, w_mark0 = 0, w_mark1 = 0, h_mark0 = 0, h_mark1 = 0;
    // RESIDENT WEIGHTS: whole-kernel phases (64-channel tiles), two chunks (Cin = 64) and one channel block: the two weight
    // stages then hold chunk 0 and chunk 1 for good -- every later request would rewrite a stage with the bytes it already
    // holds -- so after the first two only the tile's bias piece is requested: 72 of a tile's 152 DMA pieces go away
    // (64->64 @512^2, same box: forward 0.770 -> 0.740 ms, dgrad 0.644 -> 0.604 ms).
    const bool w_resident = NPH == 1 && nchunks == 2 && nco == 1;
    int w_pushed = 0;
    auto push_w = [&]() {
        if (!wq_.valid) return;
        n_issued += issue_w(wq_.t, wq_.c, wq_.ph, wstage_i, wq_.seq, w_resident && w_pushed >= 2);
        w_pushed += w_pushed < 2;
        if (wstage_i) w_mark1 = n_issued; else w_mark0 = n_issued;
        wstage_i ^= 1;
        if (++wq_.ph == NPH) { wq_.ph = 0; if (++wq_.c == nchunks) next_tile(wq_); }
    };
    auto push_h = [&]() {
        if (!hq_.valid) return;
        n_issued += issue_h(hq_.t, hq_.c, hstage_i, hmask);
        if (hstage_i) h_mark1 = n_issued; else h_mark0 = n_issued;
        hstage_i ^= 1;
        if (++hq_.c == nchunks) { next_tile(hq_); if (hq_.valid) hmask = halo_mask(hq_.t); }
    };
    // REFILL after the barrier that ends phase (c, ph): the weights two phases ahead go into the stage
    // just freed; after a chunk's first phase also the next chunk's halo (weights first: the halo may
    // then stay in flight past the next wait).
    // The epilogue borrows, as its transposition tile, the two pieces of the halo stage just read that
    // THIS wave refills (pieces wave and wave + 8); so at a tile's end a wave requests that halo only
    // after its own epilogue (matters for one-phase chunks, where every barrier frees a halo stage).
    auto refill = [&](int ph_done) {
        push_w();
        if (ph_done == 0) push_h();
    };
    auto wait_landed = [&](int need) {
        const int allowed = n_issued - need;      // my requests younger than the ones I need
        if (allowed <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (allowed == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (allowed == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (allowed == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (allowed == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (allowed == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (allowed == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (allowed == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (allowed < 10) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (allowed < 12) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (allowed < 16) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    };
    // prologue: phases 0 and 1 and halo chunk 0 (one-phase chunks: halo chunks 0 and 1)
    push_w(); push_h(); push_w();
    if (NPH == 1) push_h();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if constexpr (SPEC) {
        if (producer) {
            // ---- request waves (one-phase chunks): after the barrier that ends chunk k they request chunk k+2 into the
            // stages just freed, then wait for chunk k+1 and arrive at the next barrier.  At a tile's end the matrix
            // waves transpose their output through the halo stage just read, so that stage's refill waits for the
            // barrier behind the epilogue (the statistics' first barrier, or one of its own).
            int ws_r = 0, hs_r = 0;
            for (int t_i = lblock, t_sn = 0; t_i < total_tiles; t_i = DYN ? ring_at(++t_sn) : t_i + (int)gridDim.x) {
                for (int c = 0; c < nchunks; ++c) {
                    const int need = max(ws_r ? w_mark0 : w_mark1, hs_r ? h_mark0 : h_mark1);
                    wait_landed(need);
                    __syncthreads();
                    push_w();
                    if (c + 1 < nchunks) push_h();
                    ws_r ^= 1; hs_r ^= 1;
                }
                __syncthreads();
                push_h();
                // statistics: the matrix waves' cross-wave reduction has a second barrier -- per tile, or (one row per
                // workgroup, DST) on the workgroup's last tile only
                if (a.stats != nullptr && (!DST || t_i + (int)gridDim.x >= total_tiles)) __syncthreads();
            }
            return;
        }
    }

    int tile_i = lblock, tseq = 0, tile_sn = 0;
    Tile cur = decode(tile_i);
    int hs = 0, ws = 0;        // stages holding the current chunk's halo / the current phase's weights
    // STAGGER (symmetric scheme, 64-channel tiles): waves 0-3 refill right after the barrier, their SIMD partners
    // (waves 4-7) in the middle of the next phase's MFMAs.
    const bool late = !ASYM && !SPEC && wave >= 4;
    int late_ph = -1;          // the refill waves 4-7 still owe (-1: none)
    bool h_owed = false;       // waves 0-3, one-phase chunks: the halo request held back until after the epilogue
    // DEFERRED STATISTICS (64-channel tiles of the specialised kernel: one channel block, so every tile of this
    // workgroup covers the same 64 channels).  The lane keeps its (sum, sum of squares) partials in registers across
    // tiles; the 16-lane reduction, the cross-wave reduction and its two barriers run once, on the workgroup's LAST
    // tile, whose stats row then carries the whole workgroup's sums - the rows of its other tiles are written as
    // zeros, so "sum over rows" (all any consumer does with them) is unchanged.  Per tile this removes 32 DPP
    // reductions per wave and two barriers: the forward epilogue of these HBM-bound layers was 2.4x the dgrad one.
    // With REQUEST WAVES (SPEC) the kernel has to fit three waves per SIMD (168 registers) and cannot afford those 32: there the
    // 16-lane reduction stays per tile (128 DPP adds per wave, ~3 % of a tile), but its result is ADDED to the wave's private
    // row of the reduction area instead of being handed to a cross-wave reduction, so the two barriers, the cross-wave sum and
    // the row store still run once per workgroup and the statistics cost no register across the main loop.
    constexpr bool DEFER = DST;      // own instantiation: the 32 extra live registers cost the launches without statistics 6 %
    constexpr bool DEFER_REG = DST && !SPEC, DEFER_LDS = DST && SPEC;
    static_assert(!DST || (!GEN && WC == 1 && !DYN), "deferred statistics are for the specialised 64-channel kernel");
    float run1[DEFER_REG ? 4 : 1][4], run2[DEFER_REG ? 4 : 1][4];
#pragma unroll
    for (int i = 0; i < (DEFER_REG ? 4 : 1); ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { run1[i][r] = 0.f; run2[i][r] = 0.f; }
    if constexpr (DEFER_LDS) {       // this wave's row [BCO][2] of the reduction area starts at zero (wave-private: no barrier)
        float* row = (float*)(smem + C::OFF_RED) + wr * BCO * 2;
        row[lane] = 0.f; row[lane + 64] = 0.f;
    }
    for (;;) {
        const bool last_tile = tile_i + (int)gridDim.x >= total_tiles;
        if constexpr (DYN) { if (wave == 7) tq_request(); }        // id of tile k+2, collected at this tile's last phase
        f32x4 acc[4][NPG];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NPG; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                STAMP(1);
                if constexpr (DYN) {
                    if (wave == 7 && c == nchunks - 1 && ph == NPH - 1) {
                        const int t3 = tq_collect();
                        if (lane == 0) ring[(tile_sn + 2) & 3] = t3;
                    }
                }
                // ---- this phase: TPP taps -----------------------------------------------------------
                const char* hb = smem + hs * C::H_BYTES;
                const char* wb = smem + ws * C::W_BYTES;
                // A phase is KWP kernel COLUMNS kw (one for 128-channel tiles, all three for 64-channel ones).  Inside a column
                // the wave walks its HR = RW + 2 halo rows: the B fragment of (halo row hr, 16-pixel half cc) is read from
                // LDS ONCE and multiplied with the weights of every kernel row kh that maps it to one of the wave's output
                // rows (rr = hr - kh), so a fragment feeds up to 12 MFMAs instead of 4 and the halo is read 1.5-2x less
                // often than tap by tap.  (Tap by tap the LDS reads of a 128-channel tile were 2,300 of the 3,072 cycles a
                // phase's MFMAs take: the kernel was as much LDS-bound as MFMA-bound.)  The A fragments of the column's
                // three taps stay in registers (12 x 4 VGPRs); B fragments run DEPTH steps ahead in a small ring.
                constexpr int HR = RW + KH - 1, KWP = TPP / KH, DEPTH = 2, RING = 3, NSTEP = KWP * HR * 2;
                static_assert(KH == 3 || KH == 1 || KWP == 1, "cross-column A prefetch: three kernel rows, or one (double-buffered)");
                constexpr int NA = KH == 1 ? 2 : KH;           // A register sets: one per kernel row; 1 x k: two, alternating per column
                uint4 fa[NA][4], fb[RING];
                auto load_a = [&](int kwi, int kh) {
                    const int j = (KWP == 1) ? kh : kh * KW + kwi;         // tap slot inside the phase's weight image
                    const int set = KH == 1 ? (kwi & 1) : kh;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) fa[set][ct] = *(const uint4*)(wb + a_base + j * (BCO * 64) + ct * 1024);
                };
                auto load_b = [&](int step) {
                    const int kw = ph * KWP + step / (HR * 2), hr = (step >> 1) % HR, cc = step & 1;
                    return *(const uint4*)(hb + b_base[kw][cc] + hr * (HP * 64));
                };
                load_a(0, 0);
#pragma unroll
                for (int s0 = 0; s0 < DEPTH; ++s0) fb[s0 % RING] = load_b(s0);
                auto steps = [&](auto first_half) {
                    constexpr int S0 = decltype(first_half)::value ? 0 : NSTEP / 2;
                    constexpr int S1 = decltype(first_half)::value ? NSTEP / 2 : NSTEP;
#pragma unroll
                    for (int st = S0; st < S1; ++st) {
                        const int kwi = st / (HR * 2), hr = (st >> 1) % HR, cc = st & 1;
                        if (st + DEPTH < NSTEP) fb[(st + DEPTH) % RING] = load_b(st + DEPTH);
                        // A fragments: a column's kh = 1, 2 taps follow its first steps; the next column's taps are
                        // requested as soon as the registers they replace have seen their last MFMA
                        if (kwi == 0 && 1 + 2 * hr + cc < KH) load_a(0, 1 + 2 * hr + cc);      // kh = 1.. follow the first steps
                        if constexpr (KH == 3) {
                            if (cc == 0 && kwi + 1 < KWP && hr == RW) load_a(kwi + 1, 0);
                            if (cc == 0 && kwi + 1 < KWP && hr == RW + 1) load_a(kwi + 1, 1);
                            if (cc == 0 && kwi > 0 && hr == 0) load_a(kwi, 2);
                        } else if constexpr (KH == 1) {
                            if (cc == 0 && hr == 0 && kwi + 1 < KWP) load_a(kwi + 1, 0);       // the other register set
                        }
#pragma unroll
                        for (int kh = 0; kh < KH; ++kh) {
                            const int rr = hr - kh;
                            if (rr < 0 || rr >= RW) continue;
                            const int pg = rr * 2 + cc;
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) {
#ifdef CV2_X_NOMFMA
                                asm volatile("" :: "v"(__builtin_bit_cast(bf16x8, fa[KH == 1 ? (kwi & 1) : kh][ct])), "v"(__builtin_bit_cast(bf16x8, fb[st % RING])));
#else
                                acc[ct][pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, fa[KH == 1 ? (kwi & 1) : kh][ct]), __builtin_bit_cast(bf16x8, fb[st % RING]), acc[ct][pg], 0, 0, 0);
#endif
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);      // keep the fragment requests DEPTH steps ahead of their MFMAs
                    }
                };
                steps(std::true_type{});
                STAMP(7);
                if (late_ph >= 0) { refill(late_ph); late_ph = -1; }      // waves 4-7: the refill they owe
                STAMP(8);
                steps(std::false_type{});
                // ---- the next phase's weights (and, before a new chunk, its halo) must have landed; younger
                // requests stay in flight across the barrier -----------------------------------------------
                STAMP(2);
                if (producer) {
                    int need = ws ? w_mark0 : w_mark1;
                    if (ph == NPH - 1) need = max(need, hs ? h_mark0 : h_mark1);
                    wait_landed(need);
                }
                STAMP(3);
#ifndef CV2_X_NOBAR
                __syncthreads();                        // everyone's pieces; and the stages just read are free
#endif
                STAMP(4);
                if constexpr (!SPEC) {
                    if (late) late_ph = ph;
                    else if (NPH == 1 && c + 1 == nchunks) { push_w(); h_owed = true; }     // halo after the epilogue
                    else refill(ph);
                }
                ws ^= 1;
                STAMP(5);
            }
            hs ^= 1;
        }

        // ---- epilogue: + bias, per-channel sum / sum-of-squares partials, store -----------------------
        const bool do_stats = a.stats != nullptr;
        float* red = (float*)(smem + C::OFF_RED);     // [wr][BCO][2]
        const int co_w = cur.cb * BCO + wc * 64;
        // FAST: the tile lies inside the output, plain store (no accumulate; ReLU allowed), and this wave's 64
        // channels go to one destination: every store is a wave-uniform base plus one per-lane offset
        constexpr bool fast = !GEN;
        auto epilogue = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            const bool second = a.csplit > 0 && co_w >= a.csplit;                       // wave-uniform when FAST
            char* ydst = second ? (char*)a.y2 : (char*)a.y;
            const int ld = second ? a.ldy2 : a.ldy;
            const int chan0 = second ? a.yoff2 + co_w - a.csplit : a.yoff + co_w;
            const size_t pix0 = ((size_t)cur.n * a.Hout + cur.h0 + wr * RW) * a.Wout + cur.w0;
            char* ybase = ydst + (pix0 * ld + chan0) * 2;                                // wave-uniform
            if constexpr (FAST) {
                // Lane-derived addresses of the epilogue are recomputed here from an opaque copy of the lane id: left
                // to itself hipcc hoists them out of the tile loop and then spills them around the main loop.
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int ge = lane_e >> 4, l15e = lane_e & 15;
                // + bias, in place (one float4 of biases per 16-channel tile; nothing to do for dgrad launches)
                if (a.bias != nullptr) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const float4 b4 = *(const float4*)(smem + C::OFF_BIAS + tseq * 1024 + (wc * 64 + ct * 16 + 4 * ge) * 4);
#pragma unroll
                        for (int pg = 0; pg < NPG; ++pg) {
                            acc[ct][pg][0] += b4.x; acc[ct][pg][1] += b4.y; acc[ct][pg][2] += b4.z; acc[ct][pg][3] += b4.w;
                        }
                    }
                }
                if constexpr (ACC) {
                    if (a.flags & 1) {
                        // the destination's values join the fp32 accumulators in the accumulator layout (lane = pixel l15,
                        // four channels 4*ge..+3 of each 16-channel tile: 8 bytes per load, a pixel's four ge-lanes cover 32
                        // contiguous bytes), so the sum is rounded once, as in the generic kernel
                        const char* obase = ybase + ((size_t)l15e * ld + 4 * ge) * 2;
                        const size_t orow = (size_t)a.Wout * ld * 2, ohalf = (size_t)16 * ld * 2;
#pragma unroll
                        for (int pg = 0; pg < NPG; ++pg) {
                            const char* op = obase + (pg >> 1) * orow + (pg & 1) * ohalf;
                            uint2 old[4];
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) old[ct] = *(const uint2*)(op + ct * 32);
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) {
                                acc[ct][pg][0] += __uint_as_float(old[ct].x << 16); acc[ct][pg][1] += __uint_as_float(old[ct].x & 0xffff0000u);
                                acc[ct][pg][2] += __uint_as_float(old[ct].y << 16); acc[ct][pg][3] += __uint_as_float(old[ct].y & 0xffff0000u);
                            }
                        }
                    }
                }
                // pass 1 (forward only): BN partial statistics.  Every lane stores its (sum, sum of squares) pair:
                // lanes other than the group's first aim at a scratch slot, which is cheaper than sixteen exec-mask
                // round trips.
                if (do_stats && (!DEFER_REG || last_tile)) {
                    // one address select per lane; the 16 pairs then go to base + immediate (the scratch slots of
                    // different lanes overlap, which is fine for values nobody reads, and stay on distinct banks)
                    char* sbase = l15e == 0 ? (char*)&red[(wr * BCO + wc * 64 + 4 * ge) * 2]
                                           : smem + (hs ^ 1) * C::H_BYTES + (wave + 16) * 1024 + lane_e * 8;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (DEFER_REG) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { s1[r] = run1[ct][r]; s2[r] = run2[ct][r]; }
                        }
                        if constexpr (BNR) {
                            // z in the accumulator layout: 8 bytes per lane and 16-pixel group (a pixel's four ge-lanes cover 32
                            // contiguous bytes).  MEASURED (round 3): bringing z in as whole 128-byte lines through a wave-private LDS
                            // transposition instead (one group ahead, scale / shift for all 16 channels in registers) costs 40 spilled
                            // registers in the 128-channel kernel and is no faster (3.88 against 3.76 ms over the 7 launches of a step).
                            const int chan = co_w + ct * 16 + 4 * ge;
                            const float4 psc = *(const float4*)(a.bn_scale + chan), psh = *(const float4*)(a.bn_shift + chan);
                            const float4 pmu = *(const float4*)(a.bn_mean + chan), pis = *(const float4*)(a.bn_invstd + chan);
                            const float sc_[4] = {psc.x, psc.y, psc.z, psc.w}, sh_[4] = {psh.x, psh.y, psh.z, psh.w};
                            const float mu_[4] = {pmu.x, pmu.y, pmu.z, pmu.w}, is_[4] = {pis.x, pis.y, pis.z, pis.w};
                            const char* zb = (const char*)a.bn_z + ((pix0 + l15e) * a.bn_ldz + a.bn_zoff + chan) * 2;
                            uint2 zq[NPG];
#pragma unroll
                            for (int pg = 0; pg < NPG; ++pg)
                                zq[pg] = *(const uint2*)(zb + ((size_t)(pg >> 1) * a.Wout + (pg & 1) * 16) * a.bn_ldz * 2);
#pragma unroll
                            for (int pg = 0; pg < NPG; ++pg) {
                                const float zf[4] = {__uint_as_float(zq[pg].x << 16), __uint_as_float(zq[pg].x & 0xffff0000u),
                                                     __uint_as_float(zq[pg].y << 16), __uint_as_float(zq[pg].y & 0xffff0000u)};
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float d = fmaf(zf[r], sc_[r], sh_[r]) > 0.f ? acc[ct][pg][r] : 0.f;
                                    s1[r] += d;
                                    s2[r] = fmaf(d, (zf[r] - mu_[r]) * is_[r], s2[r]);
                                }
                            }
                        } else {
#pragma unroll
                        for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float v = acc[ct][pg][r]; s1[r] += v; s2[r] = fmaf(v, v, s2[r]); }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float2 t;
                            t.x = wave_sum16(s1[r]);
                            t.y = wave_sum16(s2[r]);
                            if constexpr (DEFER_LDS) {           // running sums of this wave's tiles (scratch lanes add garbage to garbage)
                                const float2 old = *(const float2*)(sbase + (ct * 16 + r) * 8);
                                t.x += old.x; t.y += old.y;
                            }
                            *(float2*)(sbase + (ct * 16 + r) * 8) = t;
                        }
                    }
                } else if (DEFER_REG && do_stats) {
#pragma unroll
                    for (int ct = 0; ct < (DEFER_REG ? 4 : 1); ++ct)
#pragma unroll
                        for (int pg = 0; pg < NPG; ++pg)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = acc[ct][pg][r];
                                run1[ct][r] += v; run2[ct][r] = fmaf(v, v, run2[ct][r]);
                            }
                }
                // pass 2: every 16-pixel group goes through a wave-private LDS tile [16 px][64 co] (16-byte
                // slots XORed with px & 7) and leaves as two stores of 8 pixels x 128 contiguous bytes:
                // whole cache lines, instead of 64 scattered 8-byte pieces per instruction.
                // (hs was toggled after the chunk loop: the stage just read is hs ^ 1)
                char* scr = smem + (hs ^ 1) * C::H_BYTES + wave * 1024;
                const int wpx = l15e, rpx = lane_e >> 3, rch = lane_e & 7;
                const unsigned st_off = (unsigned)(rpx * ld + rch * 8) * 2u;
                const size_t row_stride = (size_t)a.Wout * ld * 2, half_stride = (size_t)8 * ld * 2;
                char* wbase = scr + (wpx >> 3) * 8192 + (wpx & 7) * 128 + (ge & 1) * 8;
                const char* rbase = scr + rpx * 128 + ((rch ^ rpx) << 4);
                auto pass2 = [&](auto relu_tag, auto nt_tag) {
                    constexpr bool RELU = decltype(relu_tag)::value;            // deployed / BN-folded inference convs
                    constexpr bool NT = decltype(nt_tag)::value;                // streaming stores (flags bit 3, set by the launcher)
#pragma unroll
                    for (int pg = 0; pg < NPG; ++pg) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
                            float v0 = acc[ct][pg][0], v1 = acc[ct][pg][1], v2 = acc[ct][pg][2], v3 = acc[ct][pg][3];
                            if (RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                            uint2 pk;
                            pk.x = cvt_pk_bf16(v0, v1);
                            pk.y = cvt_pk_bf16(v2, v3);
                            *(uint2*)(wbase + (((ct * 2 + (ge >> 1)) ^ (wpx & 7)) << 4)) = pk;
                        }
                        if (pg == 0) STAMP(9);
                        if (pg == NPG / 2) STAMP(10);
                        char* pp = ybase + (pg >> 1) * row_stride + (pg & 1) * 2 * half_stride + st_off;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint4 val = *(const uint4*)(rbase + h * 8192);
                            if constexpr (NT) {
                                typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
                                __builtin_nontemporal_store(u32x4_nt{val.x, val.y, val.z, val.w}, (u32x4_nt*)(pp + h * half_stride));
                            } else {
                                *(uint4*)(pp + h * half_stride) = val;
                            }
                        }
                    }
                };
                if (a.flags & 8) { if (a.flags & 2) pass2(std::true_type{}, std::true_type{}); else pass2(std::false_type{}, std::true_type{}); }
                else { if (a.flags & 2) pass2(std::true_type{}, std::false_type{}); else pass2(std::false_type{}, std::false_type{}); }
            } else {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int co = co_w + ct * 16 + 4 * g;
                const bool co_ok = co < a.Cout;
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias != nullptr) {        // staged in LDS with the tile's first weights (zeros past Cout)
                    const float4 bv = *(const float4*)(smem + C::OFF_BIAS + tseq * 1024 + (wc * 64 + ct * 16 + 4 * g) * 4);
                    b4[0] = bv.x; b4[1] = bv.y; b4[2] = bv.z; b4[3] = bv.w;
                }
#pragma unroll
                for (int pg = 0; pg < NPG; ++pg) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[ct][pg][r] + b4[r];
                    {
                        const int ho = cur.h0 + wr * RW + (pg >> 1), wo = cur.w0 + (pg & 1) * 16 + l15;
                        if (co_ok && ho < a.Hout && wo < a.Wout) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { s1[r] += v[r]; s2[r] += v[r] * v[r]; }
                            const size_t pix = ((size_t)cur.n * a.Hout + ho) * a.Wout + wo;
                            __bf16* dst = (__bf16*)a.y + pix * a.ldy + a.yoff + co;
                            if (a.csplit > 0 && co >= a.csplit) dst = (__bf16*)a.y2 + pix * a.ldy2 + a.yoff2 + (co - a.csplit);
                            if (a.flags & 1) {
                                const uint2 old = *(const uint2*)dst;
                                v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                                v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
                            }
                            if (a.flags & 2) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                            }
                            uint2 pk;
                            pk.x = Vec16<__bf16>::pk(v[0], v[1]);
                            pk.y = Vec16<__bf16>::pk(v[2], v[3]);
                            *(uint2*)dst = pk;
                        }
                    }
                }
                if (do_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float t1 = wave_sum16(s1[r]);
                        const float t2 = wave_sum16(s2[r]);
                        if (l15 == 0) {
                            red[(wr * BCO + wc * 64 + ct * 16 + 4 * g + r) * 2 + 0] = t1;
                            red[(wr * BCO + wc * 64 + ct * 16 + 4 * g + r) * 2 + 1] = t2;
                        }
                    }
                }
            }
            }
        };
#ifdef CV2_X_NOEPI
        {   // timing experiment: the accumulators stay live, nothing is stored
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NPG; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            if (t == 1234.5f) *(float*)a.y = t;
        }
#else
        epilogue(std::integral_constant<bool, fast>{});
#endif
        if (h_owed) { push_h(); h_owed = false; }
        if (DEFER && do_stats && !last_tile) {
            // nothing to write: this tile's sums ride in the workgroup's one statistics row (written on its last tile)
            if constexpr (SPEC) __syncthreads();        // the request waves may refill the halo stage the epilogue borrowed
        } else if (do_stats) {
            __syncthreads();
            if (tid < BCO * 2) {
                const int ch = tid >> 1, k = tid & 1;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < C::NWR; ++w) s += red[(w * BCO + ch) * 2 + k];
                const int row = DEFER ? lblock : cur.pt;            // deferred statistics: one row per workgroup
                if (cur.cb * BCO + ch < a.Cout) a.stats[((size_t)row * a.Cout + cur.cb * BCO + ch) * 2 + k] = s;
            }
            __syncthreads();
        } else if (SPEC) {
            __syncthreads();            // the request waves may refill the halo stage the epilogue borrowed
        }
        STAMP(6);
        tseq ^= 1;
        if constexpr (DYN) tile_i = ring_at(++tile_sn); else tile_i += gridDim.x;
        if (tile_i >= total_tiles) break;
        cur = decode(tile_i);
    }
    if constexpr (DYN) { if (wave == 7) tq_done(); }
#ifdef CV2_TRACE
    if (trc && wave == 0) { g_cv2_trace[2050] = __builtin_amdgcn_s_memtime(); g_cv2_trace[2051] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <int WC, bool GEN, bool DST = false, bool SPEC = false, int KH = 3, int KW = 3, bool ACC = false, bool DYN = false, bool BNR = false>
static int launch_conv2(ConvArgs a, hipStream_t stream) {
    using C = Cv2Cfg<WC, KH, KW>;
    constexpr int LDS = C::SMEM + (DYN ? 64 : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static DynLdsOnce lds_once;
    if (int rc = ensure_dyn_lds(lds_once, (const void*)conv3x3_v2_kernel<WC, GEN, DST, SPEC, KH, KW, ACC, DYN>, LDS)) return rc;
    a.tiles_h = (a.Hout + C::TH - 1) / C::TH; a.tiles_w = (a.Wout + C::TW - 1) / C::TW;
    // STREAMING STORES for outputs far larger than the L2s (8 x 4 MiB): the store burst of one round of tiles (32 workgroups
    // x 128 KiB per XCD) otherwise evicts the weights and halos the next round wants.  Same box: 256->256 @128^2 (268 MB
    // out) forward -4 %, dgrad -3 %; 1024->1024 @32^2 (67 MB) +1.5 % with them, hence the threshold (nt_threshold_bytes(), 100 MiB); whole step -0.25 ms.
    if ((size_t)a.N * a.Hout * a.Wout * a.Cout * 2 >= nt_threshold_bytes()) a.flags |= 8;
    const int nco = (a.Cout + C::BCO - 1) / C::BCO;
    const int total = a.N * a.tiles_h * a.tiles_w * nco;
    const int grid = total < 256 ? total : 256;            // one persistent workgroup per CU
    UNET_CHECK((unsigned long long)total * (unsigned)(nco > a.tiles_w ? (nco > a.tiles_h ? nco : a.tiles_h) : (a.tiles_w > a.tiles_h ? a.tiles_w : a.tiles_h)) < (1ull << 32),
               "tile count too large for the index arithmetic");
    if constexpr (!DYN && !GEN && !DST && !ACC && !BNR && KH == 3 && KW == 3 && (WC == 2 || SPEC)) {
        // tile queues (tile_queue.h) for launches with at least three rounds of tiles (two are drawn up front)
        if (grid == 256 && a.Cin >= 64 && total >= (tileq_level() == 2 ? 256 : tileq_level() == 1 ? 3 * 256 : INT_MAX)
            && !tileq_capturing(stream)) {
            a.tq = tileq_next_slot();
            if (a.tq == nullptr) return -3;
            return launch_conv2<WC, GEN, DST, SPEC, KH, KW, ACC, true>(a, stream);
        }
    }
    hipLaunchKernelGGL((conv3x3_v2_kernel<WC, GEN, DST, SPEC, KH, KW, ACC, DYN, BNR>), dim3((unsigned)grid), dim3(SPEC ? 768 : 512), LDS, stream, a, nco, total,
                       make_fastdiv((unsigned)nco), make_fastdiv((unsigned)a.tiles_w), make_fastdiv((unsigned)a.tiles_h));
    UNET_LAUNCH_CHECK();
    return 0;
}

static bool conv2_gen(int Cin, int Cout, int Hout, int Wout, int flags, int csplit) {
    const int bco = Cout > 64 ? 128 : 64;
    return (Cin & 31) != 0 || (Cout % bco) != 0 || (Hout & 15) != 0 || (Wout & 31) != 0 || (csplit & 63) != 0;
}
// deferred statistics (specialised 64-channel kernel, statistics wanted, not per-image rows): ONE row per workgroup
static bool conv2_defer(int Cin, int Cout, int Hout, int Wout, int flags, int csplit) {
    return Cout <= 64 && !conv2_gen(Cin, Cout, Hout, Wout, flags, csplit) && (flags & 5) == 0;
}
// stat rows the bf16 3x3 path writes: one per 16x32 pixel tile, or one per persistent workgroup (deferred statistics)
int conv2_rows(int N, int Hout, int Wout, int Cin, int Cout, int flags, int csplit) {
    const int tiles = N * ((Hout + 15) / 16) * ((Wout + 31) / 32);
    if (conv2_defer(Cin, Cout, Hout, Wout, flags, csplit)) return tiles < 256 ? tiles : 256;
    return tiles;
}

// MEASURED AND NOT KEPT (round 2, same-box A/B against this kernel):
//  * 256-thread workgroups on 8x32-pixel tiles, two per CU, for the HBM-bound 64-channel layers (one workgroup's store
//    burst under the other's MFMAs): correct, 6-12 % SLOWER (64->64 @512^2 forward 0.86 vs 0.78 ms).  The weights are
//    re-streamed per tile, so halving the tile doubles the LDS-DMA instructions per MFMA (29 vs 19 per wave and tile);
//  * producer scheme with the refill SPREAD over the phase (the barrier only plans it; one or two requests per MFMA step
//    instead of one burst while the partner has the matrix pipe): correct, 4-5 % SLOWER on every 128-channel-tile launch
//    (256->256 @128^2 forward 0.527 -> 0.547 ms): a request issued between a wave's own MFMAs stalls that wave's MFMA
//    stream for longer than the burst costs;
//  * output channels permuted in the weight image so that a lane's accumulators are two 8-channel runs of its pixel and
//    the epilogue stores 2 x 16 bytes per lane straight from registers (no LDS transposition): correct, 0-4 % SLOWER --
//    64-byte runs per pixel and instruction cost more at the memory side than the whole 128-byte lines of the transposed
//    form (128-channel dgrad into a 2.1 GB tensor: 1.106 -> 1.147 ms).
int dispatch_conv2(const ConvArgs& a, hipStream_t s) {
    const bool gen = conv2_gen(a.Cin, a.Cout, a.Hout, a.Wout, a.flags, a.csplit);
    if (!gen && (a.flags & 1)) return a.Cout > 64 ? launch_conv2<2, false, false, false, 3, 3, true>(a, s) : launch_conv2<1, false, false, false, 3, 3, true>(a, s);
    if (a.Cout > 64) return gen ? launch_conv2<2, true>(a, s) : launch_conv2<2, false>(a, s);
    if (gen) return launch_conv2<1, true>(a, s);
    // deferred statistics put a workgroup's sums (over tiles of several images) into one row: not for flags bit 2
    // (per-tile rows grouped per image, what GroupNorm's finalize reads); conv2_rows() tells the host the row count.
    // Everything else on 64-channel tiles (the dgrad launches, GroupNorm forwards) runs with request waves: 64->64 @512^2
    // dgrad 0.594 -> 0.554 ms on one box.  (The same kernel on the 128...1024-channel layers, 64 output channels per
    // workgroup, is 2-4 % slower than the 128-channel-tile kernel: 1.35x the DMA bytes per MFMA; with the deferred
    // statistics' 32 extra registers it does not fit three waves per SIMD.)
    // The statistics-carrying forward of the 64-channel layers has request waves as well (round 3; UNET_AMD_DST_NOSPEC=1 selects
    // the round-2 form with the sums in registers and no request waves, for A/B runs).
    static const bool dst_nospec = [] { const char* e = getenv("UNET_AMD_DST_NOSPEC"); return e && atoi(e) != 0; }();
    if (a.stats != nullptr && conv2_defer(a.Cin, a.Cout, a.Hout, a.Wout, a.flags, a.csplit))
        return dst_nospec ? launch_conv2<1, false, true>(a, s) : launch_conv2<1, false, true, true>(a, s);
    return launch_conv2<1, false, false, true>(a, s);
}

// 3x3 dgrad + BatchNorm-backward sums (BNR): specialised shapes only; one statistics row per 16x32 pixel tile
bool conv2_bnred_ok(int Cin, int Cout, int Hout, int Wout) { return !conv2_gen(Cin, Cout, Hout, Wout, 0, 0) && (Cout & 63) == 0; }
int dispatch_conv2_bnred(const ConvArgs& a, hipStream_t s) {
    return a.Cout > 64 ? launch_conv2<2, false, false, false, 3, 3, false, false, true>(a, s)
                       : launch_conv2<1, false, false, true, 3, 3, false, false, true>(a, s);
}

// 5x5 stride 1 on whole 16x32 tiles and whole 64-channel blocks: the LDS-DMA kernel (one stats row per pixel tile); everything
// else stays on conv_igemm.hip (8x32 tiles).  unet_conv_tiles() asks the same question to size the statistics rows.
bool conv5_v2_ok(int Cin, int Cout, int Hout, int Wout, int flags, int csplit) {
    return (Cin & 31) == 0 && (Cout & 63) == 0 && (Hout & 15) == 0 && (Wout & 31) == 0 && (csplit & 63) == 0;
}
int conv5_rows(int N, int Hout, int Wout) { return N * (Hout / 16) * (Wout / 32); }
int dispatch_conv5(const ConvArgs& a, hipStream_t s) {
    return (a.flags & 1) ? launch_conv2<1, false, false, false, 5, 5, true>(a, s) : launch_conv2<1, false, false, false, 5, 5>(a, s);
}
// k x 1 / 1 x k, same eligibility and row count
int dispatch_convk(int kh, int kw, const ConvArgs& a, hipStream_t s) {
    const bool acc = (a.flags & 1) != 0;
    if (kh == 3 && kw == 1) return acc ? launch_conv2<1, false, false, false, 3, 1, true>(a, s) : launch_conv2<1, false, false, false, 3, 1>(a, s);
    if (kh == 5 && kw == 1) return acc ? launch_conv2<1, false, false, false, 5, 1, true>(a, s) : launch_conv2<1, false, false, false, 5, 1>(a, s);
    if (kh == 1 && kw == 3) return acc ? launch_conv2<1, false, false, false, 1, 3, true>(a, s) : launch_conv2<1, false, false, false, 1, 3>(a, s);
    if (kh == 1 && kw == 5) return acc ? launch_conv2<1, false, false, false, 1, 5, true>(a, s) : launch_conv2<1, false, false, false, 1, 5>(a, s);
    unet_set_error("no LDS-DMA instantiation for a %dx%d kernel", kh, kw);
    return -1;
}
