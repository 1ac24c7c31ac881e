// LDS-DMA helpers shared by the second-generation bf16 kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes
// [lds_dst, lds_dst + 1024) in lane order (lds_dst wave-uniform).  Written as inline asm on purpose:
// hipcc makes every later ds_read wait vmcnt(0) for a DMA it knows about, which would drain the
// prefetch of the next stage at once; this way the kernel's own s_waitcnt before the stage barrier
// is the only wait.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// The same through a buffer descriptor: 64 lanes x 16 bytes from rsrc.base + voff (per lane) to LDS.
// A lane whose voff is >= rsrc.num_records reads nothing and writes ZEROS: that is how image borders,
// ragged tiles and channel tails are padded without a per-lane pointer select.  The leading s_nop
// covers the SALU-write -> VMEM-descriptor-read hazard that hipcc cannot see inside the statement.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned num_records) {
    const unsigned long long p = (unsigned long long)base;
    i32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r.z = (int)num_records;
    r.w = 0x00020000;
    return r;
}
#define LDS_DMA_OOB 0xFFFFFFF0u          /* voff of a padded lane */
#define LDS_DMA_RECORDS 0xFFFFFF00u      /* num_records of descriptors whose base is a tile origin */
__device__ __forceinline__ void lds_dma16_buf(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

// Division of small wave-uniform unsigned numbers by a launch constant without the float sequence hipcc
// emits for a runtime divisor: q = mulhi(x, ceil(2^32 / d)), exact while x * d < 2^32 (the launchers
// check); d == 1 is encoded as magic 0.
struct FastDiv { unsigned d, magic; };
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f; f.d = d;
    f.magic = d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d);
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned x, FastDiv f) { return f.magic ? __umulhi(x, f.magic) : x; }

// Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Give every XCD a contiguous run
// of LOGICAL block ids instead, so that blocks working on the same pixels (other channel blocks) or on
// neighbouring tiles share one L2 rather than fetching the same lines into eight of them.
__device__ __forceinline__ int xcd_block_id() {
    const int g = (int)gridDim.x, b = (int)blockIdx.x;
    return (g & 7) ? b : (b & 7) * (g >> 3) + (b >> 3);
}

// Outputs at least this large are written with streaming (nontemporal) stores by the persistent conv kernels: the store
// burst of one round of tiles otherwise evicts the operands of the next from the 8 x 4 MiB of L2.  UNET_AMD_NT_MB overrides
// the 100 MiB default (experiments).
#include <cstdlib>
static inline size_t nt_threshold_bytes() {
    static const size_t v = [] { const char* e = getenv("UNET_AMD_NT_MB"); const long mb = e ? atol(e) : 100; return (size_t)(mb > 0 ? mb : 100) << 20; }();
    return v;
}
