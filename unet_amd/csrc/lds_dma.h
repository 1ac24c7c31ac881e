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
