// Host side of the dynamic tile queues (tile_queue.h): the slot array and the switch.
#include "common.h"
#include "tile_queue.h"
#include <atomic>
#include <cstdlib>

static __device__ __align__(256) unsigned g_tileq[TQ_NSLOTS * TQ_SLOT_WORDS];      // zero-initialised; slots clean themselves

static std::atomic<int> g_dyn{-1};
int tileq_level() {
    int v = g_dyn.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("UNET_AMD_DYNAMIC_TILES");
        v = e ? atoi(e) : 0;
        v = v < 0 ? 0 : v > 2 ? 2 : v;
        g_dyn.store(v, std::memory_order_relaxed);
    }
    return v;
}
bool tileq_enabled() { return tileq_level() > 0; }

// 1: persistent conv / weight-gradient kernels draw their tiles from atomic queues (for runs that share the GPU with
// collectives: unet_amd.parallel.DataParallel does not switch it on by itself); 0: static partition (default);
// 2: as 1 and also for launches with few tiles per workgroup (the tests use it to reach the stealing path at small sizes).
// No counterpart in the reference (it has no kernels of its own).
extern "C" int unet_set_dynamic_tiles(int on) {
    g_dyn.store(on < 0 ? 0 : on > 2 ? 2 : on, std::memory_order_relaxed);
    return 0;
}

bool tileq_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return true; }     // cannot tell: stay static
    return st != hipStreamCaptureStatusNone;
}

unsigned* tileq_next_slot() {
    static std::atomic<unsigned*> base[16];
    static std::atomic<unsigned> seq[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { unet_set_error("tile queue: bad device"); return nullptr; }
    unsigned* b = base[dev].load(std::memory_order_acquire);
    if (b == nullptr) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tileq)) != hipSuccess || p == nullptr) { unet_set_error("tile queue: hipGetSymbolAddress failed"); return nullptr; }
        b = (unsigned*)p;
        base[dev].store(b, std::memory_order_release);
    }
    const unsigned s = seq[dev].fetch_add(1, std::memory_order_relaxed) % TQ_NSLOTS;
    return b + (size_t)s * TQ_SLOT_WORDS;
}
