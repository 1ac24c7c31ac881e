// DYNAMIC TILE QUEUES for the persistent matrix kernels (data-parallel runs only; UNET_AMD_DYNAMIC_TILES=1 or
// unet_set_dynamic_tiles(1)).
//
// The persistent conv / wgrad kernels own every CU and split their tiles statically.  A collective's workgroups that hold a
// few CUs while such a kernel is launched therefore delay it by their whole hold time (tools/contention_probe.py: 3 x 1 ms of
// hold cost 2.7 ms per step).  With these queues a workgroup draws its next tile from an atomic counter instead, so a CU that
// starts late simply takes fewer tiles.  One counter per XCD (per XCD and channel block for the weight gradient) keeps the
// tiles of concurrently running workgroups adjacent in that XCD's L2, as the static order does; a workgroup whose own queue
// is dry steals from the fullest other one.  Which workgroup computes which tile is then timing dependent: deferred BN
// statistics and split-K partials are summed in a different grouping from run to run (last-bit differences), which is why
// the static order stays the default.
//
// A launch takes one SLOT of counters out of a device array (round robin, 1024 slots); the last workgroup to finish zeroes
// it again.  Slots are reused after 1024 launches, long after their kernel has ended.
#pragma once
#include <hip/hip_runtime.h>

constexpr int TQ_SLOT_WORDS = 160;          // up to 8 x 16 counters, then `done`
constexpr int TQ_DONE = 128;
constexpr int TQ_NSLOTS = 1024;
constexpr int TQ_INVALID = 0x7fffffff;

bool tileq_enabled();                       // UNET_AMD_DYNAMIC_TILES / unet_set_dynamic_tiles
int tileq_level();                          // 0 off, 1 on for launches with several rounds of tiles, 2 on wherever the kernels allow
bool tileq_capturing(hipStream_t stream);   // a slot is bound at launch time, so a captured graph would replay one slot from every replay (two replays in
                                            // flight would share its counters): launches under stream capture keep the static partition
unsigned* tileq_next_slot();                // device pointer to a zeroed slot on the current device (nullptr on failure)
