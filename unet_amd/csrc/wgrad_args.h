// Argument block shared by the weight-gradient kernels (conv_wgrad.hip: exact f32; conv_wgrad2.hip: bf16).
#pragma once
struct WgradArgs {
    const void* dz;   // [N,Hout,Wout,lddz] (+dzoff), Cout channels
    const void* x;    // [N,Hin,Win,ldx]   (+xoff),  Cin channels
    float* slab;      // [nsplit][KH*KW][Cout][Cin]
    int N, Hin, Win, Cin, ldx, xoff;
    int Hout, Wout, Cout, lddz, dzoff;
    int pad_h, pad_w, tiles_h, tiles_w;
    int tiles_total, tiles_per_split;
    unsigned* tq;     // dynamic tile queue slot (tile_queue.h), or NULL: static split of the tiles
};
