// Argument block shared by the implicit-GEMM convolution kernels (conv_igemm.hip, conv_fwd2.hip).
#pragma once
struct ConvArgs {
    const void* x;
    const void* w;
    const float* bias;
    void* y;
    float* stats;
    int N, Hin, Win, Cin, ldx, xoff;
    int Hout, Wout, Cout, ldy, yoff;
    int pad_h, pad_w, tiles_h, tiles_w;
    // optional second destination: output channels >= csplit go to y2 at channel (co - csplit)
    void* y2;
    int ldy2, yoff2, csplit;
    int flags;      // bit 0: y += conv (accumulate into the existing output), bit 1: ReLU on the result
    unsigned* tq;   // dynamic tile queue slot (tile_queue.h) or NULL; set by the launcher
};
