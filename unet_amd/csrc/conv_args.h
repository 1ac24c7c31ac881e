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
    // BNR instantiations of conv_fwd2.hip (unet_conv_dgrad_bnred): the output is the gradient dy in front of a BatchNorm + ReLU
    // whose pre-norm tensor is bn_z (bf16, same pixels and channels as y); the statistics rows then carry that BatchNorm's
    // backward sums (sum dy', sum dy' * xhat) instead of (sum y, sum y^2)
    const void* bn_z;
    int bn_ldz, bn_zoff;
    const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
};
