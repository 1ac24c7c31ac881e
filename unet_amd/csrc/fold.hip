// RepUNet re-parameterisation fold (gfx950), BIT-EXACT against the reference's fp32 CPU result.
// Built with -ffp-contract=off: every multiply, add, divide and square root below is a separately
// rounded IEEE fp32 operation in the reference's order.
//
// Reference: mergeBN (/root/reference/models/reparm_unet.py:28-31), kernelFuse (:33-37),
// AsymmConvBlock.mergeAsyKernels (:74-82), MultiConvBlock.mergeMulKernels (:124-135).
//
//   std = sqrt(running_var + eps);  t = weight / std;  W' = W * t;  b' = bias - ((mean * weight) / std)
//   w_sq += w_ve (centre column)  then  w_sq += w_ho (centre row);   b = (b_sq + b_ve) + b_ho
//   across kernel sizes: target = clone(largest); target[centre kxk] += smaller, ascending; bias likewise
#include "common.h"

// std: optional per-channel sqrt(running_var + eps) computed by the caller (unet_fold_asymm_std); NULL = computed here
struct BNv { const float* w; const float* b; const float* rm; const float* rv; const float* std; };

// Correctly rounded fp32 sqrt and divide: evaluate in fp64 (IEEE-correct on gfx950) and round once more to
// fp32.  With 53 >= 2*24+2 significand bits the double rounding is innocuous for sqrt and divide, so the
// result equals the CPU's correctly rounded sqrtps/divps bit for bit (the fp32 device sqrt is not always).
__device__ __forceinline__ float sqrt_cr(float x) { return (float)sqrt((double)x); }
__device__ __forceinline__ float div_cr(float a, float b) { return (float)((double)a / (double)b); }

__device__ __forceinline__ float bn_t(const BNv& bn, int c, float eps) {
    const float std_ = bn.std != nullptr ? bn.std[c] : sqrt_cr(bn.rv[c] + eps);
    return div_cr(bn.w[c], std_);
}
__device__ __forceinline__ float bn_b(const BNv& bn, int c, float eps) {
    const float std_ = bn.std != nullptr ? bn.std[c] : sqrt_cr(bn.rv[c] + eps);
    return bn.b[c] - div_cr(bn.rm[c] * bn.w[c], std_);
}

// one AsymmConvBlock: out_w (O,I,k,k), out_b (O)
__global__ void fold_asymm_kernel(int k, int O, int I, const float* __restrict__ w_sq, const float* __restrict__ w_ve,
                                  const float* __restrict__ w_ho, BNv bn_sq, BNv bn_ve, BNv bn_ho, float eps,
                                  float* __restrict__ out_w, float* __restrict__ out_b) {
    const size_t total = (size_t)O * I * k * k;
    const int c = k / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int kw = (int)(e % k); size_t r = e / k;
        const int kh = (int)(r % k); r /= k;
        const int i = (int)(r % I);
        const int o = (int)(r / I);
        float v = w_sq[e] * bn_t(bn_sq, o, eps);
        if (kw == c) v = v + w_ve[((size_t)o * I + i) * k + kh] * bn_t(bn_ve, o, eps);     // (k,1) kernel: centre column
        if (kh == c) v = v + w_ho[((size_t)o * I + i) * k + kw] * bn_t(bn_ho, o, eps);     // (1,k) kernel: centre row
        out_w[e] = v;
    }
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)O; o += (size_t)gridDim.x * blockDim.x)
        out_b[o] = (bn_b(bn_sq, (int)o, eps) + bn_b(bn_ve, (int)o, eps)) + bn_b(bn_ho, (int)o, eps);
}

// target (O,I,K,K) [centre k x k] += src (O,I,k,k);  target_b += src_b
__global__ void fold_embed_add_kernel(int K, int k, int O, int I, float* __restrict__ tw, const float* __restrict__ sw,
                                      float* __restrict__ tb, const float* __restrict__ sb) {
    const size_t total = (size_t)O * I * k * k;
    const int off = K / 2 - k / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int kw = (int)(e % k); size_t r = e / k;
        const int kh = (int)(r % k); r /= k;        // r = o*I + i
        const size_t d = (r * K + (kh + off)) * K + (kw + off);
        tw[d] = tw[d] + sw[e];
    }
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)O; o += (size_t)gridDim.x * blockDim.x)
        tb[o] = tb[o] + sb[o];
}

static int fold_asymm_launch(int k, int O, int I, const float* w_sq, const float* w_ve, const float* w_ho,
                             const BNv& a, const BNv& b, const BNv& c, float eps, float* out_w, float* out_b, void* stream) {
    UNET_CHECK(k >= 1 && (k & 1) && O > 0 && I > 0, "fold: kernel size must be odd");
    UNET_CHECK(w_sq && w_ve && w_ho && out_w && out_b && a.w && a.b && a.rm && a.rv && b.w && b.b && b.rm && b.rv && c.w && c.b && c.rm && c.rv,
               "fold: null argument");
    const size_t total = (size_t)O * I * k * k;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fold_asymm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, k, O, I, w_sq, w_ve, w_ho, a, b, c, eps, out_w, out_b);
    UNET_LAUNCH_CHECK();
    return 0;
}

extern "C" int unet_fold_asymm(int k, int O, int I, const float* w_sq, const float* w_ve, const float* w_ho,
                               const float* sq_w, const float* sq_b, const float* sq_rm, const float* sq_rv,
                               const float* ve_w, const float* ve_b, const float* ve_rm, const float* ve_rv,
                               const float* ho_w, const float* ho_b, const float* ho_rm, const float* ho_rv,
                               float eps, float* out_w, float* out_b, void* stream) {
    return fold_asymm_launch(k, O, I, w_sq, w_ve, w_ho, BNv{sq_w, sq_b, sq_rm, sq_rv, nullptr}, BNv{ve_w, ve_b, ve_rm, ve_rv, nullptr},
                             BNv{ho_w, ho_b, ho_rm, ho_rv, nullptr}, eps, out_w, out_b, stream);
}

// The same with std = sqrt(running_var + eps) supplied per branch: the one step of the reference's fold that is not an
// IEEE-exact operation on its own host (torch's CPU sqrt is MKL VML, up to one ulp off), a vector of <= 1024 floats.
// A caller that hands in the reference host's own sqrt gets the reference's folded weights bit for bit.
extern "C" int unet_fold_asymm_std(int k, int O, int I, const float* w_sq, const float* w_ve, const float* w_ho,
                                   const float* sq_w, const float* sq_b, const float* sq_rm, const float* sq_rv, const float* sq_std,
                                   const float* ve_w, const float* ve_b, const float* ve_rm, const float* ve_rv, const float* ve_std,
                                   const float* ho_w, const float* ho_b, const float* ho_rm, const float* ho_rv, const float* ho_std,
                                   float* out_w, float* out_b, void* stream) {
    UNET_CHECK(sq_std && ve_std && ho_std, "fold_asymm_std: the three std vectors are required");
    return fold_asymm_launch(k, O, I, w_sq, w_ve, w_ho, BNv{sq_w, sq_b, sq_rm, sq_rv, sq_std}, BNv{ve_w, ve_b, ve_rm, ve_rv, ve_std},
                             BNv{ho_w, ho_b, ho_rm, ho_rv, ho_std}, 0.f, out_w, out_b, stream);
}

extern "C" int unet_fold_embed_add(int K, int k, int O, int I, float* target_w, const float* src_w, float* target_b,
                                   const float* src_b, void* stream) {
    UNET_CHECK(k <= K && (k & 1) && (K & 1), "fold: cannot embed %dx%d into %dx%d", k, k, K, K);
    const size_t total = (size_t)O * I * k * k;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(fold_embed_add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, K, k, O, I, target_w, src_w, target_b, src_b);
    UNET_LAUNCH_CHECK();
    return 0;
}
