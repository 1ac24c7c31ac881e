// Shared device helpers for the gfx950 (MI355X / CDNA4) U-Net kernels.
// Everything here is written for 64-wide wavefronts and the 16x16 MFMA family.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// dtype codes of the C ABI (include/unet_hip.h)
#define UNET_F32 0
#define UNET_BF16 1

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int EPV = 4;    // elements per 16-byte vector
    static constexpr int CKE = 16;   // elements per 64-byte K record
};
template <> struct TT<__bf16> {
    static constexpr int EPV = 8;
    static constexpr int CKE = 32;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// 16-byte vector of T <-> EPV floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y);
        f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Vec16<__bf16> {
    static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
        // bf16 -> f32 is a 16-bit shift
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint32_t pk(float lo, float hi) {
        __bf16 a = (__bf16)lo, b = (__bf16)hi;      // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    }
};

// Two floats -> packed bf16 pair (lo in bits 0-15), one instruction; RNE, NaN stays NaN.  Spelled as asm because
// hipcc otherwise SLP-pairs the conversions of a uint2 the other way round and shuffles halves afterwards.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// One 16x16 MFMA K-record step: 64 bytes of K per row (32 bf16 / 16 f32 channels).
// a: this lane's 16 bytes of A row (lane&15), k-slot (lane>>4); b likewise for B column (lane&15).
// D[row = 4*(lane>>4)+reg][col = lane&15].
template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<__bf16>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) {
    // exact f32 MFMA (k-ordered fmaf chain); k-slot g of step e is channel 4g+e for BOTH operands
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// 64-byte-record LDS image: row R holds 4 slots of 16 bytes; slot q is stored at q ^ 2*((R>>2)&1),
// which makes the 16-lane-group ds_read_b128 pattern (rows R..R+15, slot lane>>4) conflict-free.
__device__ __forceinline__ int rec64_off(int R, int q) { return R * 64 + ((q ^ (((R >> 2) & 1) << 1)) << 4); }

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum16(float v) {
    // sum over the 16 lanes that share lane>>4 (one DPP row): quad swaps, then the two mirrors
    v += dpp_f32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);     // row_half_mirror
    v += dpp_f32<0x140>(v);     // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wave_sum64d(double v) {
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// thread-local error text for the C ABI
void unet_set_error(const char* fmt, ...);
#define UNET_CHECK(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            unet_set_error(__VA_ARGS__);           \
            return -1;                             \
        }                                          \
    } while (0)
#define UNET_LAUNCH_CHECK()                                                        \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            unet_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return -2;                                                             \
        }                                                                          \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: remember per kernel instantiation
// which devices already have it (bit d = device d), so a process that drives several GPUs gets it on each of them.
// Thread-safe: two threads may both set the attribute once, which is harmless.
#include <atomic>
struct DynLdsOnce { std::atomic<unsigned long long> devices{0}; };
static inline int ensure_dyn_lds(DynLdsOnce& once, const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { unet_set_error("hipGetDevice failed"); return -2; }
    if (dev >= 0 && dev < 64 && ((once.devices.load(std::memory_order_relaxed) >> dev) & 1ull)) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { unet_set_error("hipFuncSetAttribute(%d bytes of LDS) failed on device %d: %s", bytes, dev, hipGetErrorString(e)); return -2; }
    if (dev >= 0 && dev < 64) once.devices.fetch_or(1ull << dev, std::memory_order_relaxed);
    return 0;
}

