// Shared device helpers for the CP-CSV gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CPCSV_F32 0
#define CPCSV_BF16 1

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, same as torch's float -> bfloat16
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int per16 = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
    static constexpr int per16 = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.f;               // ReLU
        case 2: return v > 0.f ? v : 0.2f * v;          // LeakyReLU(0.2)
        case 3: return tanhf(v);                        // Tanh
        case 4: return 1.f / (1.f + expf(-v));          // Sigmoid
        default: return v;
    }
}
// The same in element loops: the piecewise-linear activations (none / ReLU / LeakyReLU - everything behind a BatchNorm and in the
// towers) as a compare + select on per-launch constants, the smooth ones (SMOOTH, a template argument of the kernel) through the
// switch above. A run-time switch per element costs a scalar branch tree per element (bn_bwd_reduce ran at half its bandwidth).
struct ActPl { float slope; bool relu; };
__device__ __forceinline__ ActPl act_pl(int act) { return ActPl{act == 1 ? 0.f : (act == 2 ? 0.2f : 1.f), act == 1}; }
template <bool SMOOTH>
__device__ __forceinline__ float act_apply_t(float v, int act, const ActPl& a) {
    if (SMOOTH) return act_apply(v, act);
    return v > 0.f ? v : (a.relu ? 0.f : a.slope * v);
}
// BatchNorm's per-channel affine map and its application, with the roundings PINNED (explicit mul / fma, no contraction choices):
// the backward kernels recompute the pre-activation from the saved conv output with exactly the forward's arithmetic, so their
// activation mask is the forward's mask bit for bit (the reference takes it from the stored output). Recomputed as
// gamma * ((x - mean) * invstd) + beta it agreed only to an ulp, and an output within an ulp of zero got one mask forward and the
// other backward.
__device__ __forceinline__ void bn_affine(float ga, float be, float mean, float invstd, float& scale, float& shift) {
    scale = __fmul_rn(ga, invstd);
    shift = __fmaf_rn(-mean, scale, be);
}
__device__ __forceinline__ float bn_pre(float x, float scale, float shift) { return __fmaf_rn(x, scale, shift); }
// running statistic  r <- (1 - m) r + m b  (nn.BatchNorm, momentum m) with its roundings pinned as well: every kernel that advances
// running_mean / running_var goes through this, so two kernel paths of one layer leave bit-identical buffers
__device__ __forceinline__ float bn_running(float r, float b, float mom) { return __fmaf_rn(mom, b, __fmul_rn(1.f - mom, r)); }
// d act / d pre-activation, expressed with the POST-activation value y
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return y > 0.f ? 1.f : 0.2f;
        case 3: return 1.f - y * y;
        case 4: return y * (1.f - y);
        default: return 1.f;
    }
}

#define CPCSV_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return -(int)e__;     \
    } while (0)

// 1 = reproducible reductions (cpcsv_set_deterministic): every floating-point sum has ONE fixed order, no atomics
// between blocks. Defined in small.hip.
extern int g_cpcsv_deterministic;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
