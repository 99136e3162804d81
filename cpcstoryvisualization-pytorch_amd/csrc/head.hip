// head.hip — the critics' logit layer as three launches.
//
// D_GET_LOGITS ends in Conv2d(8*ndf, 1, kernel 4, stride 4) + Sigmoid over the 4x4 map (reference model.py:79-80): ONE output
// per sample, 16*992 weights, spectral-normed, biased. Through the general layer path its update costs ~25 launches per critic
// (GEMM + split-K pass + unpad forward; per reference call - real / wrong / fake, each with its own sigma - a weight-gradient
// GEMM, a dot, an unpack and a data-gradient GEMM backward), every one of them a few microseconds of pure launch latency on
// the critic's critical chain. Here: forward = one dot product per row; backward = dz and dX in one launch, the weight /
// bias gradients incl. the spectral-norm rank-1 terms in two.
//   x   [R][K] activations (dtype), K = taps * Cin_s: the flattened NHWC map, pads zero
//   w   [K] packed weight (cpcsv_pack_weight forward layout of the one output channel: k = tap * Cin_s + c)
//   groups: rows [row[g], row[g+1]) are reference call g with its own {sigma, 1/sigma}, u (1 value), v (Cin*taps values,
//   master order c*taps + t)
#include "common.h"
#include "../../include/cpcsv_hip.h"

namespace {

__device__ __forceinline__ float blk_sum(float v, float* sh) {
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}
__device__ __forceinline__ int group_of(const cpcsv_logit_groups& g, int r) {
    int k = 0;
    for (int i = 1; i < g.n; ++i) k += r >= g.row[i];
    return k;
}
__device__ __forceinline__ float inv_sigma(const cpcsv_logit_groups& g, int k) { return g.sigma[k] ? g.sigma[k][1] : 1.f; }

// p[r] = sigmoid(<x[r], w> / sigma_g + b): one block per row
template <typename T>
__global__ __launch_bounds__(256) void logit_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ p, int K, cpcsv_logit_groups g) {
    __shared__ float sh[4];
    const int r = blockIdx.x;
    const T* xr = x + (long)r * K;
    float acc = 0.f;
    constexpr int EPC = elem<T>::per16;
    for (int k = threadIdx.x * EPC; k < K; k += 256 * EPC) {           // K is a multiple of 8 (channel pads)
        const u32x4 xv = *reinterpret_cast<const u32x4*>(xr + k), wv = *reinterpret_cast<const u32x4*>(w + k);
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc += __uint_as_float(xv[e] << 16) * __uint_as_float(wv[e] << 16);
                acc += __uint_as_float(xv[e] & 0xffff0000u) * __uint_as_float(wv[e] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += __uint_as_float(xv[e]) * __uint_as_float(wv[e]);
        }
    }
    const float s = blk_sum(acc, sh);
    if (threadIdx.x == 0) {
        const float t = s * inv_sigma(g, group_of(g, r)) + (bias ? bias[0] : 0.f);
        p[r] = 1.f / (1.f + expf(-t));
    }
}

// dz[r] = dy[r] * p (1 - p);  dx[r][:] = dz[r] / sigma_g * w[:]  (dx may be NULL)
template <typename T>
__global__ __launch_bounds__(256) void logit_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ p, const T* __restrict__ w,
                                                        T* __restrict__ dx, float* __restrict__ dz, int K, cpcsv_logit_groups g) {
    const int r = blockIdx.x;
    const float pr = p[r], d = dy[r] * pr * (1.f - pr);
    if (threadIdx.x == 0) dz[r] = d;
    if (!dx) return;
    const float sc = d * inv_sigma(g, group_of(g, r));
    constexpr int EPC = elem<T>::per16;
    T* xr = dx + (long)r * K;
    for (int k = threadIdx.x * EPC; k < K; k += 256 * EPC) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(w + k);
        u32x4 o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(wv[e] << 16) * sc, hi = __uint_as_float(wv[e] & 0xffff0000u) * sc;
                o[e] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(__uint_as_float(wv[e]) * sc);
        }
        *reinterpret_cast<u32x4*>(xr + k) = o;
    }
}

// pass 1 of the weight gradient: G[g][k] = sum_{r in group g} dz[r] x[r][k] for a block's 256 columns; per-block partials of
// <G_g, w> go to dots[g][blockIdx.x] (summed in a fixed order by pass 2: deterministic)
template <typename T>
__global__ __launch_bounds__(256) void logit_wgrad1_kernel(const float* __restrict__ dz, const T* __restrict__ x, const T* __restrict__ w,
                                                           float* __restrict__ G, float* __restrict__ dots, int R, int K,
                                                           cpcsv_logit_groups g) {
    __shared__ float sh[4];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const float wk = k < K ? elem<T>::ld(w + k) : 0.f;
    for (int gi = 0; gi < g.n; ++gi) {
        float acc = 0.f;
        if (k < K)
            for (int r = g.row[gi]; r < g.row[gi + 1]; ++r) acc += dz[r] * elem<T>::ld(x + (long)r * K + k);
        if (k < K) G[(long)gi * K + k] = acc;
        const float s = blk_sum(acc * wk, sh);
        if (threadIdx.x == 0) dots[gi * gridDim.x + blockIdx.x] = s;
    }
}
// pass 2: dW_master[c*taps + t] += sum_g ( G_g[k] / sigma_g - (<G_g, w> / sigma_g^2) u_g v_g[c*taps + t] ),  k = t*Cin_s + c;
// db += sum_r dz[r] (block 0)
__global__ __launch_bounds__(256) void logit_wgrad2_kernel(const float* __restrict__ G, const float* __restrict__ dots, int nblk1,
                                                           const float* __restrict__ dz, float* __restrict__ dW, float* __restrict__ db,
                                                           int R, int K, int Cin, int Cin_s, int taps, cpcsv_logit_groups g) {
    __shared__ float sh[4];
    __shared__ float coef[4];
    if (threadIdx.x < (unsigned)g.n) {
        float d = 0.f;
        for (int b = 0; b < nblk1; ++b) d += dots[threadIdx.x * nblk1 + b];
        const float sg = g.sigma[threadIdx.x] ? g.sigma[threadIdx.x][0] : 1.f;
        coef[threadIdx.x] = (g.sigma[threadIdx.x] && g.u[threadIdx.x] && g.v[threadIdx.x]) ? d / (sg * sg) * g.u[threadIdx.x][0] : 0.f;
    }
    __syncthreads();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < K) {
        const int t = k / Cin_s, c = k - t * Cin_s;
        if (c < Cin) {
            const int mi = c * taps + t;
            float v = 0.f;
            for (int gi = 0; gi < g.n; ++gi) {
                v += G[(long)gi * K + k] * inv_sigma(g, gi);
                if (coef[gi] != 0.f) v -= coef[gi] * g.v[gi][mi];
            }
            dW[mi] += v;
        }
    }
    if (blockIdx.x == 0 && db) {
        float a = 0.f;
        for (int r = threadIdx.x; r < R; r += 256) a += dz[r];
        const float s = blk_sum(a, sh);
        if (threadIdx.x == 0) db[0] += s;
    }
}

bool groups_ok(const cpcsv_logit_groups* g, int R) {
    if (!g || g->n < 1 || g->n > 4 || g->row[0] != 0 || g->row[g->n] != R) return false;
    for (int i = 0; i < g->n; ++i) if (g->row[i + 1] <= g->row[i]) return false;
    return true;
}

}  // namespace

extern "C" int cpcsv_logit_head_fwd(const void* x, const void* w, const float* bias, float* p, int dtype, int R, int K,
                                    const cpcsv_logit_groups* g, void* stream) {
    if (!x || !w || !p || R <= 0 || K <= 0 || (K & 7) || !groups_ok(g, R)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(logit_fwd_kernel<bf16_t>, dim3(R), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, bias, p, K, *g);
    else hipLaunchKernelGGL(logit_fwd_kernel<float>, dim3(R), dim3(256), 0, s, (const float*)x, (const float*)w, bias, p, K, *g);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_logit_head_bwd(const float* dy, const float* p, const void* w, void* dx, float* dz, int dtype, int R, int K,
                                    const cpcsv_logit_groups* g, void* stream) {
    if (!dy || !p || !w || !dz || R <= 0 || K <= 0 || (K & 7) || !groups_ok(g, R)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(logit_bwd_kernel<bf16_t>, dim3(R), dim3(256), 0, s, dy, p, (const bf16_t*)w, (bf16_t*)dx, dz, K, *g);
    else hipLaunchKernelGGL(logit_bwd_kernel<float>, dim3(R), dim3(256), 0, s, dy, p, (const float*)w, (float*)dx, dz, K, *g);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" long cpcsv_logit_head_scratch(int K, int ngroups) { return (long)ngroups * K + (long)ngroups * ((K + 255) / 256); }

extern "C" int cpcsv_logit_head_wgrad(const float* dz, const void* x, const void* w, float* scratch, float* dW, float* db, int dtype,
                                      int R, int K, int Cin, int Cin_s, int taps, const cpcsv_logit_groups* g, void* stream) {
    if (!dz || !x || !w || !scratch || !dW || R <= 0 || K != taps * Cin_s || Cin > Cin_s || !groups_ok(g, R)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (K + 255) / 256;
    float* G = scratch;
    float* dots = scratch + (long)g->n * K;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(logit_wgrad1_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, dz, (const bf16_t*)x, (const bf16_t*)w, G, dots, R, K, *g);
    else hipLaunchKernelGGL(logit_wgrad1_kernel<float>, dim3(nb), dim3(256), 0, s, dz, (const float*)x, (const float*)w, G, dots, R, K, *g);
    CPCSV_CHECK_LAUNCH();
    hipLaunchKernelGGL(logit_wgrad2_kernel, dim3(nb), dim3(256), 0, s, G, dots, nb, dz, dW, db, R, K, Cin, Cin_s, taps, *g);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
