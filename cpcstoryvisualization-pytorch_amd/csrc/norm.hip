// norm.hip — BatchNorm (train mode), spectral-norm power iteration, weight (un)packing.
// All of these are HBM-streaming kernels: 16-byte vector accesses, fp32 math, no LDS tiles
// except for the cross-row reductions.
#include "common.h"
#include <cstdlib>
#include "../../include/cpcsv_hip.h"

namespace {

// d act / d z from the pre-activation z (recomputed from x: saves re-reading the activation output)
__device__ __forceinline__ float act_grad_from_pre(float z, int act) {
    switch (act) {
        case 1: return z > 0.f ? 1.f : 0.f;
        case 2: return z > 0.f ? 1.f : 0.2f;
        case 3: { const float t = tanhf(z); return 1.f - t * t; }
        case 4: { const float g = 1.f / (1.f + expf(-z)); return g * (1.f - g); }
        default: return 1.f;
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm forward
// ---------------------------------------------------------------------------------------------
// device image of cpcsv_bn_groups (row groups of one layer call: see include/cpcsv_hip.h)
struct BnG {
    int n; long row[5]; long pstride; int tile[5]; int nph, TM; const float* sigma[4];
};

// 16 channels x 16 tile-lanes per block: the per-block partials of the GEMM are summed in double. Row groups (several
// passes of the layer in one launch) are finalised one after the other by the same thread, so the running statistics
// see the passes in call order (r <- (1-m) r + m b does not commute).
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partials, int ldstat, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var, float* mean,
                                   float* invstd, float* scale, float* shift, int C, int Cs, float eps, float momentum,
                                   int update_running, float* bwd_sums, BnG G, long rows_per_count) {
    // block = 16 channels x 64 row lanes (16 wavefronts; a wavefront load covers 4 partial rows x 16 channels = 4 x 64 B).
    // The big maps have thousands of partial rows per channel and only Cs/16 blocks: with 16 row lanes and one dependent load
    // per trip this launch took 50-80 us on the generator's critical path.
    __shared__ double sh[2][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 15, rl = (lane >> 4) + 4 * wave;
    const int c = blockIdx.x * 16 + cl;
    for (int g = 0; g < G.n; ++g) {
        double s = 0.0, q = 0.0;
        if (c < C) {
            const int t0 = G.tile[g], nt = G.tile[g + 1] - G.tile[g];
            const int n = nt * G.nph;
            // eight partial rows (16 loads) in flight per thread and trip: the partials were written by the GEMM that has just
            // retired, so every trip pays a full memory round trip - with two rows per trip the 480-1920 partial rows of the big
            // maps took 8-15 dependent trips (37 us on the generator's critical path)
            for (int k = rl; k < n; k += 512) {
                float a[8], b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = k + 64 * j;
                    const bool ok = kk < n;
                    const long t = ok ? (long)(kk / nt) * G.TM + t0 + (kk % nt) : 0;
                    a[j] = ok ? partials[(t * 2 + 0) * ldstat + c] : 0.f;
                    b[j] = ok ? partials[(t * 2 + 1) * ldstat + c] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { s += (double)a[j]; q += (double)b[j]; }
            }
        }
        s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
        s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
        __syncthreads();
        if (lane < 16) { sh[0][wave][cl] = s; sh[1][wave][cl] = q; }
        __syncthreads();
        if (threadIdx.x >= 16 || c >= Cs) continue;
        const long off = (long)g * G.pstride;
        if (bwd_sums)                                              // accumulators of the backward pass, zeroed for free
            for (int k = 0; k < 2 * CPCSV_BN_SUM_COPIES; ++k) bwd_sums[off + k * Cs + c] = 0.f;
        if (c >= C) { scale[off + c] = 0.f; shift[off + c] = 0.f; mean[off + c] = 0.f; invstd[off + c] = 0.f; continue; }
        s = 0.0; q = 0.0;
        for (int k = 0; k < 16; ++k) { s += sh[0][k][cl]; q += sh[1][k][cl]; }
        const double count = (double)((G.row[g + 1] - G.row[g]) * rows_per_count);
        const double inv_count = 1.0 / count, unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
        const double mu = s * inv_count;
        double var = q * inv_count - mu * mu;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        mean[off + c] = (float)mu;
        invstd[off + c] = is;
        const float ga = gamma[c];
        float sc_, sh_;
        bn_affine(ga, beta[c], (float)mu, is, sc_, sh_);
        scale[off + c] = sc_;
        shift[off + c] = sh_;
        if (update_running) {
            running_mean[c] = bn_running(running_mean[c], (float)mu, momentum);
            running_var[c] = bn_running(running_var[c], (float)(var * unbias), momentum);
        }
    }
}

// Elementwise BN kernels: a thread owns ONE 16-byte channel chunk (its per-channel constants live in
// registers) and walks rows; block = cw chunk columns x (256/cw) row lanes; no per-element division.
template <typename T, bool SMOOTH>
__global__ void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale,
                                const float* __restrict__ shift, int cpr, int cw, int rows_per_block, int C,
                                int act, BnG G) {
    constexpr int EPC = elem<T>::per16;
    const int rl = blockDim.x / cw;
    const int cx = threadIdx.x % cw, ry = threadIdx.x / cw;
    const int chunk = blockIdx.x * cw + cx;
    if (ry >= rl || chunk >= cpr) return;
    const int c0 = chunk * EPC;
    const long goff = (long)blockIdx.z * G.pstride, rows = G.row[blockIdx.z + 1];
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[goff + c0 + e]; sh[e] = shift[goff + c0 + e]; }
    const long r0 = G.row[blockIdx.z] + (long)blockIdx.y * rows_per_block;
    if (r0 >= rows) return;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const ActPl apl = act_pl(act);
    auto one = [&](const u32x4& raw, long i) {
        const T* xs = reinterpret_cast<const T*>(&raw);
        u32x4 outv;
        T* ys = reinterpret_cast<T*>(&outv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float v = bn_pre(elem<T>::ld(xs + e), sc[e], sh[e]);
            elem<T>::st(ys + e, c0 + e < C ? act_apply_t<SMOOTH>(v, act, apl) : 0.f);   // pad channels stay zero
        }
        reinterpret_cast<u32x4*>(y)[i] = outv;
    };
    // four independent 16-byte loads in flight per thread before the first use: a single load per loop iteration left
    // this kernel latency-bound at ~1.2 TB/s
    long r = r0 + ry;
    for (; r + 3L * rl < r1; r += 4L * rl) {
        const long i0 = r * cpr + chunk, st = (long)rl * cpr;
        const u32x4 a = reinterpret_cast<const u32x4*>(x)[i0], b = reinterpret_cast<const u32x4*>(x)[i0 + st];
        const u32x4 c = reinterpret_cast<const u32x4*>(x)[i0 + 2 * st], d = reinterpret_cast<const u32x4*>(x)[i0 + 3 * st];
        one(a, i0); one(b, i0 + st); one(c, i0 + 2 * st); one(d, i0 + 3 * st);
    }
    for (; r < r1; r += rl) {
        const long i = r * cpr + chunk;
        one(reinterpret_cast<const u32x4*>(x)[i], i);
    }
}

// bn_finalize + bn_apply in ONE launch for calls with only a handful of statistics partial rows (dense layers over <= 16
// statistics tiles: the generator's fc / fc_seg, the critics' 4x4 maps): every block sums the GEMM's per-block partial rows of its own
// 8-channel chunks itself (a few loads and double flops per thread, fixed order), so there is no finalize launch between the GEMM
// and this pass. Block (y = 0, z = g) also stores mean / invstd / scale / shift of group g for the backward pass and zeroes that
// group's backward accumulators; block (y = 0, z = 0) updates the running statistics, group after group (call order). Layout and
// group bookkeeping of the partial rows as in bn_finalize_kernel.
// sums of the EPC consecutive channels a thread owns (c0 a multiple of EPC; ldstat a multiple of 8): 16-byte loads, the loads of
// up to four partial rows issued before their sums. Channel by channel with scalar loads the 64 loads of a thread ran one after the
// other: 38 us for the generator's 120 x 16384 `fc` output.
template <int EPC>
__device__ __forceinline__ void bn_group_sums_vec(const float* partials, int ldstat, const BnG& G, int g, int c0, int Cs,
                                                  double (&s1)[EPC], double (&s2)[EPC]) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s1[e] = 0.0; s2[e] = 0.0; }
    const int t0 = G.tile[g], nt = G.tile[g + 1] - G.tile[g];
    const int n = nt * G.nph;
    constexpr int V = EPC / 4;
    for (int k0 = 0; k0 < n; k0 += 4) {
        f32x4 a[4][V], b[4][V];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k0 + j;
            const bool ok = kk < n;
            const long t = ok ? (long)(kk / nt) * G.TM + t0 + (kk % nt) : 0;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                a[j][v] = ok ? *reinterpret_cast<const f32x4*>(partials + (t * 2 + 0) * ldstat + c0 + 4 * v) : f32x4{0.f, 0.f, 0.f, 0.f};
                b[j][v] = ok ? *reinterpret_cast<const f32x4*>(partials + (t * 2 + 1) * ldstat + c0 + 4 * v) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + j < n) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) { s1[e] += (double)a[j][e >> 2][e & 3]; s2[e] += (double)b[j][e >> 2][e & 3]; }
            }
    }
}

template <typename T, bool SMOOTH>
__global__ void bn_apply_fused_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ partials, int ldstat,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
                                      float* running_var, float* __restrict__ stat_out, float* __restrict__ bwd_sums, int cpr, int cw,
                                      int rows_per_block, int C, int Cs, int act, float eps, float momentum, BnG G) {
    constexpr int EPC = elem<T>::per16;
    const int rl = blockDim.x / cw;
    const int cx = threadIdx.x % cw, ry = threadIdx.x / cw;
    const int chunk = blockIdx.x * cw + cx;
    const bool live = ry < rl && chunk < cpr;
    const int c0 = chunk * EPC;
    const int g = blockIdx.z;
    const long goff = (long)g * G.pstride;
    float sc[EPC], sh[EPC];
    if (live) {
        const double cnt = (double)(G.row[g + 1] - G.row[g]);
        double s1v[EPC], s2v[EPC];
        float muv[EPC], isv[EPC];
        bn_group_sums_vec<EPC>(partials, ldstat, G, g, c0, Cs, s1v, s2v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int c = c0 + e;
            float scale = 0.f, shift = 0.f, mu_f = 0.f, is = 0.f;
            if (c < C) {
                const double s1 = s1v[e], s2 = s2v[e];
                const double mu = s1 / cnt;
                double var = s2 / cnt - mu * mu;
                if (var < 0.0) var = 0.0;
                is = (float)(1.0 / sqrt(var + (double)eps));
                mu_f = (float)mu;
                const float ga = gamma[c];
                bn_affine(ga, beta[c], mu_f, is, scale, shift);
            }
            sc[e] = scale; sh[e] = shift;
            muv[e] = mu_f; isv[e] = is;
        }
        if (blockIdx.y == 0 && ry == 0) {
            // this call's statistics for the backward pass and its zeroed accumulators: 16-byte stores of the thread's EPC consecutive
            // channels (element by element the 16 accumulator rows were 128 four-byte stores per thread at a 32-byte lane stride -
            // most of this kernel's 33 us on the generator's 16384-feature fc layer)
            constexpr int V = EPC / 4;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float* so = stat_out + goff + c0 + 4 * v;
                *reinterpret_cast<f32x4*>(so + 0L * Cs) = f32x4{muv[4 * v], muv[4 * v + 1], muv[4 * v + 2], muv[4 * v + 3]};
                *reinterpret_cast<f32x4*>(so + 1L * Cs) = f32x4{isv[4 * v], isv[4 * v + 1], isv[4 * v + 2], isv[4 * v + 3]};
                *reinterpret_cast<f32x4*>(so + 2L * Cs) = f32x4{sc[4 * v], sc[4 * v + 1], sc[4 * v + 2], sc[4 * v + 3]};
                *reinterpret_cast<f32x4*>(so + 3L * Cs) = f32x4{sh[4 * v], sh[4 * v + 1], sh[4 * v + 2], sh[4 * v + 3]};
                if (bwd_sums)
                    for (int k = 0; k < 2 * CPCSV_BN_SUM_COPIES; ++k)
                        *reinterpret_cast<f32x4*>(bwd_sums + goff + (long)k * Cs + c0 + 4 * v) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (blockIdx.y == 0 && blockIdx.z == 0 && ry == 0 && running_mean) {
            // running statistics: every group's batch, in call order (r <- (1-m) r + m b does not commute)
            float rm[EPC], rv[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) { rm[e] = c0 + e < C ? running_mean[c0 + e] : 0.f; rv[e] = c0 + e < C ? running_var[c0 + e] : 0.f; }
            for (int k = 0; k < G.n; ++k) {
                const double cn = (double)(G.row[k + 1] - G.row[k]);
                const double unbias = cn > 1.0 ? cn / (cn - 1.0) : 1.0;
                double r1[EPC], r2[EPC];
                bn_group_sums_vec<EPC>(partials, ldstat, G, k, c0, Cs, r1, r2);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const double mu = r1[e] / cn;
                    double var = r2[e] / cn - mu * mu;
                    if (var < 0.0) var = 0.0;
                    rm[e] = bn_running(rm[e], (float)mu, momentum);
                    rv[e] = bn_running(rv[e], (float)(var * unbias), momentum);
                }
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                if (c0 + e < C) { running_mean[c0 + e] = rm[e]; running_var[c0 + e] = rv[e]; }
        }
    }
    if (!live) return;
    const long rows = G.row[g + 1];
    const long r0 = G.row[g] + (long)blockIdx.y * rows_per_block;
    if (r0 >= rows) return;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const ActPl apl = act_pl(act);
    auto one = [&](const u32x4& raw, long i) {
        const T* xs = reinterpret_cast<const T*>(&raw);
        u32x4 outv;
        T* ys = reinterpret_cast<T*>(&outv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float v = bn_pre(elem<T>::ld(xs + e), sc[e], sh[e]);
            elem<T>::st(ys + e, c0 + e < C ? act_apply_t<SMOOTH>(v, act, apl) : 0.f);   // pad channels stay zero
        }
        reinterpret_cast<u32x4*>(y)[i] = outv;
    };
    long r = r0 + ry;
    for (; r + 3L * rl < r1; r += 4L * rl) {
        const long i0 = r * cpr + chunk, st = (long)rl * cpr;
        const u32x4 a = reinterpret_cast<const u32x4*>(x)[i0], b = reinterpret_cast<const u32x4*>(x)[i0 + st];
        const u32x4 c = reinterpret_cast<const u32x4*>(x)[i0 + 2 * st], d = reinterpret_cast<const u32x4*>(x)[i0 + 3 * st];
        one(a, i0); one(b, i0 + st); one(c, i0 + 2 * st); one(d, i0 + 3 * st);
    }
    for (; r < r1; r += rl) {
        const long i = r * cpr + chunk;
        one(reinterpret_cast<const u32x4*>(x)[i], i);
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward: pass 1 (column reductions), pass 2 (apply)
// ---------------------------------------------------------------------------------------------
// SMOOTH = false: the activation is piecewise linear (none / ReLU / LeakyReLU: every BatchNorm of the model), its derivative is a
// compare + select on a per-launch slope. (With the activation code as a run-time switch inside the element loop the kernel carried
// a scalar branch tree and the tanh / exp paths per element: 2.5 TB/s alone on the 64x64 maps, half of what the forward apply reaches.)
template <typename T, bool SMOOTH>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* sums, int C, int Cs, int cpr, int cw, int rows_per_block, int act, BnG G) {
    constexpr int EPC = elem<T>::per16;
    extern __shared__ float red[];  // [rl][cw][2*EPC]
    const int rl = blockDim.x / cw;
    const int cx = threadIdx.x % cw, ry = threadIdx.x / cw;
    const int chunk = blockIdx.x * cw + cx;
    const long goff = (long)blockIdx.z * G.pstride, rows = G.row[blockIdx.z + 1];
    mean += goff; invstd += goff; sums += goff;
    float s0[EPC], s1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) s0[e] = s1[e] = 0.f;
    const long r0 = G.row[blockIdx.z] + (long)blockIdx.y * rows_per_block;
    const bool active = ry < rl && chunk < cpr && r0 < rows;
    const float slope = act == CPCSV_ACT_RELU ? 0.f : (act == CPCSV_ACT_LRELU ? 0.2f : 1.f);
    if (active) {
        const int c0 = chunk * EPC;
        float mu[EPC], is[EPC], sc[EPC], sh[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const bool ok = c0 + e < C;
            mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e];
            bn_affine(ok ? gamma[c0 + e] : 0.f, ok ? beta[c0 + e] : 0.f, mu[e], is[e], sc[e], sh[e]);
        }
        const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
        auto one = [&](const u32x4& a, const u32x4& b) {
            const T* pa = reinterpret_cast<const T*>(&a);
            const T* pb = reinterpret_cast<const T*>(&b);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float xv = elem<T>::ld(pb + e);
                const float xh = (xv - mu[e]) * is[e];
                const float pre = bn_pre(xv, sc[e], sh[e]);            // the forward's pre-activation, bit for bit
                const float dz = elem<T>::ld(pa + e) * (SMOOTH ? act_grad_from_pre(pre, act) : (pre > 0.f ? 1.f : slope));
                s0[e] += dz;
                s1[e] += dz * xh;
            }
        };
        long r = r0 + ry;
        for (; r + 3L * rl < r1; r += 4L * rl) {       // eight independent 16-byte loads in flight per thread
            const long i0 = r * cpr + chunk, st = (long)rl * cpr;
            const u32x4 a0 = reinterpret_cast<const u32x4*>(dy)[i0], b0 = reinterpret_cast<const u32x4*>(x)[i0];
            const u32x4 a1 = reinterpret_cast<const u32x4*>(dy)[i0 + st], b1 = reinterpret_cast<const u32x4*>(x)[i0 + st];
            const u32x4 a2 = reinterpret_cast<const u32x4*>(dy)[i0 + 2 * st], b2 = reinterpret_cast<const u32x4*>(x)[i0 + 2 * st];
            const u32x4 a3 = reinterpret_cast<const u32x4*>(dy)[i0 + 3 * st], b3 = reinterpret_cast<const u32x4*>(x)[i0 + 3 * st];
            one(a0, b0); one(a1, b1); one(a2, b2); one(a3, b3);
        }
        for (; r < r1; r += rl) {
            const long i = r * cpr + chunk;
            one(reinterpret_cast<const u32x4*>(dy)[i], reinterpret_cast<const u32x4*>(x)[i]);
        }
    }
    float* mine = red + ((long)ry * cw + cx) * 2 * EPC;
    if (ry < rl) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { mine[e] = s0[e]; mine[EPC + e] = s1[e]; }
    }
    __syncthreads();
    if (ry == 0 && chunk < cpr && r0 < rows) {
        for (int k = 1; k < rl; ++k) {
            const float* o = red + ((long)k * cw + cx) * 2 * EPC;
#pragma unroll
            for (int e = 0; e < EPC; ++e) { s0[e] += o[e]; s1[e] += o[EPC + e]; }
        }
        const int c0 = chunk * EPC;
        // the row slabs spread their column sums over CPCSV_BN_SUM_COPIES copies of the accumulator: contended
        // device-scope float atomics on ONE address retire at ~0.15 us each, which used to dominate this kernel
        float* mysums = sums + (long)(blockIdx.y % CPCSV_BN_SUM_COPIES) * 2 * Cs;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            atomicAdd(mysums + c0 + e, s0[e]);
            atomicAdd(mysums + Cs + c0 + e, s1[e]);
        }
    }
}

template <typename T, bool SMOOTH>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                    T* __restrict__ dx, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta,
                                    const float* __restrict__ sums, float* dgamma, float* dbeta,
                                    int cpr, int cw, int rows_per_block, int C, int Cs, int act,
                                    int accumulate, float* gw_out, float eps, BnG G) {
    constexpr int EPC = elem<T>::per16;
    constexpr int NS = CPCSV_BN_SUM_COPIES;
    const long goff = (long)blockIdx.z * G.pstride, rows = G.row[blockIdx.z + 1];
    mean += goff; invstd += goff; sums += goff;
    const float* sigma = G.sigma[blockIdx.z];
    if (gw_out) gw_out += blockIdx.z;
    const float inv_rows = 1.f / (float)(rows - G.row[blockIdx.z]);
    const float oscale = sigma ? sigma[1] : 1.f;          // 1/sigma of the spectral-normed conv in front, folded into dz
    const float slope = act == CPCSV_ACT_RELU ? 0.f : (act == CPCSV_ACT_LRELU ? 0.2f : 1.f);
    auto total = [&](int which, int c) {                      // column sum over the accumulator copies of pass 1
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) t += sums[(long)k * 2 * Cs + which * Cs + c];
        return t;
    };
    if (blockIdx.x == 0 && blockIdx.y == 0 && gw_out && sigma) {
        // <dL/dW_eff, W_orig> of the spectral-normed conv in front of this BatchNorm, in closed form: BN removes the
        // mean and (up to eps) the scale of its input, so sum_m dx*x = gamma * (sum_m dz*xhat) * eps * invstd^2 per channel
        __shared__ float red[16];
        float a = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) a += gamma[c] * total(1, c) * eps * invstd[c] * invstd[c];
        for (int off = 32; off; off >>= 1) a += __shfl_xor(a, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
            gw_out[0] = sigma[0] * t;
        }
    }
    // this block's columns: totals of the copies, once per block through LDS
    __shared__ float tot[2][256 * 8];
    for (int q = threadIdx.x; q < cw * EPC; q += blockDim.x) {
        const int c = blockIdx.x * cw * EPC + q;
        tot[0][q] = c < C ? total(0, c) : 0.f;
        tot[1][q] = c < C ? total(1, c) : 0.f;
    }
    __syncthreads();
    if (blockIdx.y == 0 && blockIdx.z == 0 && dgamma) {       // the first row slab of every column block owns its dgamma/dbeta
        for (int q = threadIdx.x; q < cw * EPC; q += blockDim.x) {
            const int c = blockIdx.x * cw * EPC + q;
            if (c < C) {
                // all row groups, summed here in group order (one writer per channel and launch: a fixed order)
                float t1 = tot[1][q], t0 = tot[0][q];
                for (int g = 1; g < G.n; ++g) {
                    const float* sg = sums + (long)g * G.pstride;
#pragma unroll
                    for (int k = 0; k < NS; ++k) { t0 += sg[(long)k * 2 * Cs + c]; t1 += sg[(long)k * 2 * Cs + Cs + c]; }
                }
                // atomics: the two halves of a generator pass may run their backward on two streams at once
                if (accumulate) { atomicAdd(dgamma + c, t1); atomicAdd(dbeta + c, t0); }
                else { dgamma[c] = t1; dbeta[c] = t0; }
            }
        }
    }
    const int rl = blockDim.x / cw;
    const int cx = threadIdx.x % cw, ry = threadIdx.x / cw;
    const int chunk = blockIdx.x * cw + cx;
    if (ry >= rl || chunk >= cpr) return;
    const int c0 = chunk * EPC;
    float mu[EPC], is[EPC], ga[EPC], sc[EPC], sh[EPC], k0[EPC], k1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const bool ok = c0 + e < C;
        mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e];
        ga[e] = ok ? gamma[c0 + e] : 0.f;
        bn_affine(ga[e], ok ? beta[c0 + e] : 0.f, mu[e], is[e], sc[e], sh[e]);
        k0[e] = tot[0][cx * EPC + e] * inv_rows; k1[e] = tot[1][cx * EPC + e] * inv_rows;
    }
    const long r0 = G.row[blockIdx.z] + (long)blockIdx.y * rows_per_block;
    if (r0 >= rows) return;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    auto one = [&](const u32x4& a, const u32x4& b, long i) {
        const T* pa = reinterpret_cast<const T*>(&a);
        const T* pb = reinterpret_cast<const T*>(&b);
        u32x4 outv;
        T* po = reinterpret_cast<T*>(&outv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float xv = elem<T>::ld(pb + e);
            const float xh = (xv - mu[e]) * is[e];
            const float pre = bn_pre(xv, sc[e], sh[e]);                // the forward's pre-activation, bit for bit
            const float dz = elem<T>::ld(pa + e) * (SMOOTH ? act_grad_from_pre(pre, act) : (pre > 0.f ? 1.f : slope));
            elem<T>::st(po + e, oscale * ga[e] * is[e] * (dz - k0[e] - xh * k1[e]));     // ga = 0 on pad channels
        }
        reinterpret_cast<u32x4*>(dx)[i] = outv;
    };
    long r = r0 + ry;
    for (; r + 3L * rl < r1; r += 4L * rl) {           // eight independent 16-byte loads in flight per thread
        const long i0 = r * cpr + chunk, st = (long)rl * cpr;
        const u32x4 a0 = reinterpret_cast<const u32x4*>(dy)[i0], b0 = reinterpret_cast<const u32x4*>(x)[i0];
        const u32x4 a1 = reinterpret_cast<const u32x4*>(dy)[i0 + st], b1 = reinterpret_cast<const u32x4*>(x)[i0 + st];
        const u32x4 a2 = reinterpret_cast<const u32x4*>(dy)[i0 + 2 * st], b2 = reinterpret_cast<const u32x4*>(x)[i0 + 2 * st];
        const u32x4 a3 = reinterpret_cast<const u32x4*>(dy)[i0 + 3 * st], b3 = reinterpret_cast<const u32x4*>(x)[i0 + 3 * st];
        one(a0, b0, i0); one(a1, b1, i0 + st); one(a2, b2, i0 + 2 * st); one(a3, b3, i0 + 3 * st);
    }
    for (; r < r1; r += rl) {
        const long i = r * cpr + chunk;
        one(reinterpret_cast<const u32x4*>(dy)[i], reinterpret_cast<const u32x4*>(x)[i], i);
    }
}

struct TapMap { int8_t m[CPCSV_MAX_TAPS]; };
struct MaskTab { uint16_t m[CPCSV_MAX_TAPS]; };

// ---- weight (un)packing: LDS-tiled so that BOTH the fp32 master side ([o][i][taps], taps innermost) and the
// packed side (channel innermost) are read/written in contiguous runs. A slice's value is either one master tap
// (map.m[sl], -1 = zero slice) or, when sum != 0, the sum of the taps in masks.m[sl] (sub-pixel upsample+conv).
constexpr int PK_I = 64;       // input channels per block (forward pack / unpack)

__device__ __forceinline__ float slice_value(const float* t, int taps, int sl, const TapMap& map, const MaskTab& mk, int sum) {
    if (!sum) { const int tt = map.m[sl]; return tt >= 0 ? t[tt] : 0.f; }
    float v = 0.f;
    const unsigned m = mk.m[sl];
    for (int k = 0; k < taps; ++k) if (m & (1u << k)) v += t[k];
    return v;
}

// forward pack: dst[o][sl*Cin_s + i]; one thread per (o, stored channel i): reads its `taps` contiguous master
// floats, writes S values that are contiguous ACROSS the wave (lanes = consecutive i)
template <typename T>
__global__ void pack_fwd_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int taps, int S,
                                TapMap map, MaskTab mk, int sum, int Cin_s) {
    const unsigned total = (unsigned)Cout * Cin_s;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx % (unsigned)Cin_s;
        const unsigned o = idx / (unsigned)Cin_s;
        float t[CPCSV_MAX_TAPS];
        if (i < Cin) {
            const float* src = w + ((long)o * Cin + i) * taps;
#pragma unroll
            for (int k = 0; k < CPCSV_MAX_TAPS; ++k) t[k] = k < taps ? src[k] : 0.f;
        }
        T* drow = dst + (long)o * S * Cin_s + i;
        for (int sl = 0; sl < S; ++sl)
            elem<T>::st(drow + (long)sl * Cin_s, i < Cin ? slice_value(t, taps, sl, map, mk, sum) : 0.f);
    }
}

// backward packs: mode 1 dst[i][sl*Cout_s + o], mode 2 dst[sl*Cin_s + i][o]; block = (64 output channels, 16 input channels)
constexpr int PB_O = 64, PB_I = 8;   // LDS tile 64 x (8*16+1) floats = 33 KB
template <typename T>
__global__ void pack_bwd_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int taps, int S,
                                TapMap map, MaskTab mk, int sum, int Cin_s, int Cout_s, int mode) {
    __shared__ float sh[PB_O][PB_I * CPCSV_MAX_TAPS + 1];
    const int o0 = blockIdx.y * PB_O, i0 = blockIdx.x * PB_I;
    const int ni = Cin - i0 < PB_I ? Cin - i0 : PB_I;
    const int run = ni * taps;                                   // contiguous floats per output channel
    // (n / d for n, d < 2^16 as umulhi(n, ceil(2^32 / d)), exact; d = 1 passes through: the run-time divisions per element of
    // these two loops made the pack instruction-bound)
    const unsigned m_row = 0xFFFFFFFFu / (unsigned)(PB_I * taps) + 1u, m_s = 0xFFFFFFFFu / (unsigned)S + 1u;
    for (int k = threadIdx.x; k < PB_O * PB_I * taps; k += blockDim.x) {
        const int oo = (int)__umulhi((unsigned)k, m_row), r = k - oo * (PB_I * taps);
        if (r < run && o0 + oo < Cout) sh[oo][r] = w[((long)(o0 + oo) * Cin + i0) * taps + r];
    }
    __syncthreads();
    const int iw = (mode == 1 ? Cin : Cin_s) - i0;               // rows to write: mode 1 has one row per REAL channel
    for (int k = threadIdx.x; k < PB_I * S * PB_O; k += blockDim.x) {
        const int oo = k % PB_O, kq = k / PB_O, ii = S == 1 ? kq : (int)__umulhi((unsigned)kq, m_s), sl = kq - ii * S;
        if (ii >= iw || ii >= PB_I || o0 + oo >= Cout_s) continue;
        const float v = (ii < ni && o0 + oo < Cout) ? slice_value(&sh[oo][ii * taps], taps, sl, map, mk, sum) : 0.f;
        const long off = mode == 1 ? ((long)(i0 + ii) * S + sl) * Cout_s + o0 + oo
                                   : ((long)sl * Cin_s + i0 + ii) * Cout_s + o0 + oo;
        elem<T>::st(dst + off, v);
    }
}

// gradient unpack: dw[o][i][t] (+)= (sum over slices feeding tap t of G[o][sl*Cin_s + i]) / sigma - coef*u[o]*v[i*taps+t].
// A block owns 256 consecutive (o, i) pairs = one contiguous run of 256*taps master elements: each thread reads its S
// slice values (coalesced across the wave along i), hands them back zeroed when rezero != 0, and parks its `taps`
// results in LDS; the block then streams the run out with unit-stride loads/stores (the per-thread scatter of `taps`
// floats at a 4*taps-byte lane stride this replaces ran at ~1.2 TB/s).
constexpr int UNP_T = 256;
__global__ __launch_bounds__(UNP_T) void unpack_tiled_kernel(float* __restrict__ G, float* __restrict__ dw,
                                                              const float* __restrict__ sigma, const float* __restrict__ u,
                                                              const float* __restrict__ v, const float* __restrict__ gw_dot,
                                                              int Cout, int Cin, int taps, int S, TapMap inv, MaskTab mk,
                                                              int sum, int Cin_s, int accumulate, int rezero) {
    __shared__ float sm[UNP_T * (CPCSV_MAX_TAPS + 1)];
    float is = 1.f, coefs = 0.f;
    if (sigma) { const float sg = sigma[0]; is = 1.f / sg; coefs = gw_dot ? gw_dot[0] / (sg * sg) : 0.f; }
    const bool rank1 = sigma && u && v;
    const unsigned total = (unsigned)Cout * Cin;
    const int ldt = taps + 1;                                  // odd/padded row stride: conflict-free LDS writes
    for (unsigned base = blockIdx.x * UNP_T; base < total; base += gridDim.x * UNP_T) {
        const unsigned idx = base + threadIdx.x;
        if (idx < total) {
            const int i = idx % (unsigned)Cin;
            const unsigned o = idx / (unsigned)Cin;
            float* gp = G + (long)o * S * Cin_s + i;
            float g[CPCSV_MAX_TAPS];
#pragma unroll
            for (int sl = 0; sl < CPCSV_MAX_TAPS; ++sl) {
                g[sl] = 0.f;
                if (sl < S) { g[sl] = gp[(long)sl * Cin_s]; if (rezero) gp[(long)sl * Cin_s] = 0.f; }
            }
            for (int t = 0; t < taps; ++t) {
                float val = 0.f;
                if (!sum) {
                    const int sl = inv.m[t];
#pragma unroll
                    for (int k = 0; k < CPCSV_MAX_TAPS; ++k) if (k == sl) val = g[k];
                } else {
#pragma unroll
                    for (int k = 0; k < CPCSV_MAX_TAPS; ++k) if (mk.m[k] & (1u << t)) val += g[k];
                }
                sm[threadIdx.x * ldt + t] = val * is;
            }
        }
        __syncthreads();
        const unsigned n_here = (total - base < (unsigned)UNP_T ? total - base : (unsigned)UNP_T) * taps;
        float* out = dw + (long)base * taps;
        for (unsigned e = threadIdx.x; e < n_here; e += UNP_T) {
            const unsigned p = e / (unsigned)taps, t = e - p * taps;
            float val = sm[p * ldt + t];
            if (rank1) {
                const unsigned o = (base + p) / (unsigned)Cin, i = (base + p) - o * Cin;
                val -= coefs * u[o] * v[(long)i * taps + t];
            }
            if (accumulate) out[e] += val; else out[e] = val;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// fused per-layer optimiser step (include/cpcsv_hip.h: cpcsv_layer_update): accumulator -> gradient -> Adam -> operand
// copies, every master element touched once. Block = tile of UT_O output x UT_I input channels, all taps, staged in LDS
// (row strides odd: the o-fastest and the i-fastest passes are both conflict-free):
//   1. thread (o, i) reads its S accumulator slices (coalesced along i), folds them into the `taps` gradients
//   2. threads walk the tile's master runs (for one o: UT_I*taps contiguous floats): Adam on p, m, v; new p back to LDS
//   3. forward copy  [o][sl*Cin_s + i]   (i fastest)      4. backward copy [i][sl*Cout_s + o] / dense [sl*Cin_s + i][o] (o fastest)
// ---------------------------------------------------------------------------------------------
// optimiser-state traffic (5 GB per step: accumulator, master, m, v) is touched once per step: it is marked non-temporal so
// that it does not evict the operand panels the concurrently running GEMMs re-read from L2 (19.50 -> 19.44 ms/step; the
// GEMMs do depend on those hits: marking THEIR operand loads non-temporal costs +0.9 ... +1.9 ms). -DCPCSV_UPD_TEMPORAL: A/B.
#ifndef CPCSV_UPD_TEMPORAL
#define UPD_LD(p) __builtin_nontemporal_load(p)
#define UPD_ST(p, v) __builtin_nontemporal_store(v, p)
#else
#define UPD_LD(p) (*(p))
#define UPD_ST(p, v) (*(p) = (v))
#endif
struct UpdTerms { int n; const float* gw[4]; const float* sigma[4]; const float* u[4]; const float* v[4]; };

template <typename T, int UT_O, int UT_I>
__global__ __launch_bounds__(256) void layer_update_kernel(const float* __restrict__ G, float* __restrict__ p, float* __restrict__ m,
                                                           float* __restrict__ v, void* fwd_, void* bwd_, void* lin_,
                                                           const float* __restrict__ hyper, float beta1, float beta2, float eps, int Cout,
                                                           int Cin, int taps, int S, int Cin_s, int Cout_s, int sum, TapMap fmap, TapMap inv,
                                                           MaskTab mk, UpdTerms terms, int LT, int LO, int probe, float gscale, float step_add,
                                                           int g_bf16) {
    T* __restrict__ fwd = reinterpret_cast<T*>(fwd_);
    T* __restrict__ bwd = reinterpret_cast<T*>(bwd_);
    T* __restrict__ lin = reinterpret_cast<T*>(lin_);
    extern __shared__ float sm[];                      // [UT_O][LO], row o = [UT_I][LT]
    __shared__ float hs[2 + 4];
    __shared__ int8_t tl[CPCSV_MAX_TAPS][4];
    const int tid = threadIdx.x;
    int bxi = blockIdx.x, byi = blockIdx.y;
    if (probe & 0x10000) {
        // XCD-aware block map (1-D grid). The data-gradient copy leaves a block in 16-byte pieces - UT_O = 8 consecutive output
        // channels per (input channel, slice) - and the 8 blocks that own the other pieces of those 128-byte lines are the next 8
        // output tiles of the same input tile. In the (x, y) grid they sit gx blocks apart and go round-robin to different XCDs,
        // whose L2s each hold an eighth of every line and write it back on its own (without the copy the kernel is 8-26 % faster
        // for 6 % fewer bytes, profiles/r05_update_probe.txt). Here those 8 blocks are CONSECUTIVE blocks of ONE XCD: the pieces meet
        // in one L2 and leave as whole lines.
        const int gx = (Cin + UT_I - 1) / UT_I, gy = (Cout + UT_O - 1) / UT_O, gy8 = (gy + 7) >> 3;
        const unsigned L = blockIdx.x, xcd = L & 7, idx = L >> 3;
        const int grp = (int)(idx >> 3) * 8 + (int)xcd;
        if (grp >= gx * gy8) return;
        bxi = grp % gx;
        byi = (grp / gx) * 8 + (int)(idx & 7);
        if (byi >= gy) return;
    }
    const int o0 = byi * UT_O, i0 = bxi * UT_I;
    const int no = Cout - o0 < UT_O ? Cout - o0 : UT_O, ni = Cin - i0 < UT_I ? Cin - i0 : UT_I;
    if (tid == 0) {
        const float t = hyper[0] + step_add, lr = hyper[1];
        hs[0] = lr / (1.f - powf(beta1, t));           // step size
        hs[1] = 1.f / sqrtf(1.f - powf(beta2, t));     // 1/sqrt(bias_correction2)
    }
    if (tid < terms.n) { const float sg = terms.sigma[tid][0]; hs[2 + tid] = terms.gw[tid][0] / (sg * sg); }
    if (tid >= 64 && tid < 64 + S) {                     // taps of slice sl
        const int sl = tid - 64;
        int cnt = 0;
        if (sum) { for (int t = 0; t < taps; ++t) if ((mk.m[sl] & (1u << t)) && cnt < 4) tl[sl][cnt++] = (int8_t)t; }
        else if (fmap.m[sl] >= 0) tl[sl][cnt++] = fmap.m[sl];
        for (; cnt < 4; ++cnt) tl[sl][cnt] = -1;
    }
    // n / d for n, d < 2^16 as umulhi(n, ceil(2^32 / d)) (exact); d = 1 has no 32-bit reciprocal. The two run-time divisions per
    // element of the Adam pass and the two per 16-byte store of the copy passes were as many instructions as the Adam arithmetic.
    auto recip = [](int d) { return 0xFFFFFFFFu / (unsigned)d + 1u; };
    auto fdiv = [](int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); };
    const int run = ni * taps;
    const unsigned m_run = recip(run > 0 ? run : 1), m_taps = recip(taps), m_S = recip(S);
    const int n2 = (probe & 2) ? 0 : no * run;
    // (Tried in round 5: master / m / v of all of a thread's elements requested in FRONT of the accumulator loads, one memory round
    // trip for the whole input: 103 instead of 71 registers, 4 instead of 7 blocks per CU - 13.61 -> 13.72 ms per step. Dropped.)
    // ---- 1. accumulator slices -> tap gradients
    for (int q = tid; q < ((probe & 1) ? 0 : UT_O * UT_I); q += 256) {
        const int o = q / UT_I, i = q - o * UT_I;
        if (o >= no || i >= ni) continue;
        const long gi0 = (long)(o0 + o) * S * Cin_s + i0 + i;
        const float* gp = G + gi0;
        const bf16_t* gpb = reinterpret_cast<const bf16_t*>(G) + gi0;      // g_bf16: the accumulator as the bf16 wire buffer of the exchange holds it
        float g[CPCSV_MAX_TAPS];
#pragma unroll
        for (int sl = 0; sl < CPCSV_MAX_TAPS; ++sl)
            g[sl] = sl < S ? (g_bf16 ? bf16_to_f32(gpb[(long)sl * Cin_s]) : UPD_LD(gp + (long)sl * Cin_s)) : 0.f;
        float* row = sm + o * LO + i * LT;
        for (int t = 0; t < taps; ++t) {
            float val = 0.f;
            if (!sum) {
                const int sl = inv.m[t];
#pragma unroll
                for (int k = 0; k < CPCSV_MAX_TAPS; ++k) if (k == sl) val = g[k];
            } else {
#pragma unroll
                for (int k = 0; k < CPCSV_MAX_TAPS; ++k) if (mk.m[k] & (1u << t)) val += g[k];
            }
            row[t] = val * gscale;               // 1/world of a SUM-reduced accumulator (data-parallel mean), else 1
        }
    }
    __syncthreads();
    // ---- 2. Adam over the master runs
    const float step_size = hs[0], inv_bc2_sqrt = hs[1];
    // (four elements per thread and trip: their twelve loads are issued before the first use - with one element per trip the
    // pass ran at the latency of one load per 12 bytes)
    for (int q0 = tid; q0 < n2; q0 += 4 * 256) {
        long idx[4];
        int at[4];
        float mo[4], vo[4], po[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + j * 256;
            const bool in = q < n2;
            const int o = in ? fdiv(q, run, m_run) : 0, r = in ? q - o * run : 0;
            const int i = fdiv(r, taps, m_taps), t = r - i * taps;
            at[j] = in ? o * LO + i * LT + t : -1;
            idx[j] = ((long)(o0 + o) * Cin + i0) * taps + r;
            if (in) { mo[j] = UPD_LD(m + idx[j]); vo[j] = UPD_LD(v + idx[j]); po[j] = UPD_LD(p + idx[j]); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (at[j] < 0) continue;
            float g = sm[at[j]];
            if (terms.n) {
                const int q = q0 + j * 256;
                const int o = fdiv(q, run, m_run), r = q - o * run;
                for (int k = 0; k < terms.n; ++k) g -= hs[2 + k] * terms.u[k][o0 + o] * terms.v[k][(long)i0 * taps + r];
            }
            const float mi = beta1 * mo[j] + (1.f - beta1) * g;
            const float vi = beta2 * vo[j] + (1.f - beta2) * g * g;
            UPD_ST(m + idx[j], mi); UPD_ST(v + idx[j], vi);
            const float pn = po[j] - step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
            UPD_ST(p + idx[j], pn);
            sm[at[j]] = pn;
        }
    }
    __syncthreads();
    // ---- 3./4. operand copies, 16 bytes per store (2-byte stores ran these two passes at 0.46 TB/s). tl[sl] = the (<= 4)
    // taps slice sl sums (one tap unless the sub-pixel form)
    constexpr int EPC = elem<T>::per16;
    auto slice = [&](const float* row, int sl) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int t = tl[sl][j]; if (t >= 0) a += row[t]; }
        return a;
    };
    if (fwd && !(probe & 4)) {                                   // forward copy: [o][sl*Cin_s + i], EPC consecutive i per thread
        constexpr int IC = UT_I / EPC;
        for (int q = tid; q < no * S * IC; q += 256) {
            const int ic = q % IC, qi = q / IC, o = fdiv(qi, S, m_S), sl = qi - o * S;
            const int il = ic * EPC;
            if (il >= ni) continue;
            T* dst = fwd + (long)(o0 + o) * S * Cin_s + (long)sl * Cin_s + i0 + il;
            const float* base = sm + o * LO + il * LT;
            if (il + EPC <= ni) {
                u32x4 pk;
                T* pv = reinterpret_cast<T*>(&pk);
#pragma unroll
                for (int e = 0; e < EPC; ++e) elem<T>::st(pv + e, slice(base + e * LT, sl));
                *reinterpret_cast<u32x4*>(dst) = pk;
            } else {
                for (int e = 0; e < ni - il; ++e) elem<T>::st(dst + e, slice(base + e * LT, sl));
            }
        }
    }
    if ((bwd || lin) && !(probe & 8)) {                           // data-gradient copy: EPC consecutive o per thread
        constexpr int OC = UT_O / EPC;
        for (int q = tid; q < ni * S * OC; q += 256) {
            const int oc = q % OC, qo = q / OC, i = fdiv(qo, S, m_S), sl = qo - i * S;
            const int ol = oc * EPC;
            if (ol >= no) continue;
            T* dst = bwd ? bwd + ((long)(i0 + i) * S + sl) * Cout_s + o0 + ol : lin + ((long)sl * Cin_s + i0 + i) * Cout_s + o0 + ol;
            const float* base = sm + ol * LO + i * LT;
            if (ol + EPC <= no) {
                u32x4 pk;
                T* pv = reinterpret_cast<T*>(&pk);
#pragma unroll
                for (int e = 0; e < EPC; ++e) elem<T>::st(pv + e, slice(base + e * LO, sl));
                *reinterpret_cast<u32x4*>(dst) = pk;
            } else {
                for (int e = 0; e < no - ol; ++e) elem<T>::st(dst + e, slice(base + e * LO, sl));
            }
        }
    }
}

// The same update for DENSE layers (one tap, one slice, no spectral-norm terms: the generator's fc / fc_seg, 613 -> 32768 / 16384 - 30 M
// of its 86 M weights, the last two launches of the step's tail). The general kernel above walks a tile of 8 x 128 elements through four
// dependent phases of 4 KB each and ran these layers at 2.3-2.6 TB/s alone; here a block owns 64 output x 64 input channels, a thread 16
// elements of ONE input column: its 64 loads (accumulator, master, m, v) are all in flight before the first use, the new master goes
// through LDS once, and both operand copies leave with 16-byte stores that fill whole 128-byte lines (forward copy [o][Cin_s]: 64
// consecutive inputs of a row; transposed copy [i][Cout_s]: 64 consecutive outputs of a row). Same Adam expressions as above.
template <typename T>
__global__ __launch_bounds__(256) void layer_update_dense_kernel(const float* __restrict__ G, float* __restrict__ p, float* __restrict__ m,
                                                                 float* __restrict__ v, T* __restrict__ fwd, T* __restrict__ lin,
                                                                 const float* __restrict__ hyper, float beta1, float beta2, float eps,
                                                                 int Cout, int Cin, int Cin_s, int Cout_s, float gscale, float step_add,
                                                                 int g_bf16) {
    constexpr int TO = 64, TI = 64, LDT = TI + 1, RPW = TO / 4;        // rows per wavefront
    __shared__ float sm[TO * LDT];
    __shared__ float hs[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int o0 = blockIdx.y * TO, i0 = blockIdx.x * TI;
    const int no = Cout - o0 < TO ? Cout - o0 : TO, ni = Cin - i0 < TI ? Cin - i0 : TI;
    if (tid == 0) {
        const float t = hyper[0] + step_add, lr = hyper[1];
        hs[0] = lr / (1.f - powf(beta1, t));
        hs[1] = 1.f / sqrtf(1.f - powf(beta2, t));
    }
    const bool iok = lane < ni;
    float g[RPW], po[RPW], mo[RPW], vo[RPW];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int o = w * RPW + k;
        const bool ok = iok && o < no;
        const long gi = (long)(o0 + o) * Cin_s + i0 + lane, mi = (long)(o0 + o) * Cin + i0 + lane;
        g[k] = ok ? (g_bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(G)[gi]) : UPD_LD(G + gi)) : 0.f;
        po[k] = ok ? UPD_LD(p + mi) : 0.f;
        mo[k] = ok ? UPD_LD(m + mi) : 0.f;
        vo[k] = ok ? UPD_LD(v + mi) : 0.f;
    }
    __syncthreads();
    const float step_size = hs[0], inv_bc2_sqrt = hs[1];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int o = w * RPW + k;
        float pn = 0.f;
        if (iok && o < no) {
            const long mi_ = (long)(o0 + o) * Cin + i0 + lane;
            const float gg = g[k] * gscale;
            const float mi = beta1 * mo[k] + (1.f - beta1) * gg;
            const float vi = beta2 * vo[k] + (1.f - beta2) * gg * gg;
            UPD_ST(m + mi_, mi); UPD_ST(v + mi_, vi);
            pn = po[k] - step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
            UPD_ST(p + mi_, pn);
        }
        sm[o * LDT + lane] = pn;
    }
    __syncthreads();
    constexpr int EPC = elem<T>::per16;
    if (fwd) {                                                    // forward copy [o][Cin_s]: EPC consecutive inputs per store
        constexpr int IC = TI / EPC;
        for (int q = tid; q < TO * IC; q += 256) {
            const int ic = q % IC, o = q / IC, il = ic * EPC;
            if (o >= no || il >= ni) continue;
            T* dst = fwd + (long)(o0 + o) * Cin_s + i0 + il;
            const float* base = sm + o * LDT + il;
            if (il + EPC <= ni) {
                u32x4 pk;
                T* pv = reinterpret_cast<T*>(&pk);
#pragma unroll
                for (int e = 0; e < EPC; ++e) elem<T>::st(pv + e, base[e]);
                *reinterpret_cast<u32x4*>(dst) = pk;
            } else {
                for (int e = 0; e < ni - il; ++e) elem<T>::st(dst + e, base[e]);
            }
        }
    }
    if (lin) {                                                    // transposed copy [i][Cout_s]: EPC consecutive outputs per store
        constexpr int OC = TO / EPC;
        for (int q = tid; q < TI * OC; q += 256) {
            const int i = q % TI, oc = q / TI, ol = oc * EPC;      // (i fastest: conflict-free LDS reads)
            if (i >= ni || ol >= no) continue;
            T* dst = lin + (long)(i0 + i) * Cout_s + o0 + ol;
            const float* base = sm + ol * LDT + i;
            if (ol + EPC <= no) {
                u32x4 pk;
                T* pv = reinterpret_cast<T*>(&pk);
#pragma unroll
                for (int e = 0; e < EPC; ++e) elem<T>::st(pv + e, base[e * LDT]);
                *reinterpret_cast<u32x4*>(dst) = pk;
            } else {
                for (int e = 0; e < no - ol; ++e) elem<T>::st(dst + e, base[e * LDT]);
            }
        }
    }
}

// gw_dot += sum G[o][sl(t)*Cin_s+i] * w[o][i][t]; one thread per (o, i), block reduce, one atomic per block
__global__ void wgrad_dot_tiled_kernel(const float* __restrict__ G, const float* __restrict__ w, float* out, int Cout, int Cin,
                                       int taps, int S, TapMap inv, int Cin_s) {
    __shared__ float part[4];
    float acc = 0.f;
    const unsigned total = (unsigned)Cout * Cin;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx % (unsigned)Cin;
        const unsigned o = idx / (unsigned)Cin;
        const float* gp = G + (long)o * S * Cin_s + i;
        const float* wp = w + (long)idx * taps;
        for (int t = 0; t < taps; ++t) {
            const int sl = inv.m[t];
            if (sl >= 0) acc += gp[(long)sl * Cin_s] * wp[t];
        }
    }
    for (int off = 32; off; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// ---------------------------------------------------------------------------------------------
// spectral norm: one power iteration  (W [rows][cols] fp32) in THREE launches, no memsets, no copies
//   pass 1: tv += W^T u            (row slabs, one atomic per column and slab)
//   pass 2: tu += W tv             (one wavefront per row segment, UN-normalised tv; the waves of row 0 also add up
//           ||tv||^2)
//   finish (one block): v = tv/||tv||, W v = tu/||tv||, u = W v/||W v||, sigma = u . W v; writes u, v, their
//           snapshots for this call's backward pass, sigma and 1/sigma, and re-zeroes the accumulators.
// `work` = [tv(cols) | tu(rows) | ||tv||^2] is persistent per layer and all-zero between calls. (Finishing inside
// pass 2 by its last-arriving block was tried: the device-scope fence every block then needs costs more on this
// multi-XCD part than the extra launch.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    const int nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

__global__ void sn_wt_u_kernel(const float* __restrict__ w, const float* __restrict__ u, float* tv, int rows, int cols,
                               int rows_per_block) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += w[(long)r * cols + c] * u[r];
    atomicAdd(tv + c, acc);
}

// tu[r] += W[r][seg] . x[seg]: one wavefront per (row, 2048-column segment), 8 independent loads in flight per lane;
// segments of a row combine with one atomic each. x = tv (iterate; un-normalised) or the stored v (eval).
constexpr int SN_SEG = 2048;
__global__ void sn_w_v_kernel(const float* __restrict__ w, const float* __restrict__ x, float* tu, float* nv2, int rows,
                              int cols, int segs, int seg_len, int want_norm) {
    const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= rows * segs) return;
    const int r = wid / segs, sg = wid - r * segs;
    const int lane = threadIdx.x & 63;
    const int c0 = sg * seg_len, c1 = c0 + seg_len < cols ? c0 + seg_len : cols;
    const float* wr = w + (long)r * cols;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, nn = 0.f;
    int c = c0 + lane;
    for (; c + 7 * 64 < c1; c += 8 * 64) {
        a0 += wr[c] * x[c];             a1 += wr[c + 64] * x[c + 64];
        a2 += wr[c + 128] * x[c + 128]; a3 += wr[c + 192] * x[c + 192];
        a4 += wr[c + 256] * x[c + 256]; a5 += wr[c + 320] * x[c + 320];
        a6 += wr[c + 384] * x[c + 384]; a7 += wr[c + 448] * x[c + 448];
    }
    for (; c < c1; c += 64) a0 += wr[c] * x[c];
    float acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    const bool norm = want_norm && r == 0;
    if (norm)
        for (int cc = c0 + lane; cc < c1; cc += 64) nn += x[cc] * x[cc];
    for (int off = 32; off; off >>= 1) { acc += __shfl_xor(acc, off); nn += __shfl_xor(nn, off); }
    if (lane == 0) {
        atomicAdd(tu + r, acc);
        if (norm) atomicAdd(nv2, nn);
    }
}

__global__ void sn_finish_kernel(float* tv, float* tu, float* nv2, float* u, float* v, float* out, float* u_snap, float* v_snap,
                                 int rows, int cols, float eps, int iterate) {
    __shared__ float sh[16];
    float s;
    if (iterate) {
        const float inv_v = 1.f / fmaxf(sqrtf(*nv2), eps);                    // v = tv / max(||tv||, eps)
        float acc = 0.f;
        for (int i = threadIdx.x; i < rows; i += blockDim.x) { const float t = tu[i] * inv_v; acc += t * t; }
        const float inv_u = 1.f / fmaxf(sqrtf(block_sum(acc, sh)), eps);      // u = W v / max(||W v||, eps)
        float dot = 0.f;
        for (int i = threadIdx.x; i < rows; i += blockDim.x) {
            const float t = tu[i] * inv_v, un = t * inv_u;
            u[i] = un;
            if (u_snap) u_snap[i] = un;
            dot += un * t;
            tu[i] = 0.f;
        }
        s = block_sum(dot, sh);                                               // sigma = u . W v
        for (int i = threadIdx.x; i < cols; i += blockDim.x) {
            const float vn = tv[i] * inv_v;
            v[i] = vn;
            if (v_snap) v_snap[i] = vn;
            tv[i] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x == 0) *nv2 = 0.f;
    } else {                                                                  // eval: sigma = u_old . W v_old
        float acc = 0.f;
        for (int i = threadIdx.x; i < rows; i += blockDim.x) {
            acc += tu[i] * u[i];
            if (u_snap) u_snap[i] = u[i];
            tu[i] = 0.f;
        }
        s = block_sum(acc, sh);
        if (v_snap)
            for (int i = threadIdx.x; i < cols; i += blockDim.x) v_snap[i] = v[i];
    }
    if (threadIdx.x == 0) { out[0] = s; out[1] = 1.f / s; }
}

// ---- multi-tensor form: ONE launch triple runs the power iteration of MANY layers (all spectral-normed layers of the three
// critics: 54 evaluations per step used to be 162 launches of 6-14 us each, pure launch latency). jobs[] lives in device
// memory; start1/start2 are the prefix sums of the per-job block counts of pass 1 / pass 2 (cpcsv_sn_multi_blocks).
__device__ __forceinline__ int sn_find_job(const int* __restrict__ start, int njobs, int b) {
    int j = 0;
    while (j + 1 < njobs && b >= start[j + 1]) ++j;
    return j;
}

__global__ void sn_multi_wt_u_kernel(const cpcsv_sn_job* __restrict__ jobs, const int* __restrict__ start, int njobs, int rpb_all) {
    const int j = sn_find_job(start, njobs, blockIdx.x);
    const cpcsv_sn_job jb = jobs[j];
    const int local = blockIdx.x - start[j];
    const int gx = (jb.cols + 255) / 256;
    const int bx = local % gx, by = local / gx;
    const int rpb = rpb_all > 0 ? rpb_all : jb.rows;
    const int r0 = by * rpb;
    const int r1 = r0 + rpb < jb.rows ? r0 + rpb : jb.rows;
    if (rpb_all == 32 && (jb.cols & 3) == 0 && ((uintptr_t)jb.w & 15) == 0) {
        // 64 column chunks of four x 4 row lanes, eight rows per thread: eight independent 16-byte loads in flight (the scalar walk
        // below has one 4-byte load per trip: 2.4 TB/s over the critics' 92 MB of master weights). Not in the deterministic mode
        // (one slab = all rows, sequential sum).
        __shared__ float part[4][256];
        const int cx = threadIdx.x & 63, rl = threadIdx.x >> 6;
        const int c4 = bx * 256 + cx * 4;
        f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < jb.cols) {
            f32x4 wv[8];
            float uv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = r0 + rl + 4 * k;
                const bool ok = r < r1;
                wv[k] = ok ? *reinterpret_cast<const f32x4*>(jb.w + (long)r * jb.cols + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
                uv[k] = ok ? jb.u[r] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc4 += wv[k] * uv[k];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) part[rl][cx * 4 + e] = acc4[e];
        __syncthreads();
        const int c = bx * 256 + threadIdx.x;
        if (c < jb.cols)
            atomicAdd(jb.work + c, (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
        return;
    }
    const int c = bx * 256 + threadIdx.x;
    if (c >= jb.cols) return;
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += jb.w[(long)r * jb.cols + c] * jb.u[r];
    atomicAdd(jb.work + c, acc);
}

__global__ void sn_multi_w_v_kernel(const cpcsv_sn_job* __restrict__ jobs, const int* __restrict__ start, int njobs, int seg_all,
                                    int iterate) {
    const int j = sn_find_job(start, njobs, blockIdx.x);
    const cpcsv_sn_job jb = jobs[j];
    const int seg_len = seg_all > 0 ? seg_all : jb.cols;
    const int segs = (jb.cols + seg_len - 1) / seg_len;
    const int wid = (blockIdx.x - start[j]) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= jb.rows * segs) return;
    const int r = wid / segs, sg = wid - r * segs;
    const int lane = threadIdx.x & 63;
    const int c0 = sg * seg_len, c1 = c0 + seg_len < jb.cols ? c0 + seg_len : jb.cols;
    const float* wr = jb.w + (long)r * jb.cols;
    const float* x = iterate ? jb.work : jb.v;
    float* tu = jb.work + jb.cols;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, nn = 0.f;
    int c = c0 + lane;
    if (seg_all > 0 && ((jb.cols | seg_len) & 3) == 0 && (((uintptr_t)jb.w | (uintptr_t)x) & 15) == 0) {
        // 16-byte loads: four consecutive columns per lane, four pairs in flight (not in the deterministic mode: other summation order)
        f32x4 s4[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        int cv = c0 + lane * 4;
        for (; cv + 3 * 256 < c1; cv += 4 * 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                s4[k] += *reinterpret_cast<const f32x4*>(wr + cv + k * 256) * *reinterpret_cast<const f32x4*>(x + cv + k * 256);
        }
        for (; cv < c1; cv += 256) s4[0] += *reinterpret_cast<const f32x4*>(wr + cv) * *reinterpret_cast<const f32x4*>(x + cv);
        const f32x4 t4 = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        a0 = (t4[0] + t4[1]) + (t4[2] + t4[3]);
        c = c1;                                            // nothing left for the scalar walk
    }
    for (; c + 7 * 64 < c1; c += 8 * 64) {
        a0 += wr[c] * x[c];             a1 += wr[c + 64] * x[c + 64];
        a2 += wr[c + 128] * x[c + 128]; a3 += wr[c + 192] * x[c + 192];
        a4 += wr[c + 256] * x[c + 256]; a5 += wr[c + 320] * x[c + 320];
        a6 += wr[c + 384] * x[c + 384]; a7 += wr[c + 448] * x[c + 448];
    }
    for (; c < c1; c += 64) a0 += wr[c] * x[c];
    float acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    const bool norm = iterate && r == 0;
    if (norm)
        for (int cc = c0 + lane; cc < c1; cc += 64) nn += x[cc] * x[cc];
    for (int off = 32; off; off >>= 1) { acc += __shfl_xor(acc, off); nn += __shfl_xor(nn, off); }
    if (lane == 0) {
        atomicAdd(tu + r, acc);
        if (norm) atomicAdd(jb.work + jb.cols + jb.rows, nn);
    }
}

__global__ void sn_multi_finish_kernel(const cpcsv_sn_job* __restrict__ jobs, float eps, int iterate) {
    const cpcsv_sn_job jb = jobs[blockIdx.x];
    float* tv = jb.work;
    float* tu = jb.work + jb.cols;
    float* nv2 = jb.work + jb.cols + jb.rows;
    float* u_snap = jb.out + 2;
    float* v_snap = jb.out + 2 + jb.rows;
    __shared__ float sh[16];
    float s;
    if (iterate) {
        const float inv_v = 1.f / fmaxf(sqrtf(*nv2), eps);
        float acc = 0.f;
        for (int i = threadIdx.x; i < jb.rows; i += blockDim.x) { const float t = tu[i] * inv_v; acc += t * t; }
        const float inv_u = 1.f / fmaxf(sqrtf(block_sum(acc, sh)), eps);
        float dot = 0.f;
        for (int i = threadIdx.x; i < jb.rows; i += blockDim.x) {
            const float t = tu[i] * inv_v, un = t * inv_u;
            jb.u[i] = un;
            u_snap[i] = un;
            dot += un * t;
            tu[i] = 0.f;
        }
        s = block_sum(dot, sh);
        for (int i = threadIdx.x; i < jb.cols; i += blockDim.x) {
            const float vn = tv[i] * inv_v;
            jb.v[i] = vn;
            v_snap[i] = vn;
            tv[i] = 0.f;
        }
        __syncthreads();
        if (threadIdx.x == 0) *nv2 = 0.f;
    } else {
        float acc = 0.f;
        for (int i = threadIdx.x; i < jb.rows; i += blockDim.x) {
            acc += tu[i] * jb.u[i];
            u_snap[i] = jb.u[i];
            tu[i] = 0.f;
        }
        s = block_sum(acc, sh);
        for (int i = threadIdx.x; i < jb.cols; i += blockDim.x) v_snap[i] = jb.v[i];
    }
    if (threadIdx.x == 0) { jb.out[0] = s; jb.out[1] = 1.f / s; }
}

// ---- ONE pass over W per power iteration (round 6). The two-pass form above reads the fp32 master twice per iteration - W^T u, then
// W (W^T u) - 2 GB of the step's HBM traffic. Here a block owns a slab of SN1_CB = 32 COLUMNS and ALL rows, held in REGISTERS (thread =
// one column x every eighth row: up to 128 values): its t_c = sum_r W[r][c] u[r] are complete inside the block, so the second
// product's share  y_r += sum_{c in slab} W[r][c] t_c  comes from the registers without touching memory again. The blocks' y shares
// go to a partial buffer P[block][rows] (fixed order, no atomics), sn_multi_rowsum_kernel adds them up into the accumulators the
// unchanged finishing kernel reads (scale invariance: v = t/|t|, u = y/|y|, sigma = |y|/|t| need no normalised t in between).
// rows <= 1024 (the model's spectral-normed layers: <= 992); every load of a thread is issued before the first use (127 KB in
// flight per block).
constexpr int SN1_CB = 32;
// thread = FOUR consecutive columns (tx = tid & 7) x every 32nd row (ty = tid >> 3): KQ = ceil(rows / 32) <= 32 quads in registers. The
// row products of the second pass then need a 3-step butterfly over the 8 column groups per 4 elements (a column per lane needed 5
// steps per element: the kernel was bound by those shuffles, 57 us per launch against 37). VEC: aligned 16-byte loads (row length a
// multiple of 4 and W 16-byte aligned); else - the head conv's rows are 1481 * 9 floats long - four scalar loads.
template <int KQ, bool VEC>
__device__ __forceinline__ void sn_onepass_body(const cpcsv_sn_job& jb, int local, float* __restrict__ P, int nblk, float* ush, float (*red)[SN1_CB + 1],
                                                float* tsh, float* ysh) {
    const int tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;
    const int rows = jb.rows, cols = jb.cols;
    const int c = local * SN1_CB + tx * 4;
    for (int i = tid; i < KQ * 32; i += 256) ush[i] = i < rows ? jb.u[i] : 0.f;
    f32x4 w[KQ];
    const float* wp = jb.w + c;
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const int r = ty + 32 * k;
        const float* q = wp + (long)r * cols;
        if (VEC) {
            w[k] = (c < cols && r < rows) ? *reinterpret_cast<const f32x4*>(q) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            // (a dword-aligned 16-byte load - hipcc emits global_load_dwordx4 for a 4-byte aligned vector type, the target runs in
            // unaligned access mode - measured no faster than four scalar loads here: 38.0 against 37.0 us)
            const bool rok = r < rows;
            w[k][0] = (rok && c < cols) ? q[0] : 0.f;
            w[k][1] = (rok && c + 1 < cols) ? q[1] : 0.f;
            w[k][2] = (rok && c + 2 < cols) ? q[2] : 0.f;
            w[k][3] = (rok && c + 3 < cols) ? q[3] : 0.f;
        }
    }
    __syncthreads();
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KQ; ++k) acc += w[k] * ush[ty + 32 * k];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ty][tx * 4 + e] = acc[e];
    __syncthreads();
    if (tid < SN1_CB) {
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 32; ++y) t += red[y][tid];
        tsh[tid] = t;
        float nn = 0.f;
        if (local * SN1_CB + tid < cols) { jb.work[local * SN1_CB + tid] = t; nn = t * t; }
#pragma unroll
        for (int o = 16; o; o >>= 1) nn += __shfl_xor(nn, o);
        if (tid == 0) P[(long)nblk * rows + local] = nn;           // |t|^2 share of this slab, behind the y shares
    }
    __syncthreads();
    const f32x4 t4 = f32x4{tsh[tx * 4], tsh[tx * 4 + 1], tsh[tx * 4 + 2], tsh[tx * 4 + 3]};
#pragma unroll
    for (int k = 0; k < KQ; ++k) {
        const f32x4 p4 = w[k] * t4;
        float pr = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        pr += __shfl_xor(pr, 1); pr += __shfl_xor(pr, 2); pr += __shfl_xor(pr, 4);      // over the 8 column groups of the slab
        if (tx == 0) ysh[ty + 32 * k] = pr;
    }
    __syncthreads();
    float* Pb = P + (long)local * rows;
    for (int i = tid; i < rows; i += 256) Pb[i] = ysh[i];
}

__global__ __launch_bounds__(256) void sn_multi_onepass_kernel(const cpcsv_sn_job* __restrict__ jobs, const int* __restrict__ start, int njobs,
                                                               float* __restrict__ part, const long long* __restrict__ part_off) {
    __shared__ float ush[1024], ysh[1024], tsh[SN1_CB];
    __shared__ float red[32][SN1_CB + 1];
    const int j = sn_find_job(start, njobs, blockIdx.x);
    const cpcsv_sn_job jb = jobs[j];
    const int local = blockIdx.x - start[j], nblk = start[j + 1] - start[j];
    float* P = part + part_off[j];
    const bool vec = (jb.cols & 3) == 0 && ((uintptr_t)jb.w & 15) == 0;
    if (jb.rows <= 256) { if (vec) sn_onepass_body<8, true>(jb, local, P, nblk, ush, red, tsh, ysh); else sn_onepass_body<8, false>(jb, local, P, nblk, ush, red, tsh, ysh); }
    else if (jb.rows <= 512) { if (vec) sn_onepass_body<16, true>(jb, local, P, nblk, ush, red, tsh, ysh); else sn_onepass_body<16, false>(jb, local, P, nblk, ush, red, tsh, ysh); }
    else { if (vec) sn_onepass_body<32, true>(jb, local, P, nblk, ush, red, tsh, ysh); else sn_onepass_body<32, false>(jb, local, P, nblk, ush, red, tsh, ysh); }
}

// y[r] = sum over the slabs of P[slab][r] -> the tu accumulator; |t|^2 = sum of the slabs' shares -> nv2. Block = 16 rows x 16 slab lanes,
// grid (ceil(max rows / 16), jobs); fixed summation order.
__global__ __launch_bounds__(256) void sn_multi_rowsum_kernel(const cpcsv_sn_job* __restrict__ jobs, const int* __restrict__ start, int njobs,
                                                              const float* __restrict__ part, const long long* __restrict__ part_off) {
    __shared__ float red[16][17];
    __shared__ float sh[16];
    const int j = blockIdx.y;
    const cpcsv_sn_job jb = jobs[j];
    const int rows = jb.rows, nblk = start[j + 1] - start[j];
    const int r0 = blockIdx.x * 16;
    if (r0 >= rows) return;
    const float* P = part + part_off[j];
    const int rr = threadIdx.x & 15, bl = threadIdx.x >> 4, r = r0 + rr;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (r < rows) {
        int b = bl;
        for (; b + 48 < nblk; b += 64) {
            a0 += P[(long)b * rows + r]; a1 += P[(long)(b + 16) * rows + r];
            a2 += P[(long)(b + 32) * rows + r]; a3 += P[(long)(b + 48) * rows + r];
        }
        for (; b < nblk; b += 16) a0 += P[(long)b * rows + r];
    }
    red[bl][rr] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 16 && r0 + threadIdx.x < rows) {
        float sacc = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) sacc += red[y][threadIdx.x];
        jb.work[jb.cols + r0 + threadIdx.x] = sacc;
    }
    if (blockIdx.x == 0) {
        float nn = 0.f;
        for (int b = threadIdx.x; b < nblk; b += 256) nn += P[(long)nblk * rows + b];
        nn = block_sum(nn, sh);
        if (threadIdx.x == 0) jb.work[jb.cols + rows] = nn;
    }
}

// out[c] += sum over rows of x[r][c]  (bias gradients), c < C; one thread per column per row slab
// block = 64 columns x 4 row lanes, four independent loads in flight per lane (one column per thread walking its rows with
// a dependent load per trip took 16 us for the 12-60 rows of the text-encoder layers, on the tail of the backward pass)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* out, long rows, int C, int Cs, int rows_per_block) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        long r = r0 + ry;
        for (; r + 12 < r1; r += 16) {
            const float v0 = elem<T>::ld(x + r * Cs + c), v1 = elem<T>::ld(x + (r + 4) * Cs + c);
            const float v2 = elem<T>::ld(x + (r + 8) * Cs + c), v3 = elem<T>::ld(x + (r + 12) * Cs + c);
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; r < r1; r += 4) a0 += elem<T>::ld(x + r * Cs + c);
    }
    part[ry][cx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ry == 0 && c < C) atomicAdd(out + c, (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
}

// block = cw chunk columns x (256/cw) row lanes; row slabs sized for >= ~2048 blocks on large tensors
inline void ew_geometry(int cpr, long rows, int& cw, int& rpb, dim3& grid) {
    cw = 1;
    while (cw * 2 <= cpr && cw * 2 <= 256) cw *= 2;
    const int rl = 256 / cw;
    const int gx = cdiv(cpr, cw);
    constexpr int ew_rows = 8;        // (swept in round 3; knob retired)
    long per = (long)rl * ew_rows;                             // rows per thread
    long gy = (rows + per - 1) / per;
    const long cap = 4096 / gx > 1 ? 4096 / gx : 1;
    if (gy > cap) { gy = cap; per = (rows + gy - 1) / gy; gy = (rows + per - 1) / per; }
    rpb = (int)per;
    grid = dim3((unsigned)gx, (unsigned)gy);
}

inline int grid_for(long n, int block = 256, int cap = 2048 * 4) {
    long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

static BnG make_groups(const cpcsv_bn_groups* g, long rows) {
    BnG G;
    G.n = 1; G.pstride = 0; G.nph = 1; G.TM = 0;
    for (int k = 0; k < 5; ++k) { G.row[k] = rows; G.tile[k] = 0; }
    G.row[0] = 0;
    for (int k = 0; k < 4; ++k) G.sigma[k] = nullptr;
    if (g) {
        G.n = g->n; G.pstride = g->pstride; G.nph = g->nph > 0 ? g->nph : 1; G.TM = g->TM;
        for (int k = 0; k < 5; ++k) { G.row[k] = k <= g->n ? g->row[k] : g->row[g->n]; G.tile[k] = k <= g->n ? g->tile[k] : g->tile[g->n]; }
        for (int k = 0; k < 4; ++k) G.sigma[k] = k < g->n ? g->sigma[k] : nullptr;
    }
    return G;
}
static bool groups_ok(const cpcsv_bn_groups* g, long rows) {
    if (!g) return true;
    if (g->n < 1 || g->n > 4 || g->row[0] != 0 || g->row[g->n] != rows) return false;
    for (int k = 0; k < g->n; ++k) if (g->row[k + 1] <= g->row[k]) return false;
    return true;
}
static long max_group_rows(const BnG& G) {
    long m = 0;
    for (int k = 0; k < G.n; ++k) m = G.row[k + 1] - G.row[k] > m ? G.row[k + 1] - G.row[k] : m;
    return m;
}

extern "C" int cpcsv_bn_finalize(const float* partials, int mtiles, int ldstat, long count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float* mean,
                                 float* invstd, float* scale, float* shift, int C, int Cs, float eps,
                                 float momentum, int update_running, float* bwd_sums, const cpcsv_bn_groups* groups, void* stream) {
    if (!partials || count <= 0 || C <= 0 || Cs < C) return -1001;
    BnG G;
    long per_row = 1;
    if (groups) {
        // rows of the groups are in units of `count / row[n]` statistics samples each (pixels per GEMM row: 1, or 4 for the
        // sub-pixel form whose groups are given in low-resolution rows)
        if (groups->n < 1 || groups->n > 4 || groups->row[groups->n] <= 0 || count % groups->row[groups->n]) return -1002;
        G = make_groups(groups, groups->row[groups->n]);
        per_row = count / groups->row[groups->n];
        if (G.TM <= 0) G.TM = G.tile[G.n];
    } else {
        G = make_groups(nullptr, count);
        G.tile[0] = 0;
        for (int k = 1; k < 5; ++k) G.tile[k] = mtiles;
        G.TM = mtiles;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(Cs, 16)), dim3(1024), 0, (hipStream_t)stream, partials,
                       ldstat, gamma, beta, running_mean, running_var, mean, invstd,
                       scale, shift, C, Cs, eps, momentum, update_running, bwd_sums, G, per_row);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_bn_apply(const void* x, void* y, int dtype, const float* scale, const float* shift, long rows,
                              int C, int Cs, int act, const cpcsv_bn_groups* groups, void* stream) {
    if (!x || !y || Cs % 8 || !groups_ok(groups, rows)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const BnG G = make_groups(groups, rows);
    const long grows = max_group_rows(G);
    int cw, rpb; dim3 grid;
    if (dtype == CPCSV_BF16) {
        const int cpr = Cs / 8;
        ew_geometry(cpr, grows, cw, rpb, grid);
        grid.z = G.n;
        if (act >= CPCSV_ACT_TANH) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, scale, shift, cpr, cw, rpb, C, act, G);
        else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, scale, shift, cpr, cw, rpb, C, act, G);
    } else {
        const int cpr = Cs / 4;
        ew_geometry(cpr, grows, cw, rpb, grid);
        grid.z = G.n;
        if (act >= CPCSV_ACT_TANH) hipLaunchKernelGGL((bn_apply_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, (float*)y, scale, shift, cpr, cw, rpb, C, act, G);
        else hipLaunchKernelGGL((bn_apply_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, (float*)y, scale, shift, cpr, cw, rpb, C, act, G);
    }
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_bn_apply_partials(const void* x, void* y, int dtype, const float* partials, int ldstat, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var, float* stat_out, float* bwd_sums,
                                       long rows, int C, int Cs, int act, float eps, float momentum, const cpcsv_bn_groups* groups,
                                       void* stream) {
    if (!x || !y || !partials || !gamma || !beta || !stat_out || !groups || Cs % 8 || C <= 0 || Cs < C || !groups_ok(groups, rows)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    BnG G = make_groups(groups, rows);
    if (G.TM <= 0) G.TM = G.tile[G.n];
    if ((long)G.tile[G.n] * G.nph > 64) return -1002;            // a handful of partial rows only: every block sums them itself
    const long grows = max_group_rows(G);
    int cw, rpb; dim3 grid;
    if (dtype == CPCSV_BF16) {
        const int cpr = Cs / 8;
        ew_geometry(cpr, grows, cw, rpb, grid);
        grid.z = G.n;
        if (act >= CPCSV_ACT_TANH) hipLaunchKernelGGL((bn_apply_fused_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, partials, ldstat,
                           gamma, beta, running_mean, running_var, stat_out, bwd_sums, cpr, cw, rpb, C, Cs, act, eps, momentum, G);
        else hipLaunchKernelGGL((bn_apply_fused_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, partials, ldstat,
                           gamma, beta, running_mean, running_var, stat_out, bwd_sums, cpr, cw, rpb, C, Cs, act, eps, momentum, G);
    } else {
        const int cpr = Cs / 4;
        ew_geometry(cpr, grows, cw, rpb, grid);
        grid.z = G.n;
        if (act >= CPCSV_ACT_TANH) hipLaunchKernelGGL((bn_apply_fused_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, (float*)y, partials, ldstat,
                           gamma, beta, running_mean, running_var, stat_out, bwd_sums, cpr, cw, rpb, C, Cs, act, eps, momentum, G);
        else hipLaunchKernelGGL((bn_apply_fused_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, (float*)y, partials, ldstat,
                           gamma, beta, running_mean, running_var, stat_out, bwd_sums, cpr, cw, rpb, C, Cs, act, eps, momentum, G);
    }
    CPCSV_CHECK_LAUNCH();
    return 0;
}


template <typename T>
static int bn_bwd_reduce_t(const void* dy, const void* x, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, float* sums, long rows_all, int C, int Cs, int act, const BnG& G, hipStream_t s) {
    const long rows = max_group_rows(G);
    constexpr int EPC = elem<T>::per16;
    const int cpr = Cs / EPC;
    constexpr int cw_cap = 8;         // (swept in rounds 3-4; knob retired)
    int cw = 1;
    // a handful of rows over thousands of channels (BatchNorm1d of the generator's fc / fc_seg: 60 rows x 16384): 64 channel chunks x
    // 4 row lanes per block and ONE slab per group. With 8 x 32 the launch spent 120 us on two rows per thread, a 31-step serial
    // reduction over the row lanes and 512 blocks' worth of atomics.
    const bool few_rows = rows <= 128 && cpr >= 256 && !g_cpcsv_deterministic;
    while (cw * 2 <= cpr && cw * 2 <= (few_rows ? 64 : cw_cap)) cw *= 2;
    const int rl = 256 / cw;
    // enough row slabs to stream at full bandwidth (~256 blocks: alone on the 64x64 maps 5.4 TB/s against 3.7 with 1024 - fewer slabs
    // mean fewer float atomics and longer streams per block; 128 is slower again); their atomics are spread over the accumulator
    // copies, so a column address sees gy / CPCSV_BN_SUM_COPIES of them
    const int gx = cdiv(cpr, cw);
    constexpr int red_cap = 256;      // (128 / 1024 slower, section 4.5b; knob retired)
    const int cap = red_cap / gx > 1 ? red_cap / gx : 1;
    long rpb = 16L * rl;                                        // 16 rows per thread ...
    while (rpb > 4L * rl && (rows + rpb - 1) / rpb * gx < 256) rpb >>= 1;   // ... fewer when that leaves CUs without a block
    if (few_rows) rpb = rows;
    int gy = (int)((rows + rpb - 1) / rpb);
    if (gy > cap) { gy = cap; rpb = (rows + gy - 1) / gy; gy = (int)((rows + rpb - 1) / rpb); }
    if (g_cpcsv_deterministic && gy > CPCSV_BN_SUM_COPIES) {
        // every accumulator copy gets exactly ONE slab's contribution (0 + x is exact), the copies are summed in order
        gy = CPCSV_BN_SUM_COPIES; rpb = (rows + gy - 1) / gy; gy = (int)((rows + rpb - 1) / rpb);
    }
    const size_t shmem = (size_t)rl * cw * 2 * EPC * sizeof(float);
    if (act >= CPCSV_ACT_TANH)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true>), dim3(cdiv(cpr, cw), gy, G.n), dim3(256), shmem, s, (const T*)dy, (const T*)x,
                           mean, invstd, gamma, beta, sums, C, Cs, cpr, cw, (int)rpb, act, G);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false>), dim3(cdiv(cpr, cw), gy, G.n), dim3(256), shmem, s, (const T*)dy, (const T*)x,
                           mean, invstd, gamma, beta, sums, C, Cs, cpr, cw, (int)rpb, act, G);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_bn_bwd_reduce(const void* dy, const void* x, int dtype, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, float* sums, long rows, int C, int Cs, int act,
                                   const cpcsv_bn_groups* groups, void* stream) {
    if (!dy || !x || !sums || Cs % 8 || !groups_ok(groups, rows)) return -1001;
    const BnG G = make_groups(groups, rows);
    return dtype == CPCSV_BF16 ? bn_bwd_reduce_t<bf16_t>(dy, x, mean, invstd, gamma, beta, sums, rows, C, Cs, act, G, (hipStream_t)stream)
                               : bn_bwd_reduce_t<float>(dy, x, mean, invstd, gamma, beta, sums, rows, C, Cs, act, G, (hipStream_t)stream);
}

extern "C" int cpcsv_bn_bwd_apply(const void* dy, const void* x, void* dx, int dtype, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, const float* sums,
                                  float* dgamma, float* dbeta, long rows, int C, int Cs, int act, int accumulate,
                                  float* gw_out, const float* sigma, float eps, const cpcsv_bn_groups* groups, void* stream) {
    if (!dy || !x || !dx || Cs % 8 || !groups_ok(groups, rows)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    BnG G = make_groups(groups, rows);
    if (!groups) G.sigma[0] = sigma;                 // single pass: `sigma` is that pass's {sigma, 1/sigma}
    for (int k = 0; k < G.n; ++k) if (gw_out && !G.sigma[k]) return -1001;
    const long grows = max_group_rows(G);
    int cw, rpb; dim3 grid;
    if (dtype == CPCSV_BF16) {
        const int cpr = Cs / 8;
        ew_geometry(cpr, grows, cw, rpb, grid);
        grid.z = G.n;
        if (act >= CPCSV_ACT_TANH)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                               (bf16_t*)dx, mean, invstd, gamma, beta, sums, dgamma, dbeta, cpr, cw, rpb, C, Cs, act, accumulate,
                               gw_out, eps, G);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x,
                               (bf16_t*)dx, mean, invstd, gamma, beta, sums, dgamma, dbeta, cpr, cw, rpb, C, Cs, act, accumulate,
                               gw_out, eps, G);
    } else {
        const int cpr = Cs / 4;
        ew_geometry(cpr, grows, cw, rpb, grid);
        grid.z = G.n;
        if (act >= CPCSV_ACT_TANH)
            hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), grid, dim3(256), 0, s, (const float*)dy, (const float*)x,
                               (float*)dx, mean, invstd, gamma, beta, sums, dgamma, dbeta, cpr, cw, rpb, C, Cs, act, accumulate,
                               gw_out, eps, G);
        else
            hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), grid, dim3(256), 0, s, (const float*)dy, (const float*)x,
                               (float*)dx, mean, invstd, gamma, beta, sums, dgamma, dbeta, cpr, cw, rpb, C, Cs, act, accumulate,
                               gw_out, eps, G);
    }
    CPCSV_CHECK_LAUNCH();
    return 0;
}

static TapMap make_map(const int8_t* m, int n, int identity_n) {
    TapMap t;
    for (int i = 0; i < CPCSV_MAX_TAPS; ++i) t.m[i] = m ? (i < n ? m[i] : -1) : (i < identity_n ? (int8_t)i : -1);
    return t;
}
static TapMap invert(const TapMap& f, int S, int taps) {
    TapMap inv;
    for (int i = 0; i < CPCSV_MAX_TAPS; ++i) inv.m[i] = -1;
    for (int sl = 0; sl < S; ++sl) if (f.m[sl] >= 0 && f.m[sl] < taps) inv.m[f.m[sl]] = (int8_t)sl;
    return inv;
}
static MaskTab make_masks(const uint16_t* m, int S) {
    MaskTab t;
    for (int i = 0; i < CPCSV_MAX_TAPS; ++i) t.m[i] = (m && i < S) ? m[i] : 0;
    return t;
}

// blocks per weight row: one block walks a whole row when there are many rows; rows are split only when there are
// too few of them to fill the chip (each block should still see several KB)
static int row_blocks(int channels, int rows) {
    const int chunks = cdiv(channels, PK_I);
    int per_row = cdiv(1024, rows > 0 ? rows : 1);
    if (per_row > chunks) per_row = chunks;
    return per_row < 1 ? 1 : per_row;
}

template <typename T>
static int pack_all(const float* w, void* dst_fwd, void* dst_bwd, void* dst_lin, int Cout, int Cin, int taps, int S,
                    const TapMap& map, const MaskTab& mk, int sum, int Cin_s, int Cout_s, hipStream_t s) {
    if (dst_fwd) {
        hipLaunchKernelGGL(pack_fwd_kernel<T>, dim3(grid_for((long)Cout * Cin_s)), dim3(256), 0, s, w, (T*)dst_fwd, Cout, Cin, taps, S,
                           map, mk, sum, Cin_s);
        CPCSV_CHECK_LAUNCH();
    }
    if (dst_bwd) {
        hipLaunchKernelGGL(pack_bwd_kernel<T>, dim3(cdiv(Cin, PB_I), cdiv(Cout_s, PB_O)), dim3(256), 0, s, w, (T*)dst_bwd, Cout, Cin,
                           taps, S, map, mk, sum, Cin_s, Cout_s, 1);
        CPCSV_CHECK_LAUNCH();
    }
    if (dst_lin) {
        hipLaunchKernelGGL(pack_bwd_kernel<T>, dim3(cdiv(Cin_s, PB_I), cdiv(Cout_s, PB_O)), dim3(256), 0, s, w, (T*)dst_lin, Cout, Cin,
                           taps, S, map, mk, sum, Cin_s, Cout_s, 2);
        CPCSV_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int cpcsv_colsum(const void* x, int dtype, float* out, long rows, int C, int Cs, void* stream) {
    if (!x || !out) return -1001;
    const int rpb = g_cpcsv_deterministic ? (int)(rows > 0 ? rows : 1) : 256;      // deterministic: one row slab per column
    const dim3 grid(cdiv(C, 64), cdiv(rows, rpb));
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, rows, C, Cs, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, out, rows, C, Cs, rpb);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_pack_weight(const float* w, void* dst_fwd, void* dst_bwd, void* dst_lin, int dtype, int Cout,
                                 int Cin, int taps, int S, const int8_t* tapmap, int Cin_s, int Cout_s, void* stream) {
    if (!w || Cin_s % 8 || Cout_s % 8 || Cin_s < Cin || Cout_s < Cout || S < 1 || S > CPCSV_MAX_TAPS || taps > CPCSV_MAX_TAPS) return -1001;
    const TapMap map = make_map(tapmap, S, taps);
    const MaskTab mk = make_masks(nullptr, 0);
    return dtype == CPCSV_BF16 ? pack_all<bf16_t>(w, dst_fwd, dst_bwd, dst_lin, Cout, Cin, taps, S, map, mk, 0, Cin_s, Cout_s, (hipStream_t)stream)
                               : pack_all<float>(w, dst_fwd, dst_bwd, dst_lin, Cout, Cin, taps, S, map, mk, 0, Cin_s, Cout_s, (hipStream_t)stream);
}

extern "C" int cpcsv_pack_weight_sum(const float* w, void* dst_fwd, void* dst_bwd, int dtype, int Cout, int Cin, int taps,
                                     int S, const uint16_t* masks, int Cin_s, int Cout_s, void* stream) {
    if (!w || !masks || Cin_s % 8 || Cout_s % 8 || S < 1 || S > CPCSV_MAX_TAPS || taps > CPCSV_MAX_TAPS) return -1001;
    const TapMap map = make_map(nullptr, 0, 0);
    const MaskTab mk = make_masks(masks, S);
    return dtype == CPCSV_BF16 ? pack_all<bf16_t>(w, dst_fwd, dst_bwd, nullptr, Cout, Cin, taps, S, map, mk, 1, Cin_s, Cout_s, (hipStream_t)stream)
                               : pack_all<float>(w, dst_fwd, dst_bwd, nullptr, Cout, Cin, taps, S, map, mk, 1, Cin_s, Cout_s, (hipStream_t)stream);
}

extern "C" int cpcsv_wgrad_dot(const float* G, const float* w, float* gw_dot, int Cout, int Cin, int taps, int S,
                               const int8_t* tapmap, int Cin_s, void* stream) {
    if (!G || !w || !gw_dot) return -1001;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(gw_dot, 0, sizeof(float), s);
    if (e != hipSuccess) return -(int)e;
    const TapMap inv = invert(make_map(tapmap, S, taps), S, taps);
    hipLaunchKernelGGL(wgrad_dot_tiled_kernel, dim3(g_cpcsv_deterministic ? 1 : grid_for((long)Cout * Cin, 256, 1024)), dim3(256), 0, s, G, w,
                       gw_dot, Cout, Cin, taps, S, inv, Cin_s);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_unpack_wgrad(float* G, float* dw, const float* sigma, const float* u, const float* v,
                                  const float* gw_dot, int Cout, int Cin, int taps, int S, const int8_t* tapmap,
                                  int Cin_s, int accumulate, int rezero, void* stream) {
    if (!G || !dw) return -1001;
    if (sigma && ((u || v || gw_dot) && (!u || !v || !gw_dot))) return -1002;     // all three or none (rank-1 term = 0)
    const TapMap inv = invert(make_map(tapmap, S, taps), S, taps);
    hipLaunchKernelGGL(unpack_tiled_kernel, dim3(grid_for((long)Cout * Cin)), dim3(256), 0, (hipStream_t)stream, G, dw, sigma, u, v,
                       gw_dot, Cout, Cin, taps, S, inv, make_masks(nullptr, 0), 0, Cin_s, accumulate, rezero);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

// dw[o][k] -= (gw / sigma^2) * u[o] * v[k]: the d sigma / dW_orig term of ONE call of a spectral-normed layer, for
// accumulators whose contributions were already divided by that call's sigma (cpcsv_bn_bwd_apply folds 1/sigma into dz)
__global__ void rank1_sub_kernel(float* __restrict__ dw, const float* __restrict__ gw, const float* __restrict__ sigma,
                                 const float* __restrict__ u, const float* __restrict__ v, long rows, long cols) {
    const float sg = sigma[0], coef = gw[0] / (sg * sg);
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        dw[i] -= coef * u[i / cols] * v[i % cols];
}

extern "C" int cpcsv_rank1_sub(float* dw, const float* gw, const float* sigma, const float* u, const float* v, long rows,
                               long cols, void* stream) {
    if (!dw || !gw || !sigma || !u || !v || rows <= 0 || cols <= 0) return -1001;
    hipLaunchKernelGGL(rank1_sub_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, dw, gw, sigma, u, v, rows, cols);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_unpack_wgrad_sum(float* G, float* dw, int Cout, int Cin, int taps, int S, const uint16_t* masks,
                                      int Cin_s, int accumulate, int rezero, void* stream) {
    if (!G || !dw || !masks || S < 1 || S > CPCSV_MAX_TAPS) return -1001;
    hipLaunchKernelGGL(unpack_tiled_kernel, dim3(grid_for((long)Cout * Cin)), dim3(256), 0, (hipStream_t)stream, G, dw,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, Cout, Cin, taps, S,
                       make_map(nullptr, 0, 0), make_masks(masks, S), 1, Cin_s, accumulate, rezero);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_spectral_sigma(const float* w, float* u, float* v, float* out, float* work, int rows, int cols,
                                    int iterate, int snapshot, void* stream) {
    if (!w || !u || !v || !out || !work) return -1001;
    hipStream_t s = (hipStream_t)stream;
    float* tv = work;              // [cols]   all-zero between calls
    float* tu = work + cols;       // [rows]
    float* nv2 = work + cols + rows;
    float* u_snap = snapshot ? out + 2 : nullptr;
    float* v_snap = snapshot ? out + 2 + rows : nullptr;
    const float eps = 1e-12f;
    if (iterate) {
        const int rpb = g_cpcsv_deterministic ? rows : 32;       // deterministic: one row slab, no atomics between blocks
        hipLaunchKernelGGL(sn_wt_u_kernel, dim3(cdiv(cols, 256), cdiv(rows, rpb)), dim3(256), 0, s, w, u, tv, rows, cols, rpb);
        CPCSV_CHECK_LAUNCH();
    }
    const int seg_len = g_cpcsv_deterministic ? cols : SN_SEG;    // deterministic: one wavefront per whole row
    const int segs = cdiv(cols, seg_len);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(cdiv((long)rows * segs, 4)), dim3(256), 0, s, w, iterate ? tv : v, tu, nv2, rows, cols, segs,
                       seg_len, iterate);
    CPCSV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(1024), 0, s, tv, tu, nv2, u, v, out, u_snap, v_snap, rows, cols, eps, iterate);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_sn_multi_blocks(int rows, int cols, int pass) {
    if (pass == 3) return cdiv(cols, SN1_CB);                     // the one-pass form: column slabs
    if (pass == 1) return cdiv(cols, 256) * cdiv(rows, g_cpcsv_deterministic ? rows : 32);
    const int seg_len = g_cpcsv_deterministic ? cols : SN_SEG;
    return cdiv((long)rows * cdiv(cols, seg_len), 4);
}

extern "C" int cpcsv_spectral_sigma_multi1(const cpcsv_sn_job* jobs, int njobs, const int* start, int nblk, float* part,
                                           const long long* part_off, int max_rows, void* stream) {
    if (!jobs || njobs <= 0 || !start || nblk <= 0 || !part || !part_off) return -1001;
    if (max_rows <= 0 || max_rows > 1024) return -1002;          // a thread holds every eighth row of its column in registers
    if (g_cpcsv_deterministic) return -1003;                      // the reproducible mode keeps the two-pass form's summation orders
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sn_multi_onepass_kernel, dim3(nblk), dim3(256), 0, s, jobs, start, njobs, part, part_off);
    CPCSV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_multi_rowsum_kernel, dim3(cdiv(max_rows, 16), njobs), dim3(256), 0, s, jobs, start, njobs, part, part_off);
    CPCSV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_multi_finish_kernel, dim3(njobs), dim3(1024), 0, s, jobs, 1e-12f, 1);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_spectral_sigma_multi(const cpcsv_sn_job* jobs, int njobs, const int* start1, int nblk1, const int* start2,
                                          int nblk2, int iterate, void* stream) {
    if (!jobs || njobs <= 0 || !start2 || nblk2 <= 0 || (iterate && (!start1 || nblk1 <= 0))) return -1001;
    hipStream_t s = (hipStream_t)stream;
    if (iterate) {
        hipLaunchKernelGGL(sn_multi_wt_u_kernel, dim3(nblk1), dim3(256), 0, s, jobs, start1, njobs, g_cpcsv_deterministic ? 0 : 32);
        CPCSV_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(sn_multi_w_v_kernel, dim3(nblk2), dim3(256), 0, s, jobs, start2, njobs, g_cpcsv_deterministic ? 0 : SN_SEG, iterate);
    CPCSV_CHECK_LAUNCH();
    hipLaunchKernelGGL(sn_multi_finish_kernel, dim3(njobs), dim3(1024), 0, s, jobs, 1e-12f, iterate);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_layer_update(const cpcsv_update_desc* d, void* stream) {
    if (!d || !d->G || !d->p || !d->m || !d->v || !d->hyper) return -1001;
    if (d->taps < 1 || d->taps > CPCSV_MAX_TAPS || d->S < 1 || d->S > CPCSV_MAX_TAPS || d->nterms < 0 || d->nterms > 4) return -1002;
    if (d->Cin_s % 8 || d->Cout_s % 8 || d->Cin_s < d->Cin || d->Cout_s < d->Cout) return -1003;
    TapMap fmap = d->sum ? make_map(nullptr, 0, 0) : make_map(d->tapmap, d->S, d->taps);
    if (!d->sum) {                 // tapmap[sl] >= S marks "identity" requests from callers that passed no map
        bool ident = true;
        for (int i = 0; i < d->S; ++i) ident = ident && d->tapmap[i] == (int8_t)i;
        if (ident) fmap = make_map(nullptr, d->S, d->taps);
    }
    const TapMap inv = invert(fmap, d->S, d->taps);
    const MaskTab mk = make_masks(d->sum ? d->masks : nullptr, d->S);
    UpdTerms terms;
    terms.n = d->nterms;
    for (int k = 0; k < 4; ++k) { terms.gw[k] = d->gw[k]; terms.sigma[k] = d->sigma[k]; terms.u[k] = d->u[k]; terms.v[k] = d->v_sn[k]; }
    for (int k = 0; k < d->nterms; ++k) if (!terms.gw[k] || !terms.sigma[k] || !terms.u[k] || !terms.v[k]) return -1004;
    int LT = d->taps + 1;
    if (!(LT & 1)) ++LT;
    hipStream_t s = (hipStream_t)stream;
    static const int upd_probe = [] { const char* e = getenv("CPCSV_UPD_PROBE"); return e ? atoi(e) : 0; }();   // tools only
    auto launch = [&](auto kern, int UO, int UI) {
        int LO = UI * LT;
        if (!(LO & 1)) ++LO;
        const size_t lds = (size_t)UO * LO * sizeof(float);
        dim3 grid(cdiv(d->Cin, UI), cdiv(d->Cout, UO));
        int flags = upd_probe;
        static const int upd_xcd = [] { const char* e = getenv("CPCSV_UPD_XCD"); return e ? atoi(e) : 1; }();   // 0: the (x, y) grid (A/B)
        if (upd_xcd && (d->bwd || d->lin) && grid.y > 1) {       // (only the transposed copies have the 16-byte pieces)
            const long groups = (long)grid.x * cdiv((int)grid.y, 8);
            grid = dim3((unsigned)(cdiv(groups, 8) * 64), 1);
            flags |= 0x10000;
        }
        if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, d->G, d->p, d->m, d->v, d->fwd, d->bwd, d->lin, d->hyper, d->beta1, d->beta2, d->eps,
                           d->Cout, d->Cin, d->taps, d->S, d->Cin_s, d->Cout_s, d->sum, fmap, inv, mk, terms, LT, LO, flags,
                           d->gscale != 0.f ? d->gscale : 1.f, d->step_add, d->g_bf16);
    };
    // tile = 8 output x 32 input channels (all taps): the pass is latency-bound, so the tile is as small as the 16-byte
    // stores of the data-gradient copy allow (8 consecutive output channels) - measured in the step: 32x32 20.48 ms,
    // 16x32 20.07, 8x32 20.03, 16x16 20.05. Single-tap (dense) layers take 128 input channels so that a master run is
    // 512 bytes instead of 128 (round 5, dense tiles with 128-byte runs in the TRANSPOSED copy instead - 64x32 / 32x64 / 64x64:
    // 13.54 / 13.48 / 13.59 against 13.45-13.48 ms per step).
    const bool wide = d->taps == 1 && d->S == 1;
    constexpr int upd_tile = 0;       // (tile sweeps of rounds 4-5: 8 x 32 / 8 x 128 kept; knob retired)
    static const int upd_dense = [] { const char* e = getenv("CPCSV_UPD_DENSE"); return e ? atoi(e) : 1; }();   // 0: the general kernel (A/B)
    if (wide && upd_dense && terms.n == 0 && !upd_probe && !d->bwd && !d->sum && d->tapmap[0] == 0 && (d->Cout_s % 8) == 0 && (d->Cin_s % 8) == 0) {
        const dim3 grid(cdiv(d->Cin, 64), cdiv(d->Cout, 64));
        const float gs = d->gscale != 0.f ? d->gscale : 1.f;
        if (d->dtype == CPCSV_BF16)
            hipLaunchKernelGGL(layer_update_dense_kernel<bf16_t>, grid, dim3(256), 0, s, d->G, d->p, d->m, d->v, (bf16_t*)d->fwd, (bf16_t*)d->lin,
                               d->hyper, d->beta1, d->beta2, d->eps, d->Cout, d->Cin, d->Cin_s, d->Cout_s, gs, d->step_add, d->g_bf16);
        else
            hipLaunchKernelGGL(layer_update_dense_kernel<float>, grid, dim3(256), 0, s, d->G, d->p, d->m, d->v, (float*)d->fwd, (float*)d->lin,
                               d->hyper, d->beta1, d->beta2, d->eps, d->Cout, d->Cin, d->Cin_s, d->Cout_s, gs, d->step_add, d->g_bf16);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    if (d->dtype == CPCSV_BF16) {
        if (wide && upd_tile == 6) launch(layer_update_kernel<bf16_t, 32, 128>, 32, 128);
        else if (wide) launch(layer_update_kernel<bf16_t, 8, 128>, 8, 128);
        else if (upd_tile == 7) launch(layer_update_kernel<bf16_t, 32, 32>, 32, 32);
        else launch(layer_update_kernel<bf16_t, 8, 32>, 8, 32);
    } else {
        if (wide) launch(layer_update_kernel<float, 8, 128>, 8, 128);
        else launch(layer_update_kernel<float, 8, 32>, 8, 32);
    }
    CPCSV_CHECK_LAUNCH();
    return 0;
}
