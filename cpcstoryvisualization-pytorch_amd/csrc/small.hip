// small.hip — the launch-bound small ops of the step: GRU gate math, the per-sample 1-D dynamic
// filter, CA_NET reparametrisation, the scalar losses, multi-tensor Adam. All fp32.
#include "common.h"
#include "../../include/cpcsv_hip.h"

namespace {

inline int grid_for(long n, int block = 256, int cap = 8192) {
    long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

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

// ---- GRUCell (gate order r, z, n) ------------------------------------------
// h / hnew / dhnew / dh rows have stride ldh >= H (the padded width the next dense layer reads: the recurrence then needs no
// re-padding launch per time step); their pad columns are written as zeros.
__global__ void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h,
                               float* __restrict__ hnew, float* __restrict__ gates, int B, int H, int ldg, int ldh) {
    GRID_STRIDE(i, (long)B * ldh) {
        const int b = (int)(i / ldh), k = (int)(i % ldh);
        if (k >= H) { hnew[i] = 0.f; continue; }
        const float* a = gi + (long)b * ldg;
        const float* c = gh + (long)b * ldg;
        const float r = sigm(a[k] + c[k]);
        const float z = sigm(a[H + k] + c[H + k]);
        const float hn = c[2 * H + k];
        const float n = tanhf(a[2 * H + k] + r * hn);
        const float hp = h[i];
        hnew[i] = (1.f - z) * n + z * hp;
        float* g = gates + (long)b * 4 * H;
        g[k] = r; g[H + k] = z; g[2 * H + k] = n; g[3 * H + k] = hn;
    }
}
__global__ void gru_bwd_kernel(const float* __restrict__ dhnew, const float* __restrict__ gates,
                               const float* __restrict__ h, float* __restrict__ dgi, float* __restrict__ dgh,
                               float* __restrict__ dh, int B, int H, int ldg, int ldh) {
    GRID_STRIDE(i, (long)B * ldh) {
        const int b = (int)(i / ldh), k = (int)(i % ldh);
        if (k >= H) { dh[i] = 0.f; continue; }
        const float* g = gates + (long)b * 4 * H;
        const float r = g[k], z = g[H + k], n = g[2 * H + k], hn = g[3 * H + k];
        const float d = dhnew[i];
        const float dn_pre = d * (1.f - z) * (1.f - n * n);
        const float dz_pre = d * (h[i] - n) * z * (1.f - z);
        const float dr_pre = dn_pre * hn * r * (1.f - r);
        float* a = dgi + (long)b * ldg;
        float* c = dgh + (long)b * ldg;
        a[k] = dr_pre; c[k] = dr_pre;
        a[H + k] = dz_pre; c[H + k] = dz_pre;
        a[2 * H + k] = dn_pre; c[2 * H + k] = dn_pre * r;
        if (k == 0)
            for (int q = 3 * H; q < ldg; ++q) { a[q] = 0.f; c[q] = 0.f; }      // row pads of the gate-gradient matrices
        dh[i] = d * z;
    }
}

// ---- DynamicFilterLayer1D (layers.py:69-80) -----------------------------------------------------
__global__ void dfl_fwd_kernel(const float* __restrict__ sig, const float* __restrict__ taps, float* __restrict__ out,
                               int C, int L, int K, int pad) {
    const int n = blockIdx.x;
    extern __shared__ float sh[];      // sig [C][L] then taps [C][K]
    float* ssig = sh; float* stap = sh + C * L;
    for (int i = threadIdx.x; i < C * L; i += blockDim.x) ssig[i] = sig[(long)n * C * L + i];
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) stap[i] = taps[(long)n * C * K + i];
    __syncthreads();
    for (int x = threadIdx.x; x < L; x += blockDim.x) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < K; ++k) {
                const int xi = x + k - pad;
                if (xi >= 0 && xi < L) acc += ssig[c * L + xi] * stap[c * K + k];
            }
        out[(long)n * L + x] = acc;
    }
}
__global__ void dfl_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ sig,
                               const float* __restrict__ taps, float* __restrict__ dsig, float* __restrict__ dtaps,
                               int C, int L, int K, int pad) {
    const int n = blockIdx.x;
    extern __shared__ float sh[];      // dout [L], sig [C][L], taps [C][K]
    float* sd = sh; float* ssig = sh + L; float* stap = ssig + C * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) sd[i] = dout[(long)n * L + i];
    for (int i = threadIdx.x; i < C * L; i += blockDim.x) ssig[i] = sig[(long)n * C * L + i];
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) stap[i] = taps[(long)n * C * K + i];
    __syncthreads();
    for (int i = threadIdx.x; i < C * L; i += blockDim.x) {      // dsig[c][x'] = sum_k dout[x'-k+pad]*taps[c][k]
        const int c = i / L, xp = i % L;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) { const int x = xp - k + pad; if (x >= 0 && x < L) acc += sd[x] * stap[c * K + k]; }
        dsig[(long)n * C * L + i] = acc;
    }
    for (int i = threadIdx.x; i < C * K; i += blockDim.x) {      // dtaps[c][k] = sum_x dout[x]*sig[c][x+k-pad]
        const int c = i / K, k = i % K;
        float acc = 0.f;
        for (int x = 0; x < L; ++x) { const int xi = x + k - pad; if (xi >= 0 && xi < L) acc += sd[x] * ssig[c * L + xi]; }
        dtaps[(long)n * C * K + i] = acc;
    }
}

// ---- CA_NET reparametrisation (model.py:53-60) -----------------------------------------------
__global__ void reparam_fwd_kernel(const float* mu, const float* lv, const float* eps, float* out, long n) {
    GRID_STRIDE(i, n) out[i] = eps[i] * expf(0.5f * lv[i]) + mu[i];
}
__global__ void reparam_bwd_kernel(const float* dout, const float* lv, const float* eps, float* dmu, float* dlv, long n, int acc) {
    GRID_STRIDE(i, n) {
        const float d = dout[i];
        const float g = d * eps[i] * 0.5f * expf(0.5f * lv[i]);
        if (acc) { dmu[i] += d; dlv[i] += g; } else { dmu[i] = d; dlv[i] = g; }
    }
}

// ---- losses --------------------------------------------------------------------------------------
// nn.BCELoss on probabilities: log clamped at -100; backward (p-t)/max(p(1-p),1e-12)  (ATen semantics)
__global__ void bce_kernel(const float* p, const float* t, float* loss, float* grad, long n) {
    __shared__ float sh[16];
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float pi = p[i], ti = t[i];
        const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(logf(1.f - pi), -100.f);
        acc += -(ti * lp + (1.f - ti) * lq);
        grad[i] = (pi - ti) / fmaxf(pi * (1.f - pi), 1e-12f) * inv;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = s * inv;
}
// The three BCE terms of a critic update in one launch (miscc/utils.py:76-101): p = [real | wrong | fake] probabilities,
// groups of n0 / n1 / n2 entries, out[g] = nn.BCELoss (mean) of group g, out[3] = sum_g w[g] * out[g];
// grad[i] = d out[3] / d p[i] = w[g] * (p-t)/max(p(1-p),1e-12) / n_g.
__global__ void bce_groups_kernel(const float* p, const float* t, float* out, float* grad, int n0, int n1, int n2, float w0, float w1,
                                  float w2) {
    __shared__ float sh[16];
    const int n[3] = {n0, n1, n2};
    const float w[3] = {w0, w1, w2};
    float total = 0.f;
    long base = 0;
    for (int g = 0; g < 3; ++g) {
        if (n[g] <= 0) { if (threadIdx.x == 0) out[g] = 0.f; continue; }
        float acc = 0.f;
        const float inv = 1.f / (float)n[g];
        for (long i = threadIdx.x; i < n[g]; i += blockDim.x) {
            const float pi = p[base + i], ti = t[base + i];
            const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(logf(1.f - pi), -100.f);
            acc += -(ti * lp + (1.f - ti) * lq);
            grad[base + i] = w[g] * (pi - ti) / fmaxf(pi * (1.f - pi), 1e-12f) * inv;
        }
        const float s_ = block_sum(acc, sh) * inv;
        if (threadIdx.x == 0) out[g] = s_;
        total += w[g] * s_;
        base += n[g];
    }
    if (threadIdx.x == 0) out[3] = total;
}
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
// nn.MultiLabelSoftMarginLoss: mean_n mean_c -[t logsig(x) + (1-t) logsig(-x)]
__global__ void mlsm_kernel(const float* x, const float* t, float* loss, float* grad, int N, int C, int ld) {
    __shared__ float sh[16];
    float acc = 0.f;
    const float inv = 1.f / ((float)N * (float)C);
    for (long i = threadIdx.x; i < (long)N * ld; i += blockDim.x) {
        const int n = (int)(i / ld), c = (int)(i % ld);
        if (c >= C) { grad[i] = 0.f; continue; }                  // column pads of the logits matrix
        const float xi = x[i], ti = t[(long)n * C + c];
        acc += -(ti * log_sigmoid(xi) + (1.f - ti) * log_sigmoid(-xi));
        grad[i] = (sigm(xi) - ti) * inv;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = s * inv;
}
// KL_loss (miscc/utils.py:184-188): -0.5*mean(1 + lv - mu^2 - exp(lv))
__global__ void kl_kernel(const float* mu, const float* lv, float* loss, float* dmu, float* dlv, long n) {
    __shared__ float sh[16];
    float acc = 0.f;
    const float inv = 1.f / (float)n;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const float m = mu[i], l = lv[i], e = expf(l);
        acc += 1.f + l - m * m - e;
        dmu[i] = m * inv;
        dlv[i] = -0.5f * (1.f - e) * inv;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = -0.5f * s * inv;
}
template <typename T>
__global__ void mse_kernel(const T* a, const T* b, float* loss, T* da, T* db, long n, float inv) {
    __shared__ float sh[4];
    float acc = 0.f;
    GRID_STRIDE(i, n) {
        const float d = elem<T>::ld(a + i) - elem<T>::ld(b + i);
        acc += d * d;
        if (da) elem<T>::st(da + i, 2.f * d * inv);
        if (db) elem<T>::st(db + i, -2.f * d * inv);
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(loss, s * inv);
}

// ---- multi-tensor Adam (torch.optim.Adam defaults, no amsgrad / weight decay) ---------------------
// Step count and learning rate live in DEVICE memory (hyper = {step, lr}), so a captured HIP graph of the whole
// training step replays with the right bias corrections; adam_tick advances the counter once per optimiser step.
constexpr int ADAM_CHUNK = 4096;
__global__ void adam_tick_kernel(float* hyper) { hyper[0] += 1.f; }
__global__ void adam_kernel(void* const* __restrict__ table, const long* __restrict__ sizes,
                            const int* __restrict__ chunk_tensor, const long* __restrict__ chunk_offset,
                            const float* __restrict__ hyper, float beta1, float beta2, float eps) {
    __shared__ float hs[2];
    if (threadIdx.x == 0) {
        const float t = hyper[0], lr = hyper[1];
        const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
        hs[0] = lr / bc1;                 // step size
        hs[1] = 1.f / sqrtf(bc2);         // 1/sqrt(bias_correction2)
    }
    __syncthreads();
    const float step_size = hs[0], inv_bc2_sqrt = hs[1];
    const int ti = chunk_tensor[blockIdx.x];
    const long off = chunk_offset[blockIdx.x];
    float* p = (float*)table[4 * ti + 0];
    const float* g = (const float*)table[4 * ti + 1];
    float* m = (float*)table[4 * ti + 2];
    float* v = (float*)table[4 * ti + 3];
    const long n = sizes[ti];
    const long end = off + ADAM_CHUNK < n ? off + ADAM_CHUNK : n;
    for (long i = off + threadIdx.x; i < end; i += blockDim.x) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
    }
}

}  // namespace

extern "C" int cpcsv_gru_gates_fwd(const float* gi, const float* gh, const float* h, float* hnew, float* gates, int B,
                                   int H, int ldg, int ldh, void* stream) {
    if (ldh < H) return -1001;
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(grid_for((long)B * ldh)), dim3(256), 0, (hipStream_t)stream, gi, gh, h, hnew, gates, B, H, ldg, ldh);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_gru_gates_bwd(const float* dhnew, const float* gates, const float* h, float* dgi, float* dgh,
                                   float* dh, int B, int H, int ldg, int ldh, void* stream) {
    if (ldh < H) return -1001;
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(grid_for((long)B * ldh)), dim3(256), 0, (hipStream_t)stream, dhnew, gates, h, dgi, dgh, dh, B, H, ldg, ldh);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_dfl1d_fwd(const float* sig, const float* taps, float* out, int N, int C, int L, int K, int pad, void* stream) {
    const size_t sh = (size_t)(C * L + C * K) * sizeof(float);
    hipLaunchKernelGGL(dfl_fwd_kernel, dim3(N), dim3(128), sh, (hipStream_t)stream, sig, taps, out, C, L, K, pad);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_dfl1d_bwd(const float* dout, const float* sig, const float* taps, float* dsig, float* dtaps, int N,
                               int C, int L, int K, int pad, void* stream) {
    const size_t sh = (size_t)(L + C * L + C * K) * sizeof(float);
    hipLaunchKernelGGL(dfl_bwd_kernel, dim3(N), dim3(128), sh, (hipStream_t)stream, dout, sig, taps, dsig, dtaps, C, L, K, pad);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* out, long n, void* stream) {
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, out, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_reparam_bwd(const float* dout, const float* logvar, const float* eps, float* dmu, float* dlogvar, long n,
                                 int accumulate, void* stream) {
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dout, logvar, eps, dmu, dlogvar, n, accumulate);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_bce_fwd(const float* p, const float* target, float* loss, float* grad, long n, void* stream) {
    hipLaunchKernelGGL(bce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, target, loss, grad, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_bce_groups(const float* p, const float* target, float* out, float* grad, int n0, int n1, int n2, float w0,
                                float w1, float w2, void* stream) {
    if (!p || !target || !out || !grad || n0 < 0 || n1 < 0 || n2 < 0) return -1001;
    hipLaunchKernelGGL(bce_groups_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, p, target, out, grad, n0, n1, n2, w0, w1, w2);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_mlsm_fwd(const float* logits, const float* target, float* loss, float* grad, int N, int C, int ld, void* stream) {
    hipLaunchKernelGGL(mlsm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, loss, grad, N, C, ld);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_kl_fwd(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, long n, void* stream) {
    hipLaunchKernelGGL(kl_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mu, logvar, loss, dmu, dlogvar, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_mse_fwd(const void* a, const void* b, int dtype, float* loss, void* da, void* db, long n, long count, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), s);
    if (e != hipSuccess) return -(int)e;
    const float inv = 1.f / (float)(count > 0 ? count : n);
    const int grid = g_cpcsv_deterministic ? 1 : grid_for(n, 256, 1024);       // one block: one summation order
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(mse_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, loss, (bf16_t*)da, (bf16_t*)db, n, inv);
    else hipLaunchKernelGGL(mse_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, (const float*)b, loss, (float*)da, (float*)db, n, inv);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_adam_step(void* const* table, const long* sizes, int ntensors, long total_chunks,
                               const int* chunk_tensor, const long* chunk_offset, float* hyper, float beta1, float beta2,
                               float eps, void* stream) {
    if (!table || !sizes || !hyper || ntensors <= 0 || total_chunks <= 0) return -1001;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, hyper);
    CPCSV_CHECK_LAUNCH();
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)total_chunks), dim3(256), 0, s, table, sizes, chunk_tensor, chunk_offset,
                       hyper, beta1, beta2, eps);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_adam_chunk(void) { return ADAM_CHUNK; }
int g_cpcsv_deterministic = 0;
extern "C" int cpcsv_set_deterministic(int on) {
    const int was = g_cpcsv_deterministic;
    g_cpcsv_deterministic = on ? 1 : 0;
    return was;
}
extern "C" int cpcsv_version(void) { return 100; }
extern "C" const char* cpcsv_arch(void) { return "gfx950"; }
