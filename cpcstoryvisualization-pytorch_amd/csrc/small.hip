// small.hip — the launch-bound small ops of the step: GRU gate math, the per-sample 1-D dynamic
// filter, CA_NET reparametrisation, the scalar losses, multi-tensor Adam. All fp32.
#include "common.h"
#include "../../include/cpcsv_hip.h"

// No implicit multiply-add fusion in this file: a*b + c is two roundings unless the source says fmaf. The per-layer kernels of
// small.hip and the stage kernel of text.hip evaluate the SAME expressions (GRU gates, reparametrisation, dynamic filter, bias adds)
// and must agree bit for bit; left to the compiler, which of two products of  (1-z)*n + z*h  is fused depends on the code around it.
#pragma clang fp contract(off)

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
__global__ void mlsm_kernel(const float* x, const float* t, float* loss, float* grad, float* acc_out, int N, int C, int ld) {
    __shared__ float sh[16];
    float acc = 0.f, hit = 0.f, pos = 0.f;
    const float inv = 1.f / ((float)N * (float)C);
    for (long i = threadIdx.x; i < (long)N * ld; i += blockDim.x) {
        const int n = (int)(i / ld), c = (int)(i % ld);
        if (c >= C) { grad[i] = 0.f; continue; }                  // column pads of the logits matrix
        const float xi = x[i], ti = t[(long)n * C + c];
        acc += -(ti * log_sigmoid(xi) + (1.f - ti) * log_sigmoid(-xi));
        grad[i] = (sigm(xi) - ti) * inv;
        // get_multi_acc (reference miscc/utils.py:313-321): labels that are 1 and predicted sigmoid(x) >= .5 <=> x >= 0
        if (ti == 1.f) { pos += 1.f; if (xi >= 0.f) hit += 1.f; }
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) loss[0] = s * inv;
    if (acc_out) {                                                 // uniform branch
        __syncthreads();
        const float h = block_sum(hit, sh);
        __syncthreads();
        const float q = block_sum(pos, sh);
        if (threadIdx.x == 0) acc_out[0] = h / q;                 // divides by zero like the reference when no label is set
    }
}
// out[0] = sum_i w_i * x_i[0] over up to 8 device scalars (the generator's total loss, reference trainer.py:409-413) and
// its backward dx[i] = g[0] * w_i
__global__ void lincomb_kernel(cpcsv_scalar_list l, float* out) {
    float s = 0.f;
    for (int i = 0; i < l.n; ++i) s += l.w[i] * l.x[i][0];
    out[0] = s;
}
__global__ void lincomb_bwd_kernel(const float* g, cpcsv_scalar_list l, float* dx) {
    if ((int)threadIdx.x < l.n) dx[threadIdx.x] = g[0] * l.w[threadIdx.x];
}
// up to 8 device-to-device copies in one launch (the input tensors of a captured graph piece: blockIdx.y = pair)
__global__ void copy_many_kernel(cpcsv_copy_list l) {
    const int k = blockIdx.y;
    const long bytes = l.bytes[k];
    const unsigned char* __restrict__ src = reinterpret_cast<const unsigned char*>(l.src[k]);
    unsigned char* __restrict__ dst = reinterpret_cast<unsigned char*>(l.dst[k]);
    const bool al = ((reinterpret_cast<unsigned long long>(src) | reinterpret_cast<unsigned long long>(dst)) & 15) == 0;
    const long n16 = al ? bytes >> 4 : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
    for (long i = (n16 << 4) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
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

// ---- dense layer over a handful of rows (the text / motion encoders, GRU recurrences: 12-60 rows, K and N in the hundreds) ----
//   y[m][n] = act(alpha * sum_k x[m][k] * w[n][k] + bias[n]),  exact fp32 FMA chains, w row-major [N][ldw] (k contiguous).
// The MFMA gather-GEMM needs two launches for these (split-K + slab pass: one block per 128 columns would walk all K tiles
// behind a global->LDS round trip each) and 20-25 us of the chains at the head of the forward and the tail of the backward pass.
// Here a block owns 4 output columns x 16 rows and its 4 wavefronts a quarter of K each: lane = 16 k-lanes x 4 row groups of 4 rows,
// walking its K slice in steps of 64 floats with two steps (16 loads of 16 bytes) in flight, a 16-lane butterfly, then LDS.
// ceil(N/4) * ceil(M/16) blocks, one launch.
// stats (optional): BatchNorm partials [ceil(M/16)][2][ldstat] of the pre-activation values, as cpcsv_gemm_nt emits them.
__global__ __launch_bounds__(256) void dense_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw,
                                                         float* __restrict__ y, int ldy, int M, int N, int K, const float* alpha_p,
                                                         const float* __restrict__ bias, int act, float* stats, int ldstat,
                                                         const float* __restrict__ init, int ldi, int accumulate) {
    // block = one (4-column group, 16-row block); its 4 wavefronts take a quarter of K each (every wavefront then has its loads in
    // flight at once for K <= 512: one memory round trip), partial sums meet in LDS
    __shared__ float part[4][64][4];                                  // [k slice][row group * 16 + 4 rows... see below][column]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncg = (ldy + 3) / 4;
    const int cg = blockIdx.x % ncg, rb = blockIdx.x / ncg;
    const int m0 = rb * 16, n0 = cg * 4;
    const int kq = lane & 15, rg = lane >> 4;
    const float* wr[4];
    const float* xr[4];
    bool wok[4], xok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { wok[c] = n0 + c < N; wr[c] = w + (long)(wok[c] ? n0 + c : 0) * ldw; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int m = m0 + rg * 4 + r; xok[r] = m < M; xr[r] = x + (long)(xok[r] ? m : 0) * ldx; }
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
    const int kslice = ((K + 255) / 256) * 64;                       // per wavefront, a multiple of 64
    const int kbeg = wave * kslice, kend = kbeg + kslice < K ? kbeg + kslice : K;
    auto step = [&](int k) {                                          // k = this lane's 4 consecutive k of one 64-float step
        f32x4 wv[4], xv[4];
        const bool in = k < kend;                                    // K is a multiple of 4 (stored widths are multiples of 8)
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = in ? *reinterpret_cast<const f32x4*>(wr[c] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = in ? *reinterpret_cast<const f32x4*>(xr[r] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][c] = fmaf(xv[r][e], wv[c][e], acc[r][c]);
    };
    for (int k0 = kbeg; k0 < kend; k0 += 128) {
        step(k0 + kq * 4);
        step(k0 + 64 + kq * 4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[r][c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            acc[r][c] = v;
        }
    if (kq == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[wave][rg * 4 + r][c] = acc[r][c];
    __syncthreads();
    if (wave != 0) return;
    // wavefront 0: lane = (row of the block) * 4 + column
    const int row = lane >> 2, col = lane & 3;
    const int m = m0 + row, n = n0 + col;
    const bool live = m < M && n < N;
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const float sum = (part[0][row][col] + part[1][row][col]) + (part[2][row][col] + part[3][row][col]);
    const float t = live ? sum * alpha + (bias ? bias[n] : 0.f) : 0.f;
    // (init: added to the result - the other gradient contribution of a recurrence; its pad columns pass through)
    if (m < M && n < ldy) {
        float* dst = y + (long)m * ldy + n;
        *dst = (live ? act_apply(t, act) : 0.f) + (init ? init[(long)m * ldi + n] : 0.f) + (accumulate ? *dst : 0.f);
    }
    if (stats) {
        float cs = t, cq = t * t;                                     // column sums over the 16 rows: lanes col, col+4, ...
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { cs += __shfl_xor(cs, o); cq += __shfl_xor(cq, o); }
        if (lane < 4 && n < N) {
            stats[((long)rb * 2 + 0) * ldstat + n] = cs;
            stats[((long)rb * 2 + 1) * ldstat + n] = cq;
        }
    }
}

extern "C" int cpcsv_dense_rows(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int M, int N, int K,
                                const float* alpha, const float* bias, int act, float* stats, int ldstat, const float* init, int ldi,
                                int accumulate, void* stream) {
    if (!x || !w || !y || M <= 0 || M > 64 || N <= 0 || K <= 0 || (K & 3) || (ldx & 3) || (ldw & 3) || ldy < N || (ldy & 3)) return -1001;
    if (ldx < K || ldw < K) return -1001;          // the 16-byte loads walk k < K on every x and w row
    if (init && ldi < ldy) return -1002;
    const long blocks = (long)((ldy + 3) / 4) * ((M + 15) / 16);
    hipLaunchKernelGGL(dense_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, w, ldw, y, ldy, M, N, K,
                       alpha, bias, act, stats, ldstat, init, ldi, accumulate);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

// One GRUCell recurrence step in ONE launch (model.py:223-224,331,342: h' = GRU(gi, h) with gi = W_ih x + b_ih precomputed for all
// steps): the three W_hh products of hidden unit j for 16 rows (a cpcsv_dense_rows block whose "columns" are rows j, H+j, 2H+j of
// W_hh), then the gate math of cpcsv_gru_gates_fwd on them. gates [B][4H] = r, z, n, W_hn h + b_hn (what the backward reads).
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ gi, int ldg, const float* __restrict__ h, int ldh,
                                                           const float* __restrict__ w, int ldw, const float* __restrict__ bhh,
                                                           float* __restrict__ hnew, float* __restrict__ gates, int B, int H) {
    __shared__ float part[4][16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x % ldh, rb = blockIdx.x / ldh;
    const int m0 = rb * 16;
    if (j >= H) {                                                     // pad columns of the state stay zero
        if (threadIdx.x < 16 && m0 + (int)threadIdx.x < B) hnew[(long)(m0 + threadIdx.x) * ldh + j] = 0.f;
        return;
    }
    const int kq = lane & 15, rg = lane >> 4;
    const float* wr[3];
    const float* xr[4];
    bool xok[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) wr[c] = w + (long)(c * H + j) * ldw;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int m = m0 + rg * 4 + r; xok[r] = m < B; xr[r] = h + (long)(xok[r] ? m : 0) * ldh; }
    float acc[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[r][c] = 0.f;
    const int K = ldh;
    const int kslice = ((K + 255) / 256) * 64;
    const int kbeg = wave * kslice, kend = kbeg + kslice < K ? kbeg + kslice : K;
    auto step = [&](int k) {
        f32x4 wv[3], xv[4];
        const bool in = k < kend;
#pragma unroll
        for (int c = 0; c < 3; ++c) wv[c] = in ? *reinterpret_cast<const f32x4*>(wr[c] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[r] = (in && xok[r]) ? *reinterpret_cast<const f32x4*>(xr[r] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[r][c] = fmaf(xv[r][e], wv[c][e], acc[r][c]);
    };
    for (int k0 = kbeg; k0 < kend; k0 += 128) {
        step(k0 + kq * 4);
        step(k0 + 64 + kq * 4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc[r][c];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (kq == 0) part[wave][rg * 4 + r][c] = v;
        }
    __syncthreads();
    if (threadIdx.x >= 16) return;
    const int m = m0 + threadIdx.x;
    if (m >= B) return;
    float sgh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        sgh[c] = (part[0][threadIdx.x][c] + part[1][threadIdx.x][c]) + (part[2][threadIdx.x][c] + part[3][threadIdx.x][c]) + bhh[c * H + j];
    const float* a = gi + (long)m * ldg;
    const float r = sigm(a[j] + sgh[0]);
    const float z = sigm(a[H + j] + sgh[1]);
    const float hn = sgh[2];
    const float n = tanhf(a[2 * H + j] + r * hn);
    const float hp = h[(long)m * ldh + j];
    hnew[(long)m * ldh + j] = (1.f - z) * n + z * hp;
    float* g = gates + (long)m * 4 * H;
    g[j] = r; g[H + j] = z; g[2 * H + j] = n; g[3 * H + j] = hn;
}

extern "C" int cpcsv_gru_step_fwd(const float* gi, int ldg, const float* h, int ldh, const float* w_hh, int ldw, const float* b_hh,
                                  float* hnew, float* gates, int B, int H, void* stream) {
    if (!gi || !h || !w_hh || !b_hh || !hnew || !gates || B <= 0 || H <= 0 || ldh < H || (ldh & 3) || (ldw & 3) || ldw < ldh || ldg < 3 * H) return -1001;
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3((unsigned)(ldh * ((B + 15) / 16))), dim3(256), 0, (hipStream_t)stream, gi, ldg, h, ldh, w_hh, ldw,
                       b_hh, hnew, gates, B, H);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

// weight gradient of the same layers, straight into the master-layout gradient:  dW[n][k] += sum_m dz[m][n] * x[m][k]  (n < N, k < Kr;
// dW row-major [N][Kr], the real input width). M <= 64 rows: the product is a stream of N*Kr outputs with a 12-60 term sum each.
// Block = 16 output rows n x 256 columns k: a thread owns one k and 16 n, the dz values of a row are wave-uniform loads.
// Calls that add to the same dW are serialised by their stream (the weight-gradient branch / the backward's own stream).
__global__ __launch_bounds__(256) void dense_rows_wgrad_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ x, int ldx,
                                                               float* __restrict__ dW, float* __restrict__ db, int M, int N, int Kr) {
    __shared__ float sdz[64][16];                                     // the block's 16 dz columns of all rows
    const int k = blockIdx.x * 256 + threadIdx.x, n0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < M * 16; i += 256) {
        const int m = i >> 4, j = i & 15;
        sdz[m][j] = n0 + j < N ? dz[(long)m * ldz + n0 + j] : 0.f;
    }
    const bool kok = k < Kr;
    // the existing dW values and the x column of this thread: all loads in flight before the first use (a row per trip behind its
    // own round trip made this launch 20 us for 12 rows)
    float old[16], acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { old[j] = (kok && n0 + j < N) ? dW[(long)(n0 + j) * Kr + k] : 0.f; acc[j] = 0.f; }
    __syncthreads();
    if (db && blockIdx.x == 0 && threadIdx.x < 16 && n0 + (int)threadIdx.x < N) {      // bias gradient: column sums of this block's dz tile
        float t = 0.f;
        for (int m = 0; m < M; ++m) t += sdz[m][threadIdx.x];
        db[n0 + threadIdx.x] += t;
    }
    for (int m0 = 0; m0 < M; m0 += 16) {
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) xv[r] = (kok && m0 + r < M) ? x[(long)(m0 + r) * ldx + k] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (m0 + r >= M) break;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = fmaf(sdz[m0 + r][j], xv[r], acc[j]);
        }
    }
    if (!kok) return;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (n0 + j < N) dW[(long)(n0 + j) * Kr + k] = old[j] + acc[j];
}

// The same for MANY small layers in one launch (cpcsv_dense_rows_wgrad_multi): the weight gradients of the text / motion encoders and
// GRU cells feed only the optimiser, so the generator's backward parks them (one "piece" per call: dz, x, rows) and issues ONE launch
// at its end instead of ~18 on its critical chain. A block owns 256 input columns x 16 output rows of ONE weight and walks all pieces
// of that weight in the order they were parked: no atomics, one fixed summation order.
__global__ __launch_bounds__(256) void dense_rows_wgrad_multi_kernel(const cpcsv_small_wgrad_list l) {
    __shared__ float sdz[64][16];
    int t = 0;
    for (int i = 1; i < l.ntargets; ++i) t += (int)blockIdx.x >= l.t[i].block0;
    const cpcsv_wgrad_target tg = l.t[t];
    const int b = blockIdx.x - tg.block0;
    const int bxi = b % tg.bx, byi = b / tg.bx;
    const int k = bxi * 256 + threadIdx.x, n0 = byi * 16;
    const bool kok = k < tg.Kr;
    float old[16], acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { old[j] = (kok && n0 + j < tg.N) ? tg.dW[(long)(n0 + j) * tg.Kr + k] : 0.f; acc[j] = 0.f; }
    const bool bcol = tg.db && bxi == 0 && threadIdx.x < 16 && n0 + (int)threadIdx.x < tg.N;
    float bval = bcol ? tg.db[n0 + threadIdx.x] : 0.f;
    for (int pi = tg.piece0; pi < tg.piece0 + tg.npieces; ++pi) {
        const cpcsv_wgrad_piece pc = l.p[pi];
        __syncthreads();                                   // the previous piece's tile has been consumed
        for (int i = threadIdx.x; i < pc.M * 16; i += 256) {
            const int m = i >> 4, j = i & 15;
            sdz[m][j] = n0 + j < tg.N ? pc.dz[(long)m * pc.ldz + n0 + j] : 0.f;
        }
        __syncthreads();
        if (bcol) {
            float t_ = 0.f;
            for (int m = 0; m < pc.M; ++m) t_ += sdz[m][threadIdx.x];
            bval += t_;
        }
        for (int m0 = 0; m0 < pc.M; m0 += 16) {
            float xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xv[r] = (kok && m0 + r < pc.M) ? pc.x[(long)(m0 + r) * pc.ldx + k] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (m0 + r >= pc.M) break;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = fmaf(sdz[m0 + r][j], xv[r], acc[j]);
            }
        }
        // piece by piece like the one-launch-per-piece kernel: (old + acc_1) + acc_2, the same bits
#pragma unroll
        for (int j = 0; j < 16; ++j) { old[j] += acc[j]; acc[j] = 0.f; }
    }
    if (bcol) tg.db[n0 + threadIdx.x] = bval;
    if (!kok) return;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (n0 + j < tg.N) tg.dW[(long)(n0 + j) * tg.Kr + k] = old[j];
}

extern "C" int cpcsv_dense_rows_wgrad_multi(cpcsv_small_wgrad_list* l, void* stream) {
    if (!l || l->ntargets < 1 || l->ntargets > CPCSV_SMALL_WG_TARGETS || l->npieces < 1 || l->npieces > CPCSV_SMALL_WG_PIECES) return -1001;
    int blocks = 0;
    for (int i = 0; i < l->ntargets; ++i) {
        cpcsv_wgrad_target& t = l->t[i];
        if (!t.dW || t.N <= 0 || t.Kr <= 0 || t.npieces < 1 || t.piece0 < 0 || t.piece0 + t.npieces > l->npieces) return -1001;
        for (int pi = t.piece0; pi < t.piece0 + t.npieces; ++pi) {
            const cpcsv_wgrad_piece& pc = l->p[pi];
            if (!pc.dz || !pc.x || pc.M <= 0 || pc.M > 64 || pc.ldz < t.N || pc.ldx < t.Kr) return -1001;
        }
        t.bx = (t.Kr + 255) / 256;                             // (filled in here: block map of the launch)
        t.block0 = blocks;
        blocks += t.bx * ((t.N + 15) / 16);
    }
    hipLaunchKernelGGL(dense_rows_wgrad_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *l);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_dense_rows_wgrad(const float* dz, int ldz, const float* x, int ldx, float* dW, float* db, int M, int N, int Kr, void* stream) {
    if (!dz || !x || !dW || M <= 0 || M > 64 || N <= 0 || Kr <= 0 || ldz < N || ldx < Kr) return -1001;
    hipLaunchKernelGGL(dense_rows_wgrad_kernel, dim3((Kr + 255) / 256, (N + 15) / 16), dim3(256), 0, (hipStream_t)stream, dz, ldz, x, ldx, dW, db, M, N, Kr);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

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
extern "C" int cpcsv_mlsm_fwd(const float* logits, const float* target, float* loss, float* grad, float* acc, int N, int C, int ld,
                              void* stream) {
    hipLaunchKernelGGL(mlsm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, loss, grad, acc, N, C, ld);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_lincomb_fwd(const cpcsv_scalar_list* l, float* out, void* stream) {
    if (!l || !out || l->n < 1 || l->n > 8) return -1001;
    for (int i = 0; i < l->n; ++i) if (!l->x[i]) return -1001;
    hipLaunchKernelGGL(lincomb_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, *l, out);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_lincomb_bwd(const float* g, const cpcsv_scalar_list* l, float* dx, void* stream) {
    if (!g || !l || !dx || l->n < 1 || l->n > 8) return -1001;
    hipLaunchKernelGGL(lincomb_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, *l, dx);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
// ---- the small fp32 dense layers' operand copies, all layers in one launch. Block = one 32 x 32 tile (output channels x input
// channels) of one job: read once (coalesced along the input channel), written as it is into fwd and transposed through LDS into lin.
__global__ __launch_bounds__(256) void pack_dense_many_kernel(const cpcsv_pack_list l) {
    __shared__ float tile[32][33];
    int jb = 0;
    for (int k = 1; k < l.n; ++k)
        if ((int)blockIdx.x >= l.j[k].blk0) jb = k;
    const cpcsv_pack_job j = l.j[jb];
    const int tiles_i = (j.cin_s + 31) / 32;
    const int t = blockIdx.x - j.blk0;
    const int o0 = (t / tiles_i) * 32, i0 = (t % tiles_i) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads, four rows each
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + ty + 8 * r, i = i0 + tx;
        const float v = (o < j.cout && i < j.cin) ? j.w[(long)o * j.cin + i] : 0.f;
        tile[ty + 8 * r][tx] = v;
        if (j.fwd && o < j.cout && i < j.cin_s) j.fwd[(long)o * j.cin_s + i] = v;
    }
    if (!j.lin) return;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, o = o0 + tx;
        if (i < j.cin_s && o < j.cout_s) j.lin[(long)i * j.cout_s + o] = tile[tx][ty + 8 * r];
    }
}
extern "C" int cpcsv_pack_dense_many(cpcsv_pack_list* l, void* stream) {
    if (!l || l->n < 1 || l->n > CPCSV_PACK_JOBS) return -1001;
    int blocks = 0;
    for (int k = 0; k < l->n; ++k) {
        cpcsv_pack_job& j = l->j[k];
        if (!j.w || (!j.fwd && !j.lin) || j.cout < 1 || j.cin < 1 || j.cin_s % 8 || j.cout_s % 8 || j.cin_s < j.cin || j.cout_s < j.cout) return -1001;
        j.blk0 = blocks;
        blocks += ((j.cout_s + 31) / 32) * ((j.cin_s + 31) / 32);
    }
    hipLaunchKernelGGL(pack_dense_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *l);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_copy_many(const cpcsv_copy_list* l, void* stream) {
    if (!l || l->n < 1 || l->n > 8) return -1001;
    long most = 0;
    for (int i = 0; i < l->n; ++i) {
        if (!l->dst[i] || !l->src[i] || l->bytes[i] < 0) return -1001;
        most = l->bytes[i] > most ? l->bytes[i] : most;
    }
    const long per_block = 256L * 16 * 4;                       // 4 16-byte pieces per thread
    long bx = (most + per_block - 1) / per_block;
    bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)bx, (unsigned)l->n), dim3(256), 0, (hipStream_t)stream, *l);
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
