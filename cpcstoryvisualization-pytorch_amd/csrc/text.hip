// text.hip — the generator's text / motion encoders as STAGE launches (include/cpcsv_hip.h: cpcsv_text_stage).
//
// CA_NET, m_net, c_net, both GRU recurrences, image_net, filter_net and DynamicFilterLayer1D (reference model.py:37-65,302-346,
// 371-378; layers.py:69-80) are ~2 M weights over 12-60 rows: launch latency, not work. One launch = one STAGE = up to 8 independent
// jobs side by side; block -> (job, tile). Every product job uses ONE tile engine: a block of 4 wavefronts owns up to 4 output
// columns for ALL rows of the call - 4 slices of K per chunk of 16 rows, lanes = 16 k-lanes x 4 row
// groups of 4 rows, 16-byte loads, two 64-float steps in flight, a 16-lane butterfly, the K slices meeting in LDS - and because it
// owns whole columns it finishes BatchNorm1d (batch statistics from 16-row (sum, sum of squares) partials added in double, running statistics), the GRU gate math or their
// backward forms in its epilogue. fp32 FMA chains in one fixed order; no atomics.
#include "common.h"
#include "../../include/cpcsv_hip.h"

// No implicit multiply-add fusion in this file: a*b + c is two roundings unless the source says fmaf. The per-layer kernels of
// small.hip and the stage kernel of text.hip evaluate the SAME expressions (GRU gates, reparametrisation, dynamic filter, bias adds)
// and must agree bit for bit; left to the compiler, which of two products of  (1-z)*n + z*h  is fused depends on the code around it.
#pragma clang fp contract(off)

namespace {

// threads per block: 4 wavefronts. The engine needs ~160 registers (two 64-float steps of 8 operand vectors in flight): 3 wavefronts per
// SIMD, i.e. THREE such blocks per CU - 8-wavefront blocks (32-row chunks) fit once, and a stage of 370-680 blocks then took two or
// three rounds of a latency chain each (39 us per stage on average; round 5 trace)
constexpr int TT = 256;
constexpr int CH = 16;                   // rows per chunk of the tile engine

__device__ __forceinline__ float sigm_(float x) { return 1.f / (1.f + expf(-x)); }

struct Smem {
    float part[4][CH][4];                // K-slice partials of one chunk of rows
    float val[CPCSV_TXT_MAX_ROWS][4];    // the tile: [row][column of the block]
    float stat[4][2];                    // per column: scale / shift (forward), s1 / s2 (backward)
    float buf[1024];                     // pointwise jobs: signal / taps / gradient rows
};

__device__ __forceinline__ float ld_any(const void* p, long i, int bf16) {
    return bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_any(void* p, long i, int bf16, float v) {
    if (bf16) reinterpret_cast<bf16_t*>(p)[i] = f32_to_bf16(v);
    else reinterpret_cast<float*>(p)[i] = v;
}

// s.val[m][c] = sum_k x[m][k] * w[col[c]][k]   for m < M, c < 4 (columns with cok[c] == false give 0)
__device__ __forceinline__ void tile_product(Smem& s, const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw,
                                             const int (&col)[4], const bool (&cok)[4], int M, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rb = 0, ks = wave & 3;
    const int kq = lane & 15, rg = lane >> 4;
    const int kslice = ((K + 255) / 256) * 64;                       // per K slice, a multiple of 64
    const int kbeg = ks * kslice, kend = kbeg + kslice < K ? kbeg + kslice : K;
    const float* wr[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) wr[c] = w + (long)(cok[c] ? col[c] : 0) * ldw;
    for (int m0 = 0; m0 < M; m0 += CH) {
        const float* xr[4];
        bool xok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + rb * 16 + rg * 4 + r;
            xok[r] = m < M;
            xr[r] = x + (long)(xok[r] ? m : 0) * ldx;
        }
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
        auto step = [&](int k) {                                       // this lane's 4 consecutive k of one 64-float step
            f32x4 wv[4], xv[4];
            const bool in = k < kend;
#pragma unroll
            for (int c = 0; c < 4; ++c) wv[c] = (in && cok[c]) ? *reinterpret_cast<const f32x4*>(wr[c] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = (in && xok[r]) ? *reinterpret_cast<const f32x4*>(xr[r] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[r][c] = fmaf(xv[r][e], wv[c][e], acc[r][c]);
        };
        if (m0 + rb * 16 < M)                                          // (wave-uniform: this row block has live rows)
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
                if (kq == 0) s.part[ks][rb * 16 + rg * 4 + r][c] = v;
            }
        __syncthreads();
        if (tid < CH * 4) {
            const int row = tid >> 2, c = tid & 3;
            if (m0 + row < M) s.val[m0 + row][c] = (s.part[0][row][c] + s.part[1][row][c]) + (s.part[2][row][c] + s.part[3][row][c]);
        }
        __syncthreads();
    }
}

// sum over the rows of column c of s.val-like data, one wavefront per column: f(m) summed over m < M, every lane gets the total
template <typename F>
__device__ __forceinline__ float wave_rows_sum(int M, F&& f) {
    const int lane = threadIdx.x & 63;
    float v = 0.f;
    for (int m = lane; m < M; m += 64) v += f(m);
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward jobs
// ---------------------------------------------------------------------------------------------------------------------
__device__ void job_dense(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int col[4];
    bool cok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { col[c] = tile * 4 + c; cok[c] = col[c] < J.N; }
    const float* gamma = reinterpret_cast<const float*>(J.Q[0]);
    const float* beta = reinterpret_cast<const float*>(J.Q[1]);
    float* rmean = reinterpret_cast<float*>(J.Q[2]);
    float* rvar = reinterpret_cast<float*>(J.Q[3]);
    for (int p = 0; p < J.npass; ++p) {
        const int M = J.M[p];
        if (M <= 0) continue;
        tile_product(s, J.x[p], J.ldx, J.w, J.ldw, col, cok, M, J.K);
        float* y = reinterpret_cast<float*>(J.y[p]);
        float* lin = reinterpret_cast<float*>(J.P[p][0]);
        float* save = reinterpret_cast<float*>(J.P[p][1]);
        for (int i = tid; i < M * 4; i += TT) {                        // + bias; the pre-BatchNorm value is what backward re-normalises
            const int m = i >> 2, c = i & 3;
            const float t = cok[c] ? s.val[m][c] + (J.bias ? J.bias[col[c]] : 0.f) : 0.f;
            s.val[m][c] = t;
            if (lin && col[c] < J.ldy) lin[(long)m * J.ldy + col[c]] = t;
        }
        __syncthreads();
        if (gamma) {
            if (wave < 4) {
                // one wavefront per column. Batch statistics exactly as the per-layer pair computes them (cpcsv_dense_rows emits
                // sum / sum of squares per block of 16 rows - a butterfly over the row index - and cpcsv_bn_apply_partials adds the
                // blocks in double, var = E[x^2] - E[x]^2): the fused path's forward is bit-identical to the path it replaces
                const int c = wave;
                double s1 = 0.0, s2 = 0.0;
                for (int r0 = 0; r0 < M; r0 += 16) {
                    const float t = (lane < 16 && r0 + lane < M) ? s.val[r0 + lane][c] : 0.f;
                    float cs = t, cq = t * t;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) { cs += __shfl_xor(cs, o); cq += __shfl_xor(cq, o); }
                    s1 += (double)cs;
                    s2 += (double)cq;
                }
                if (lane == 0) {
                    float scale = 0.f, shift = 0.f;
                    if (cok[c]) {
                        const double cnt = (double)M;
                        const double mu = s1 / cnt;
                        double var = s2 / cnt - mu * mu;
                        if (var < 0.0) var = 0.0;
                        const float invstd = (float)(1.0 / sqrt(var + (double)J.eps));
                        const float mean = (float)mu;
                        bn_affine(gamma[col[c]], beta[col[c]], mean, invstd, scale, shift);
                        if (save) { save[col[c]] = mean; save[J.ldy + col[c]] = invstd; }
                        if (rmean) {                                   // momentum update with the UNBIASED variance (nn.BatchNorm1d)
                            const double unbias = cnt > 1.0 ? cnt / (cnt - 1.0) : 1.0;
                            rmean[col[c]] = bn_running(rmean[col[c]], mean, J.momentum);
                            rvar[col[c]] = bn_running(rvar[col[c]], (float)(var * unbias), J.momentum);
                        }
                    }
                    s.stat[c][0] = scale;
                    s.stat[c][1] = shift;
                }
            }
            __syncthreads();
        }
        for (int i = tid; i < M * 4; i += TT) {
            const int m = i >> 2, c = i & 3;
            if (col[c] >= J.ldy) continue;
            float t = s.val[m][c];
            if (gamma) t = bn_pre(t, s.stat[c][0], s.stat[c][1]);
            y[(long)m * J.ldy + col[c]] = cok[c] ? act_apply(t, J.act) : 0.f;
        }
        __syncthreads();
    }
}

__device__ void job_ca(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int C = J.A[0], j = tile, ldc = J.A[1];
    for (int p = 0; p < J.npass; ++p) {
        const int M = J.M[p];
        if (M <= 0) continue;
        float* code = reinterpret_cast<float*>(J.P[p][3]);
        if (j >= C) {                                                  // pad columns of the code matrix
            if (tid < M) code[(long)tid * ldc + j] = 0.f;
            continue;
        }
        const int col[4] = {j, C + j, 0, 0};
        const bool cok[4] = {true, true, false, false};
        tile_product(s, J.x[p], J.ldx, J.w, J.ldw, col, cok, M, J.K);
        float* y = reinterpret_cast<float*>(J.y[p]);
        float* mu_o = reinterpret_cast<float*>(J.P[p][0]);
        float* lv_o = reinterpret_cast<float*>(J.P[p][1]);
        const float* eps = reinterpret_cast<const float*>(J.P[p][2]);
        for (int m = tid; m < M; m += TT) {
            const float mu = fmaxf(s.val[m][0] + J.bias[j], 0.f);       // the ReLU precedes the mu / logvar split (model.py:50-52)
            const float lv = fmaxf(s.val[m][1] + J.bias[C + j], 0.f);
            y[(long)m * 2 * C + j] = mu;
            y[(long)m * 2 * C + C + j] = lv;
            mu_o[(long)m * C + j] = mu;
            lv_o[(long)m * C + j] = lv;
            code[(long)m * ldc + j] = eps ? eps[(long)m * C + j] * expf(0.5f * lv) + mu : mu;
        }
        __syncthreads();
    }
}

__device__ void job_gru_fwd(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int H = J.A[0], ldg = J.A[1], j = tile;
    for (int p = 0; p < J.npass; ++p) {
        const int M = J.M[p];
        if (M <= 0) continue;
        float* hnew = reinterpret_cast<float*>(J.y[p]);
        float* sm = reinterpret_cast<float*>(J.P[p][2]);                // story-major copy of the new state (or NULL)
        if (j >= H) {                                                  // pad columns of the state stay zero
            if (tid < M) {
                hnew[(long)tid * J.ldy + j] = 0.f;
                if (sm) sm[((long)tid * J.T[p] + J.A[2]) * J.ldy + j] = 0.f;
            }
            continue;
        }
        const int col[4] = {j, H + j, 2 * H + j, 0};
        const bool cok[4] = {true, true, true, false};
        tile_product(s, J.x[p], J.ldx, J.w, J.ldw, col, cok, M, J.K);
        const float* gi = reinterpret_cast<const float*>(J.P[p][0]);
        float* gates = reinterpret_cast<float*>(J.P[p][1]);
        for (int m = tid; m < M; m += TT) {
            const float* a = gi + (long)m * ldg;
            const float sg0 = s.val[m][0] + J.bias[j], sg1 = s.val[m][1] + J.bias[H + j];      // (W_hh h + b_hh first: cpcsv_gru_step_fwd's order)
            const float r = sigm_(a[j] + sg0);
            const float z = sigm_(a[H + j] + sg1);
            const float hn = s.val[m][2] + J.bias[2 * H + j];
            const float n = tanhf(a[2 * H + j] + r * hn);
            const float hp = J.x[p][(long)m * J.ldx + j];
            const float hv = (1.f - z) * n + z * hp;
            hnew[(long)m * J.ldy + j] = hv;
            if (sm) sm[((long)m * J.T[p] + J.A[2]) * J.ldy + j] = hv;
            float* g = gates + (long)m * 4 * H;
            g[j] = r; g[H + j] = z; g[2 * H + j] = n; g[3 * H + j] = hn;
        }
        __syncthreads();
    }
}

__device__ void job_prep(const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int md = J.A[0], nz = J.A[1], ldm = J.A[2], lde = J.A[3];
    for (int p = 0; p < J.npass; ++p) {
        const int B = J.M[p];
        if (B <= 0) continue;
        const float* motion = reinterpret_cast<const float*>(J.P[p][0]);
        const float* znoise = reinterpret_cast<const float*>(J.P[p][1]);
        const float* n0 = reinterpret_cast<const float*>(J.P[p][2]);
        float* mpad = reinterpret_cast<float*>(J.P[p][3]);
        float* e = reinterpret_cast<float*>(J.P[p][4]);
        float* n0pad = reinterpret_cast<float*>(J.P[p][5]);
        float* tpad = reinterpret_cast<float*>(J.y[p]);
        const int Tp = J.T[p];
        const int rows = Tp * B;
        if (tile < rows) {
            const int t = tile / B, b = tile - t * B;
            const float* src = motion + ((long)b * Tp + t) * md;
            for (int c = tid; c < ldm; c += TT) {
                const float v = c < md ? src[c] : 0.f;
                mpad[(long)tile * ldm + c] = v;
                if (tpad) tpad[((long)b * Tp + t) * ldm + c] = v;
            }
            for (int c = tid; c < lde; c += TT)
                e[(long)tile * lde + c] = c < nz ? znoise[(long)tile * nz + c] : (c < nz + md ? src[c - nz] : 0.f);
        } else if (tile < rows + B) {
            const int b = tile - rows;
            for (int c = tid; c < ldm; c += TT) n0pad[(long)b * ldm + c] = c < md ? n0[(long)b * md + c] : 0.f;
        }
    }
}

__device__ void job_joint(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int md = J.A[0], C = J.A[1], L = J.A[2], KF = J.A[3], nch = J.A[4], ldh = J.A[5], obf = J.A[6];
    const int pad = KF / 2;
    for (int p = 0; p < J.npass; ++p) {
        const int B = J.M[p];
        const int T = J.T[p];
        if (B <= 0 || tile >= B * T) continue;
        const int r = tile, b = r / T, t = r - b * T;
        const float* hall = reinterpret_cast<const float*>(J.P[p][0]);
        const float* mu = reinterpret_cast<const float*>(J.P[p][1]);
        const float* sig = reinterpret_cast<const float*>(J.P[p][2]) + (long)r * J.ldx;
        const float* taps = reinterpret_cast<const float*>(J.P[p][3]) + (long)r * J.ldw;
        float* ssig = s.buf;
        float* stap = s.buf + nch * L;
        for (int i = tid; i < nch * L; i += TT) ssig[i] = sig[i];
        for (int i = tid; i < nch * KF; i += TT) stap[i] = taps[i];
        __syncthreads();
        const long o0 = (long)r * J.ldy;
        const float* hrow = hall + ((long)(t + 1) * B + b) * ldh;
        const float* murow = mu + (long)(r % B) * C;                   // story call: c_mu = r_mu.repeat(T, 1), TILED rows (model.py:361)
        for (int c = tid; c < J.ldy; c += TT) {
            float v = 0.f;
            if (c < md) v = hrow[c];
            else if (c < md + C) v = murow[c - md];
            else if (c < md + C + L) {
                const int xq = c - md - C;
                for (int ch = 0; ch < nch; ++ch)
                    for (int k = 0; k < KF; ++k) {
                        const int xi = xq + k - pad;
                        if (xi >= 0 && xi < L) v += ssig[ch * L + xi] * stap[ch * KF + k];       // (the source form of dfl_fwd_kernel)
                    }
            }
            st_any(J.y[p], o0 + c, obf, v);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward jobs
// ---------------------------------------------------------------------------------------------------------------------
__device__ void job_dfl_bwd(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int md = J.A[0], C = J.A[1], L = J.A[2], KF = J.A[3], nch = J.A[4], ldh = J.A[5], ibf = J.A[6];
    const int pad = KF / 2;
    const int ldi = J.ldw, ldf = J.ldy;                               // leading dims of the image / filter matrices
    for (int p = 0; p < J.npass; ++p) {
        const int B = J.M[p];
        const int T = J.T[p];
        if (B <= 0) continue;
        const void* dz = J.x[p];
        const int rows = B * T;
        if (tile < rows) {
            const int r = tile, b = r / T, t = r - b * T;
            const float* sig = reinterpret_cast<const float*>(J.P[p][0]) + (long)r * ldi;
            const float* taps = reinterpret_cast<const float*>(J.P[p][1]) + (long)r * ldf;
            float* dpre = reinterpret_cast<float*>(J.P[p][2]) + (long)r * ldi;
            float* dflt = reinterpret_cast<float*>(J.P[p][3]) + (long)r * ldf;
            float* dhe = reinterpret_cast<float*>(J.P[p][4]) + ((long)t * B + b) * ldh;
            float* sd = s.buf;
            float* ssig = sd + L;
            float* stap = ssig + nch * L;
            const long z0 = (long)r * J.ldx;
            for (int i = tid; i < L; i += TT) sd[i] = ld_any(dz, z0 + md + C + i, ibf);
            for (int i = tid; i < nch * L; i += TT) ssig[i] = sig[i];
            for (int i = tid; i < nch * KF; i += TT) stap[i] = taps[i];
            __syncthreads();
            for (int i = tid; i < ldi; i += TT) {                      // d(pre-tanh image)[c][x'] = (sum_k dout[x'-k+pad] taps[c][k]) (1 - y^2)
                float v = 0.f;
                if (i < nch * L) {
                    const int ch = i / L, xp = i - ch * L;
                    float acc = 0.f;
                    for (int k = 0; k < KF; ++k) { const int xq = xp - k + pad; if (xq >= 0 && xq < L) acc = fmaf(sd[xq], stap[ch * KF + k], acc); }
                    const float yv = ssig[i];
                    v = acc * (1.f - yv * yv);
                }
                dpre[i] = v;
            }
            for (int i = tid; i < ldf; i += TT) {                      // d taps[c][k] = sum_x dout[x] sig[c][x+k-pad]
                float v = 0.f;
                if (i < nch * KF) {
                    const int ch = i / KF, k = i - ch * KF;
                    for (int xq = 0; xq < L; ++xq) { const int xi = xq + k - pad; if (xi >= 0 && xi < L) v = fmaf(sd[xq], ssig[ch * L + xi], v); }
                }
                dflt[i] = v;
            }
            for (int c = tid; c < ldh; c += TT) dhe[c] = c < md ? ld_any(dz, z0 + c, ibf) : 0.f;
            __syncthreads();
        } else if (tile < rows + B) {
            const int b = tile - rows;
            const float* ext = reinterpret_cast<const float*>(J.y[p]);
            float* dmu = reinterpret_cast<float*>(J.P[p][5]);
            for (int jc = tid; jc < C; jc += TT) {
                float v = ext ? ext[(long)b * C + jc] : 0.f;
                for (int r = b; r < rows; r += B) v += ld_any(dz, (long)r * J.ldx + md + jc, ibf);      // the rows that read mu[b]
                dmu[(long)b * C + jc] = v;
            }
        }
    }
}

__device__ void job_bn_bwd(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int col[4];
    bool cok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { col[c] = tile * 4 + c; cok[c] = col[c] < J.N; }
    const float* gamma = reinterpret_cast<const float*>(J.Q[0]);
    float* dgamma = reinterpret_cast<float*>(J.Q[1]);
    float* dbeta = reinterpret_cast<float*>(J.Q[2]);
    for (int p = 0; p < J.npass; ++p) {
        const int M = J.M[p];
        if (M <= 0) continue;
        const float* lin = reinterpret_cast<const float*>(J.P[p][0]);
        const float* save = reinterpret_cast<const float*>(J.P[p][1]);
        const float* init = reinterpret_cast<const float*>(J.P[p][2]);
        float* dlin = reinterpret_cast<float*>(J.y[p]);
        if (J.K > 0) tile_product(s, J.x[p], J.ldx, J.w, J.ldw, col, cok, M, J.K);
        for (int i = tid; i < M * 4; i += TT) {                        // val = dy, part[0] = xhat
            const int m = i >> 2, c = i & 3;
            float dy = 0.f, xh = 0.f;
            if (cok[c]) {
                dy = J.K > 0 ? s.val[m][c] + (init ? init[(long)m * J.ldy + col[c]] : 0.f) : J.x[p][(long)m * J.ldx + col[c]];
                xh = (lin[(long)m * J.ldy + col[c]] - save[col[c]]) * save[J.ldy + col[c]];
            }
            s.val[m][c] = dy;
            s.buf[m * 4 + c] = xh;
        }
        __syncthreads();
        if (wave < 4) {
            const int c = wave;
            const float s1 = wave_rows_sum(M, [&](int m) { return s.val[m][c]; });
            const float s2 = wave_rows_sum(M, [&](int m) { return s.val[m][c] * s.buf[m * 4 + c]; });
            if (lane == 0) {
                s.stat[c][0] = s1;
                s.stat[c][1] = s2;
                if (cok[c]) {
                    if (dgamma) dgamma[col[c]] += s2;
                    if (dbeta) dbeta[col[c]] += s1;
                }
            }
        }
        __syncthreads();
        const float invM = 1.f / (float)M;
        for (int i = tid; i < M * 4; i += TT) {
            const int m = i >> 2, c = i & 3;
            if (col[c] >= J.ldy) continue;
            float v = 0.f;
            if (cok[c]) v = gamma[col[c]] * save[J.ldy + col[c]] * (s.val[m][c] - s.stat[c][0] * invM - s.buf[m * 4 + c] * s.stat[c][1] * invM);
            dlin[(long)m * J.ldy + col[c]] = v;
        }
        __syncthreads();
    }
}

__device__ void job_gru_bwd(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int H = J.A[0], ldg = J.A[1], ext_rows = J.A[2] > 0 ? J.A[2] : 1;
    int col[4];
    bool cok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { col[c] = tile * 4 + c; cok[c] = col[c] < H; }
    for (int p = 0; p < J.npass; ++p) {
        const int M = J.M[p];
        if (M <= 0) continue;
        const float* dh_ext = reinterpret_cast<const float*>(J.P[p][0]);
        const float* dhz_next = reinterpret_cast<const float*>(J.P[p][1]);
        const float* gates = reinterpret_cast<const float*>(J.P[p][2]);
        const float* hprev = reinterpret_cast<const float*>(J.P[p][3]);
        float* dgi = reinterpret_cast<float*>(J.P[p][4]);
        float* dgh = reinterpret_cast<float*>(J.P[p][5]);
        float* dhz = reinterpret_cast<float*>(J.y[p]);
        if (J.K > 0) tile_product(s, J.x[p], J.ldx, J.w, J.ldw, col, cok, M, J.K);
        for (int i = tid; i < M * 4; i += TT) {
            const int m = i >> 2, c = i & 3, j = col[c];
            if (j >= J.ldy) continue;
            if (!cok[c]) { dhz[(long)m * J.ldy + j] = 0.f; continue; }
            float dh = dh_ext ? dh_ext[(long)m * ext_rows * J.ldy + j] : 0.f;
            if (J.K > 0) dh += s.val[m][c] + (dhz_next ? dhz_next[(long)m * J.ldy + j] : 0.f);
            const float* g = gates + (long)m * 4 * H;
            const float r = g[j], z = g[H + j], n = g[2 * H + j], hn = g[3 * H + j];
            const float hp = hprev[(long)m * J.ldy + j];
            const float dn_pre = dh * (1.f - z) * (1.f - n * n);
            const float dz_pre = dh * (hp - n) * z * (1.f - z);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            float* a = dgi + (long)m * ldg;
            float* b = dgh + (long)m * ldg;
            a[j] = dr_pre; b[j] = dr_pre;
            a[H + j] = dz_pre; b[H + j] = dz_pre;
            a[2 * H + j] = dn_pre; b[2 * H + j] = dn_pre * r;
            dhz[(long)m * J.ldy + j] = dh * z;
        }
        if (tile == 0)                                                 // row pads of the gate-gradient matrices (they are K pads of the next product)
            for (int i = tid; i < M * (ldg - 3 * H); i += TT) {
                const int m = i / (ldg - 3 * H), q = 3 * H + i % (ldg - 3 * H);
                dgi[(long)m * ldg + q] = 0.f;
                dgh[(long)m * ldg + q] = 0.f;
            }
        __syncthreads();
    }
}

__device__ void job_ca_bwd(Smem& s, const cpcsv_txt_job& J, int tile) {
    const int tid = threadIdx.x;
    const int C = J.A[0];
    int col[4];
    bool cok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { col[c] = tile * 4 + c; cok[c] = col[c] < C; }
    for (int p = 0; p < J.npass; ++p) {
        const int M = J.M[p];
        if (M <= 0) continue;
        const float* dmu_tot = reinterpret_cast<const float*>(J.P[p][0]);
        const float* dlv_ext = reinterpret_cast<const float*>(J.P[p][1]);
        const float* eps = reinterpret_cast<const float*>(J.P[p][2]);
        const float* xca = reinterpret_cast<const float*>(J.P[p][3]);
        float* dx = reinterpret_cast<float*>(J.y[p]);
        tile_product(s, J.x[p], J.ldx, J.w, J.ldw, col, cok, M, J.K);
        for (int i = tid; i < M * 4; i += TT) {
            const int m = i >> 2, c = i & 3, j = col[c];
            if (!cok[c]) continue;
            const float d = s.val[m][c];
            const float mu = xca[(long)m * 2 * C + j], lv = xca[(long)m * 2 * C + C + j];
            const float dmu = d + (dmu_tot ? dmu_tot[(long)m * C + j] : 0.f);
            float dlv = dlv_ext ? dlv_ext[(long)m * C + j] : 0.f;
            if (eps) dlv += d * eps[(long)m * C + j] * 0.5f * expf(0.5f * lv);
            dx[(long)m * 2 * C + j] = mu > 0.f ? dmu : 0.f;              // through the ReLU in front of the split
            dx[(long)m * 2 * C + C + j] = lv > 0.f ? dlv : 0.f;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(TT) void text_stage_kernel(const cpcsv_txt_stage st) {
    __shared__ Smem s;
    int jb = 0;
    for (int k = 1; k < st.njobs; ++k)
        if ((int)blockIdx.x >= st.job[k].blk0) jb = k;
    const cpcsv_txt_job& J = st.job[jb];
    const int tile = (int)blockIdx.x - J.blk0;
    switch (J.type) {
        case CPCSV_TXT_DENSE: job_dense(s, J, tile); break;
        case CPCSV_TXT_CA: job_ca(s, J, tile); break;
        case CPCSV_TXT_GRU_FWD: job_gru_fwd(s, J, tile); break;
        case CPCSV_TXT_PREP: job_prep(J, tile); break;
        case CPCSV_TXT_JOINT: job_joint(s, J, tile); break;
        case CPCSV_TXT_DFL_BWD: job_dfl_bwd(s, J, tile); break;
        case CPCSV_TXT_BN_BWD: job_bn_bwd(s, J, tile); break;
        case CPCSV_TXT_GRU_BWD: job_gru_bwd(s, J, tile); break;
        case CPCSV_TXT_CA_BWD: job_ca_bwd(s, J, tile); break;
        default: break;
    }
}

inline int max2(int a, int b) { return a > b ? a : b; }

}  // namespace

extern "C" int cpcsv_text_stage(cpcsv_txt_stage* st, void* stream) {
    if (!st || st->njobs < 1 || st->njobs > CPCSV_TXT_MAX_JOBS) return -1001;
    int blocks = 0;
    for (int k = 0; k < st->njobs; ++k) {
        cpcsv_txt_job& J = st->job[k];
        if (J.npass < 1 || J.npass > 2) return -1002;
        int mmax = 0;
        for (int p = 0; p < J.npass; ++p) {
            if (J.M[p] < 0 || J.M[p] > CPCSV_TXT_MAX_ROWS) return -1003;
            mmax = max2(mmax, J.M[p]);
        }
        const bool product = J.type == CPCSV_TXT_DENSE || J.type == CPCSV_TXT_CA || J.type == CPCSV_TXT_GRU_FWD || J.type == CPCSV_TXT_CA_BWD ||
                             ((J.type == CPCSV_TXT_BN_BWD || J.type == CPCSV_TXT_GRU_BWD) && J.K > 0);
        if (product) {
            if (J.K <= 0 || (J.K & 3) || (J.ldx & 3) || (J.ldw & 3) || J.ldx < J.K || J.ldw < J.K || !J.w) return -1004;
            for (int p = 0; p < J.npass; ++p)
                if (J.M[p] > 0 && (!J.x[p] || (reinterpret_cast<uintptr_t>(J.x[p]) & 15))) return -1005;
            if (reinterpret_cast<uintptr_t>(J.w) & 15) return -1005;
        }
        int n = 0;
        switch (J.type) {
            case CPCSV_TXT_DENSE:
            case CPCSV_TXT_BN_BWD:
                if (J.N <= 0 || J.ldy < J.N) return -1006;
                n = (J.ldy + 3) / 4;
                break;
            case CPCSV_TXT_GRU_BWD:
                if (J.A[0] <= 0 || J.ldy < J.A[0] || J.A[1] < 3 * J.A[0]) return -1006;
                n = (J.ldy + 3) / 4;
                break;
            case CPCSV_TXT_CA:
                if (J.A[0] <= 0 || J.A[1] < J.A[0] || !J.bias) return -1006;
                n = J.A[1];
                break;
            case CPCSV_TXT_GRU_FWD:
                if (J.A[0] <= 0 || J.ldy < J.A[0] || J.A[1] < 3 * J.A[0] || !J.bias) return -1006;
                n = J.ldy;
                break;
            case CPCSV_TXT_CA_BWD:
                if (J.A[0] <= 0) return -1006;
                n = (J.A[0] + 3) / 4;
                break;
            case CPCSV_TXT_PREP:
                for (int p = 0; p < J.npass; ++p) n = max2(n, (J.T[p] + 1) * J.M[p]);
                break;
            case CPCSV_TXT_JOINT:
                if (J.A[4] * J.A[2] + J.A[4] * J.A[3] > 1024) return -1007;
                for (int p = 0; p < J.npass; ++p) n = max2(n, J.T[p] * J.M[p]);
                break;
            case CPCSV_TXT_DFL_BWD:
                if (J.A[2] + J.A[4] * J.A[2] + J.A[4] * J.A[3] > 1024) return -1007;
                for (int p = 0; p < J.npass; ++p) n = max2(n, (J.T[p] + 1) * J.M[p]);
                break;
            default: return -1008;
        }
        if (J.type == CPCSV_TXT_BN_BWD && mmax * 4 > 1024) return -1003;      // (xhat rides in the 1024-float scratch)
        if (n <= 0) return -1009;
        J.blk0 = blocks;
        J.nblk = n;
        blocks += n;
    }
    hipLaunchKernelGGL(text_stage_kernel, dim3((unsigned)blocks), dim3(TT), 0, (hipStream_t)stream, *st);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
