// condhead.hip — D_GET_LOGITS' 3x3 conv in factored form (reference model.py:75-80,89-92; miscc/utils.py:70-84).
//
// The head concatenates the condition vector, tiled over the 4x4 map, to the 8 ndf feature channels and runs
// SN-conv3x3 -> BatchNorm -> LeakyReLU. Literal form (rounds 1-5): one gather-GEMM over 3 N - 1 samples x 16 pixels with
// K = 9 x (8 ndf + nef) - a third of K multiplies spatially constant inputs, a third of the rows repeats feature maps ("wrong" pairs =
// real features with shifted conditions), and 44 of the 144 (pixel, tap) pairs read zero padding. Factored form (include/cpcsv_hip.h,
// cpcsv_cond_head): the feature conv once per DISTINCT feature map (cpcsv_gemm_nt, K = 9 x 8 ndf, fp32 K-slice slabs), the condition
// channels as one dense product per distinct condition row and tap (cpcsv_gemm_nt with bcol_rows), and THIS file's kernel, which
// assembles every call's conv output from the two, finishes train-mode BatchNorm and applies the activation in one launch:
//   block = 4 output channels x all rows of all calls; thread = one map row (4 pixels) of one sample -> the statistics of a channel
//   never leave the block (no partial rows, no finalize launch), the running statistics see the calls in order, and the
//   condition part of a sample is 9 loads (one per tap) dealt to its pixels in registers (a tap is live at a pixel when it
//   stays inside the map: the 44 dead pairs cost nothing here).
// HBM-side: reads the slabs once (L2-resident: the GEMM has just written them), writes z and y once.
#include "common.h"
#include "../../include/cpcsv_hip.h"

namespace {

constexpr int CH_THREADS = 1024;
constexpr int CH_MAXP = 16;        // pixels per sample (4 x 4 map)
constexpr int CH_TPS = 4;          // threads per sample: a thread owns 4 of its 16 pixels (one map row)
constexpr int CH_PPT = CH_MAXP / CH_TPS;
constexpr int CH_SPT = 2;          // samples per thread slot: up to 512 samples per launch (3 x 160 - 1: the ST=32 / IM=160 batch)

// the map is 4 x 4 (64x64 images behind four stride-2 convs, reference model.py:498-514)
constexpr int CH_MH = 4, CH_MW = 4;
__device__ __forceinline__ constexpr bool tap_live(int p, int tap) {
    const int py = p / CH_MW, px = p - py * CH_MW;
    const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
    return iy >= 0 && iy < CH_MH && ix >= 0 && ix < CH_MW;
}
__device__ __forceinline__ constexpr unsigned live_mask(int p) {
    unsigned m = 0;
    for (int tap = 0; tap < 9; ++tap) m |= tap_live(p, tap) ? (1u << tap) : 0u;
    return m;
}

// One block = 4 output channels x ALL rows of all calls (1024 threads: a thread owns one map row - 4 pixels - of one sample), so a
// channel's batch statistics never leave the block. (First form: 256 threads, a whole sample per thread: 50 us per launch - four
// wavefronts per CU, every thread behind 73 dependent-latency loads; this form has 16 wavefronts per CU and 25 loads per thread.)
template <typename T>
__global__ __launch_bounds__(CH_THREADS) void cond_head_fwd_kernel(const cpcsv_cond_head d) {
    constexpr int NW = CH_THREADS / 64;
    __shared__ double red[NW][4][2][4];         // [wave][group][sum | sumsq][column]
    __shared__ float par[4][4][4];              // [group][mean | invstd | scale | shift][column]
    __shared__ double varsh[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block -> column map: consecutive block ids land on different XCDs, and a block reads 16 bytes of every 128-byte line
    // of the slabs - with c0 = 4 * blockIdx.x the 8 blocks that share a line sat on 8 XCDs and each L2 fetched the line for itself
    // (8 x 30 MB through the fabric: 50 us). Blocks of one XCD now own neighbouring columns.
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, per = nblk >> 3, rem = nblk & 7;
    const int c0 = 4 * ((xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + (blockIdx.x >> 3));
    constexpr int P = CH_MAXP;
    int total = 0;
    for (int g = 0; g < d.ngroups; ++g) total += d.count[g];

    f32x4 t[CH_SPT][CH_PPT];
    int sg[CH_SPT], sn[CH_SPT];                 // group and index-in-group of this thread's samples (-1: none)
    const int quad = tid & (CH_TPS - 1);        // which map row of its sample
    float acc[4][2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g][k][e] = 0.f;

#pragma unroll
    for (int s = 0; s < CH_SPT; ++s) {
        int idx = (tid + s * CH_THREADS) / CH_TPS;
        sg[s] = -1; sn[s] = 0;
        if (idx < total) {
            int g = 0;
            while (idx >= d.count[g]) { idx -= d.count[g]; ++g; }
            sg[s] = g; sn[s] = idx;
        }
        if (sg[s] < 0) continue;
        const int g = sg[s];
        const float alpha = d.galpha[g] ? *d.galpha[g] : 1.f;
        // condition part: the 9 tap products of this sample's condition row
        f32x4 ct[9];
        const float* prow = d.pt + ((long)(d.cond0[g] + sn[s]) * 9) * d.ldp + c0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) ct[tap] = *reinterpret_cast<const f32x4*>(prow + (long)tap * d.ldp);
        const long frow0 = (long)(d.feat0[g] + sn[s]) * P + quad * CH_PPT;
        const long slab = d.ws_rows * d.ldws;
#pragma unroll
        for (int j = 0; j < CH_PPT; ++j) {
            const float* src = d.ws + (frow0 + j) * d.ldws + c0;
            f32x4 f = *reinterpret_cast<const f32x4*>(src);
            for (int k = 1; k < d.nslabs; ++k) f += *reinterpret_cast<const f32x4*>(src + (long)k * slab);
            // taps that stay inside the map at pixel p = quad * 4 + j (the dead pairs of the literal form cost nothing here)
            unsigned live = 0;
#pragma unroll
            for (int q = 0; q < CH_TPS; ++q) live = quad == q ? live_mask(q * CH_PPT + j) : live;
            f32x4 add = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                if (live & (1u << tap)) add += ct[tap];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = c0 + e < d.C ? __fmul_rn(f[e] + add[e], alpha) : 0.f;
            t[s][j] = v;
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            if (gg == g)
#pragma unroll
                for (int j = 0; j < CH_PPT; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[gg][0][e] += t[s][j][e]; acc[gg][1][e] += t[s][j][e] * t[s][j][e]; }
    }
    // block totals per group and column: fp32 over a thread's 4-8 values, double from there on, fixed order
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g >= d.ngroups) break;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double v = (double)acc[g][k][e];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
                if (lane == 0) red[wave][g][k][e] = v;
            }
    }
    __syncthreads();
    if (tid < 4 * d.ngroups) {
        const int g = tid >> 2, e = tid & 3, c = c0 + e;
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < NW; ++w) { s1 += red[w][g][0][e]; s2 += red[w][g][1][e]; }
        const double cnt = (double)d.count[g] * P;
        float mu_f = 0.f, is = 0.f, scale = 0.f, shift = 0.f;
        double var = 0.0;
        if (c < d.C) {
            const double mu = s1 / cnt;
            var = s2 / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            is = (float)(1.0 / sqrt(var + (double)d.eps));
            mu_f = (float)mu;
            bn_affine(d.gamma[c], d.beta[c], mu_f, is, scale, shift);
        }
        par[g][0][e] = mu_f; par[g][1][e] = is; par[g][2][e] = scale; par[g][3][e] = shift;
        varsh[g][e] = var;
        if (c < d.Cs) {
            float* so = d.stat_out + (long)g * d.pstride + c;
            so[0L * d.Cs] = mu_f; so[1L * d.Cs] = is; so[2L * d.Cs] = scale; so[3L * d.Cs] = shift;
            if (d.bwd_sums)
                for (int k = 0; k < 2 * CPCSV_BN_SUM_COPIES; ++k) so[(4L + k) * d.Cs] = 0.f;
        }
    }
    __syncthreads();
    if (tid < 4 && d.running_mean && c0 + tid < d.C) {
        // running statistics: every call's batch, in call order (r <- (1-m) r + m b does not commute)
        const int c = c0 + tid;
        float rm = d.running_mean[c], rv = d.running_var[c];
        for (int g = 0; g < d.ngroups; ++g) {
            const double cn = (double)d.count[g] * P;
            const double unbias = cn > 1.0 ? cn / (cn - 1.0) : 1.0;
            rm = bn_running(rm, par[g][0][tid], d.momentum);
            rv = bn_running(rv, (float)(varsh[g][tid] * unbias), d.momentum);
        }
        d.running_mean[c] = rm; d.running_var[c] = rv;
    }
    const ActPl apl = act_pl(d.act);
    long gbase[4];
    {
        long r = 0;
        for (int g = 0; g < 4; ++g) { gbase[g] = r; if (g < d.ngroups) r += (long)d.count[g] * P; }
    }
#pragma unroll
    for (int s = 0; s < CH_SPT; ++s) {
        if (sg[s] < 0) continue;
        const int g = sg[s];
        float sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = par[g][2][e]; sh[e] = par[g][3][e]; }
        const long row0 = gbase[g] + (long)sn[s] * P + quad * CH_PPT;
#pragma unroll
        for (int j = 0; j < CH_PPT; ++j) {
            const long off = (row0 + j) * d.Cs + c0;
            if (sizeof(T) == 2) {
                bf16_t zb[4], yb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    zb[e] = f32_to_bf16(t[s][j][e]);
                    const float pre = bn_pre(bf16_to_f32(zb[e]), sc[e], sh[e]);
                    yb[e] = f32_to_bf16(c0 + e < d.C ? act_apply_t<false>(pre, d.act, apl) : 0.f);
                }
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(d.z) + off) =
                    u32x2{(uint32_t)zb[0] | ((uint32_t)zb[1] << 16), (uint32_t)zb[2] | ((uint32_t)zb[3] << 16)};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(d.y) + off) =
                    u32x2{(uint32_t)yb[0] | ((uint32_t)yb[1] << 16), (uint32_t)yb[2] | ((uint32_t)yb[3] << 16)};
            } else {
                f32x4 yv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pre = bn_pre(t[s][j][e], sc[e], sh[e]);
                    yv[e] = c0 + e < d.C ? act_apply_t<false>(pre, d.act, apl) : 0.f;
                }
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.z) + off) = t[s][j];
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(d.y) + off) = yv;
            }
        }
    }
}

// backward glue: one thread = one (feature sample | condition row, 16-byte channel chunk); fixed summation order (calls in order)
template <typename T>
__global__ __launch_bounds__(128) void cond_head_bwd_kernel(const cpcsv_cond_head_grad d) {
    constexpr int EPC = elem<T>::per16;
    const int cpr = d.Cs / EPC;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int P = CH_MAXP;
    const long nf = (long)d.nfeat * cpr, nc = d.dZt ? (long)d.ncond * cpr : 0;
    if (i >= nf + nc) return;
    long gbase[4];
    {
        long r = 0;
        for (int g = 0; g < 4; ++g) { gbase[g] = r; if (g < d.ngroups) r += (long)d.count[g] * P; }
    }
    const u32x4* dz = reinterpret_cast<const u32x4*>(d.dz);
    if (i < nf) {
        const int s = (int)(i / cpr), ch = (int)(i - (long)s * cpr);
        for (int p = 0; p < P; ++p) {
            float a[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) a[e] = 0.f;
            for (int g = 0; g < d.ngroups; ++g) {
                const int n = s - d.feat0[g];
                if (n < 0 || n >= d.count[g]) continue;
                const u32x4 raw = dz[(gbase[g] + (long)n * P + p) * cpr + ch];
                const T* v = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int e = 0; e < EPC; ++e) a[e] += elem<T>::ld(v + e);
            }
            u32x4 out;
            T* o = reinterpret_cast<T*>(&out);
#pragma unroll
            for (int e = 0; e < EPC; ++e) elem<T>::st(o + e, a[e]);
            reinterpret_cast<u32x4*>(d.dF)[((long)s * P + p) * cpr + ch] = out;
        }
        return;
    }
    const long k = i - nf;
    const int c = (int)(k / cpr), ch = (int)(k - (long)c * cpr);
    float tsum[9][EPC];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < EPC; ++e) tsum[tap][e] = 0.f;
    for (int g = 0; g < d.ngroups; ++g) {
        const int n = c - d.cond0[g];
        if (n < 0 || n >= d.count[g]) continue;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const u32x4 raw = dz[(gbase[g] + (long)n * P + p) * cpr + ch];
            const T* v = reinterpret_cast<const T*>(&raw);
            float x[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) x[e] = elem<T>::ld(v + e);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                if (tap_live(p, tap)) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) tsum[tap][e] += x[e];
                }
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        u32x4 out;
        T* o = reinterpret_cast<T*>(&out);
#pragma unroll
        for (int e = 0; e < EPC; ++e) elem<T>::st(o + e, tsum[tap][e]);
        reinterpret_cast<u32x4*>(d.dZt)[((long)c * 9 + tap) * cpr + ch] = out;
    }
}

}  // namespace

extern "C" int cpcsv_cond_head_fwd(const cpcsv_cond_head* d, void* stream) {
    if (!d || !d->ws || !d->pt || !d->z || !d->y || !d->gamma || !d->beta || !d->stat_out) return -1001;
    if (d->ngroups < 1 || d->ngroups > 4 || d->MH != CH_MH || d->MW != CH_MW) return -1002;
    if (d->Cs % 8 || d->C <= 0 || d->C > d->Cs || d->ldws % 4 || d->ldp % 4 || d->ldws < d->Cs || d->ldp < d->Cs || d->nslabs < 1) return -1003;
    if (d->act > CPCSV_ACT_LRELU) return -1004;
    long total = 0;
    for (int g = 0; g < d->ngroups; ++g) {
        if (d->count[g] < 1 || d->feat0[g] < 0 || d->cond0[g] < 0) return -1002;
        if ((long)(d->feat0[g] + d->count[g]) * d->MH * d->MW > d->ws_rows) return -1005;
        total += d->count[g];
    }
    if (total > (long)CH_THREADS * CH_SPT / CH_TPS) return -1006;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == CPCSV_BF16) hipLaunchKernelGGL(cond_head_fwd_kernel<bf16_t>, dim3(d->Cs / 4), dim3(CH_THREADS), 0, s, *d);
    else hipLaunchKernelGGL(cond_head_fwd_kernel<float>, dim3(d->Cs / 4), dim3(CH_THREADS), 0, s, *d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

// samples one launch of cpcsv_cond_head_fwd can take (the host falls back to the literal form beyond it)
extern "C" int cpcsv_cond_head_max_samples(void) { return CH_THREADS * CH_SPT / CH_TPS; }

extern "C" int cpcsv_cond_head_bwd(const cpcsv_cond_head_grad* d, void* stream) {
    if (!d || !d->dz || !d->dF) return -1001;
    if (d->ngroups < 1 || d->ngroups > 4 || d->MH != CH_MH || d->MW != CH_MW || d->Cs % 8 || d->nfeat < 1 || d->ncond < 0) return -1002;
    hipStream_t s = (hipStream_t)stream;
    const int cpr = d->Cs / (d->dtype == CPCSV_BF16 ? 8 : 4);
    const long n = (long)d->nfeat * cpr + (d->dZt ? (long)d->ncond * cpr : 0);
    if (d->dtype == CPCSV_BF16) hipLaunchKernelGGL(cond_head_bwd_kernel<bf16_t>, dim3(cdiv(n, 128)), dim3(128), 0, s, *d);
    else hipLaunchKernelGGL(cond_head_bwd_kernel<float>, dim3(cdiv(n, 128)), dim3(128), 0, s, *d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
