// elementwise.hip — streaming glue kernels: activation/gate backward, layout changes, concat,
// temporal mean. HBM-bound; 16-byte accesses where the layout allows.
#include "common.h"
#include "../../include/cpcsv_hip.h"

namespace {

inline int grid_for(long n, int block = 256, int cap = 8192) {
    long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// One 16-byte vector (8 bf16 / 4 fp32) per thread and trip; `nv` = vectors, the < 16-byte tail is walked element by element by the
// first threads. ACT is a template argument: a run-time switch per element is a scalar branch tree per element.
template <typename T, int ACT>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dz, long n) {
    constexpr int EPC = elem<T>::per16;
    const long nv = n / EPC;
    GRID_STRIDE(i, nv) {
        const u32x4 gv = reinterpret_cast<const u32x4*>(dy)[i], yv = reinterpret_cast<const u32x4*>(y)[i];
        const T* pg = reinterpret_cast<const T*>(&gv);
        const T* py = reinterpret_cast<const T*>(&yv);
        u32x4 ov;
        T* po = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int e = 0; e < EPC; ++e) elem<T>::st(po + e, elem<T>::ld(pg + e) * act_grad_from_out(elem<T>::ld(py + e), ACT));
        reinterpret_cast<u32x4*>(dz)[i] = ov;
    }
    const long t = nv * EPC + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) elem<T>::st(dz + t, elem<T>::ld(dy + t) * act_grad_from_out(elem<T>::ld(y + t), ACT));
}

template <typename T>
__global__ void gate_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long n) {
    constexpr int EPC = elem<T>::per16;
    const long nv = n / EPC;
    GRID_STRIDE(i, nv) {
        const u32x4 av = reinterpret_cast<const u32x4*>(a)[i], bv4 = reinterpret_cast<const u32x4*>(b)[i];
        const T* pa = reinterpret_cast<const T*>(&av);
        const T* pb = reinterpret_cast<const T*>(&bv4);
        u32x4 ov;
        T* po = reinterpret_cast<T*>(&ov);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float bv = elem<T>::ld(pb + e); elem<T>::st(po + e, elem<T>::ld(pa + e) * bv + bv); }
        reinterpret_cast<u32x4*>(o)[i] = ov;
    }
    const long t = nv * EPC + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) { const float bv = elem<T>::ld(b + t); elem<T>::st(o + t, elem<T>::ld(a + t) * bv + bv); }
}
template <typename T>
__global__ void gate_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a, const T* __restrict__ b,
                                T* __restrict__ da, T* __restrict__ db, long n) {
    constexpr int EPC = elem<T>::per16;
    const long nv = n / EPC;
    GRID_STRIDE(i, nv) {
        const u32x4 gv4 = reinterpret_cast<const u32x4*>(g)[i], av = reinterpret_cast<const u32x4*>(a)[i], bv4 = reinterpret_cast<const u32x4*>(b)[i];
        const T* pg = reinterpret_cast<const T*>(&gv4);
        const T* pa = reinterpret_cast<const T*>(&av);
        const T* pb = reinterpret_cast<const T*>(&bv4);
        u32x4 oa, ob;
        T* poa = reinterpret_cast<T*>(&oa);
        T* pob = reinterpret_cast<T*>(&ob);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float gv = elem<T>::ld(pg + e);
            elem<T>::st(poa + e, gv * elem<T>::ld(pb + e));
            elem<T>::st(pob + e, gv * (elem<T>::ld(pa + e) + 1.f));
        }
        reinterpret_cast<u32x4*>(da)[i] = oa;
        reinterpret_cast<u32x4*>(db)[i] = ob;
    }
    const long t = nv * EPC + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        const float gv = elem<T>::ld(g + t);
        elem<T>::st(da + t, gv * elem<T>::ld(b + t));
        elem<T>::st(db + t, gv * (elem<T>::ld(a + t) + 1.f));
    }
}

template <typename S, typename D>
__global__ void planar_to_nhwc_kernel(const S* __restrict__ src, D* __restrict__ dst, long total, int T, long sB,
                                      long sT, long sC, int C, int HW, int Cs) {
    GRID_STRIDE(i, total) {   // i over dst [frames][HW][Cs]
        const int c = (int)(i % Cs);
        const long p = (i / Cs) % HW, f = i / ((long)Cs * HW);
        elem<D>::st(dst + i, c < C ? elem<S>::ld(src + (f / T) * sB + (f % T) * sT + c * sC + p) : 0.f);
    }
}
// The image tensors of the step (1 or 3 channels in an 8-channel NHWC pixel): one thread per PIXEL - up to 8 coalesced planar loads
// (consecutive threads = consecutive pixels of a plane), one 16-byte (bf16) / two 16-byte (fp32) stores, int32 index arithmetic per
// pixel. The generic kernel above walks ELEMENTS with five 64-bit divisions each (1.4 TB/s on the critics' input batches).
template <typename S, typename D>
__global__ void planar_to_nhwc8_kernel(const S* __restrict__ src, D* __restrict__ dst, int npix, int T, long sB, long sT, long sC, int C,
                                       int HW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;            // pixel index over [frames][HW]
    if (i >= npix) return;
    const int f = i / HW, p = i - f * HW;
    const S* s0 = src + (long)(f / T) * sB + (long)(f % T) * sT + p;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < C ? elem<S>::ld(s0 + c * sC) : 0.f;
    if (sizeof(D) == 2) {
        u32x4 pk;
        D* po = reinterpret_cast<D*>(&pk);
#pragma unroll
        for (int c = 0; c < 8; ++c) elem<D>::st(po + c, v[c]);
        reinterpret_cast<u32x4*>(dst)[i] = pk;
    } else {
        f32x4* o = reinterpret_cast<f32x4*>(dst) + (long)i * 2;
        o[0] = f32x4{v[0], v[1], v[2], v[3]};
        o[1] = f32x4{v[4], v[5], v[6], v[7]};
    }
}

template <typename S, typename D>
__global__ void nhwc_to_planar_kernel(const S* __restrict__ src, D* __restrict__ dst, long total, int T, long sB,
                                      long sT, long sC, int C, int HW, int Cs) {
    GRID_STRIDE(i, total) {   // i over (frames, C, HW)
        const long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long f = i / ((long)HW * C);
        elem<D>::st(dst + (f / T) * sB + (f % T) * sT + c * sC + p, elem<S>::ld(src + (f * HW + p) * Cs + c));
    }
}
// the same copy for SMALL maps with MANY channels (the 4x4 x 1024 gradient that enters the generator's dense layer: 2 M
// elements): the element-per-thread form above reads one 2-byte element per 2 KB row there (58 us). Block = one frame x 64
// channels staged through LDS: rows of 64 channels in, [64][HW] runs (contiguous in dst when sC == HW) out.
template <typename S, typename D>
__global__ __launch_bounds__(256) void nhwc_to_planar_tiled_kernel(const S* __restrict__ src, D* __restrict__ dst, int T, long sB, long sT,
                                                                   int C, int HW, int Cs) {
    __shared__ float tile[64 * 65];
    const long f = blockIdx.y;
    const int c0 = blockIdx.x * 64, nc = C - c0 < 64 ? C - c0 : 64;
    for (int i = threadIdx.x; i < HW * 64; i += 256) {
        const int p = i >> 6, c = i & 63;
        if (c < nc) tile[c * 65 + p] = elem<S>::ld(src + (f * HW + p) * Cs + c0 + c);
    }
    __syncthreads();
    D* out = dst + (f / T) * sB + (f % T) * sT + (long)c0 * HW;
    for (int i = threadIdx.x; i < nc * HW; i += 256) elem<D>::st(out + i, tile[(i / HW) * 65 + (i % HW)]);
}

// ... and the other direction for the same shapes (the generator's `fc` / `fc_seg` output [frames][C*HW] viewed as (C, 4, 4) -> NHWC:
// FeatToNhwcFn forward, reference model.py:380,382): block = one frame x 64 channels through LDS, contiguous [64][HW] runs in,
// rows of 64 channels out (zero channel pads). The element-per-thread kernel took 16.7 us for the 3.9 MB of `fc`'s output.
template <typename S, typename D>
__global__ __launch_bounds__(256) void planar_to_nhwc_tiled_kernel(const S* __restrict__ src, D* __restrict__ dst, int T, long sB, long sT,
                                                                   int C, int HW, int Cs) {
    __shared__ float tile[64 * 65];
    const long f = blockIdx.y;
    const int c0 = blockIdx.x * 64, nc = C - c0 < 64 ? (C - c0 > 0 ? C - c0 : 0) : 64;
    const S* in = src + (f / T) * sB + (f % T) * sT + (long)c0 * HW;
    for (int i = threadIdx.x; i < nc * HW; i += 256) tile[(i / HW) * 65 + (i % HW)] = elem<S>::ld(in + i);
    __syncthreads();
    const int ncs = Cs - c0 < 64 ? Cs - c0 : 64;                   // stored channels of this block (pads included)
    for (int i = threadIdx.x; i < HW * 64; i += 256) {
        const int p = i >> 6, c = i & 63;
        if (c < ncs) elem<D>::st(dst + (f * HW + p) * Cs + c0 + c, c < nc ? tile[c * 65 + p] : 0.f);
    }
}

// F4 input pipeline, device half: pre-decoded uint8 HWC frames -> what the reference's torchvision chain yields
// (main_pororo.py:71-84: ToTensor = x/255 in fp32, Normalize = (t - mean)/std, then `video_transform` stacks frames and
// permutes to (C,T,H,W)): the fp32 channel-planar batch tensor and, in the same pass, the NHWC compute-dtype frames the
// critics' first conv reads. One thread per pixel; true IEEE divisions so the fp32 result equals torch's bit for bit.
template <typename D>
__global__ void ingest_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ planar, D* __restrict__ nhwc, long npix,
                                 int T, long sB, long sT, long sC, int C, int HW, int Cs, const float* __restrict__ mean,
                                 const float* __restrict__ stdv) {
    GRID_STRIDE(i, npix) {                       // i over [frames][HW]
        const long f = i / HW, p = i - f * HW;
        const uint8_t* s = src + i * C;
        float* pl = planar ? planar + (f / T) * sB + (f % T) * sT + p : nullptr;
        D* nh = nhwc ? nhwc + i * Cs : nullptr;
        for (int c = 0; c < Cs; ++c) {
            float v = 0.f;
            if (c < C) {
                v = __fdiv_rn(__fsub_rn(__fdiv_rn((float)s[c], 255.0f), mean[c]), stdv[c]);
                if (pl) pl[c * sC] = v;
            }
            if (nh) elem<D>::st(nh + c, v);
        }
    }
}

template <typename S, typename D>
__global__ void copy2d_kernel(const S* __restrict__ src, long lds, int scol0, D* __restrict__ dst, long ldd, int dcol0,
                              long rows, int cols, int accumulate) {
    if (accumulate == 2) {       // the whole dst row [0, ldd) is written: zeros outside the copied column window
        const long total = rows * ldd;
        GRID_STRIDE(i, total) {
            const long r = i / ldd;
            const int c = (int)(i % ldd) - dcol0;
            elem<D>::st(dst + i, (c >= 0 && c < cols) ? elem<S>::ld(src + r * lds + scol0 + c) : 0.f);
        }
        return;
    }
    const long total = rows * cols;
    GRID_STRIDE(i, total) {
        const long r = i / cols;
        const int c = (int)(i % cols);
        const float v = elem<S>::ld(src + r * lds + scol0 + c);
        D* p = dst + r * ldd + dcol0 + c;
        elem<D>::st(p, accumulate ? elem<D>::ld(p) + v : v);
    }
}

struct ConcatSrc { const float* p[4]; int w[4]; int col0[4]; int n; };
// dst[r][:] = [src0[r][:w0] | src1[r][:w1] | ... | 0-pad]  (torch.cat + pad + cast, one launch)
template <typename D>
__global__ void concat_pad_kernel(ConcatSrc src, D* __restrict__ dst, long rows, int ldd) {
    const long total = rows * ldd;
    GRID_STRIDE(i, total) {
        const long r = i / ldd;
        const int c = (int)(i % ldd);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < src.n && c >= src.col0[k] && c < src.col0[k] + src.w[k]) v = src.p[k][r * src.w[k] + (c - src.col0[k])];
        elem<D>::st(dst + i, v);
    }
}

template <typename T>
__global__ void cond_concat_kernel(const T* __restrict__ feat, const float* __restrict__ cond, T* __restrict__ out,
                                   long total, int P, int C, int Cs_f, int E, int Cs_out) {
    GRID_STRIDE(i, total) {   // i over out [N][P][Cs_out]
        const int c = (int)(i % Cs_out);
        const long np = i / Cs_out;
        const long n = np / P;
        float v = 0.f;
        if (c < C) v = elem<T>::ld(feat + np * Cs_f + c);
        else if (c >= Cs_f && c < Cs_f + E) v = cond[n * E + (c - Cs_f)];
        elem<T>::st(out + i, v);
    }
}

// D_GET_LOGITS inputs of a critic update in ONE tensor (miscc/utils.py:74-84): feat holds the real features (rows [0,N)) and
// the fake features (rows [N,2N)); out rows [0,N) = (real_i, cond_i), [N,2N-1) = (real_i, cond_{i+1}) (the "wrong" pairs),
// [2N-1,3N-1) = (fake_i, cond_i); cond tiled over the P pixels and concatenated on channels like cond_concat_kernel.
template <typename T>
__global__ void cond_triplet_kernel(const T* __restrict__ feat, const float* __restrict__ cond, T* __restrict__ out,
                                    long total, int N, int P, int C, int Cs_f, int E, int Cs_out) {
    GRID_STRIDE(i, total) {   // i over out [3N-1][P][Cs_out]
        const int c = (int)(i % Cs_out);
        const long rp = i / Cs_out;
        const long r = rp / P, p = rp - r * P;
        long fr, cr;
        if (r < N) { fr = r; cr = r; }
        else if (r < 2L * N - 1) { fr = r - N; cr = r - N + 1; }
        else { fr = r - (2L * N - 1) + N; cr = r - (2L * N - 1); }
        float v = 0.f;
        if (c < C) v = elem<T>::ld(feat + (fr * P + p) * Cs_f + c);
        else if (c >= Cs_f && c < Cs_f + E) v = cond[cr * E + (c - Cs_f)];
        elem<T>::st(out + i, v);
    }
}
template <typename T>
__global__ void cond_triplet_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dfeat, long total, int N, int P, int C,
                                        int Cs_f, int Cs_out) {
    GRID_STRIDE(i, total) {   // i over dfeat [2N][P][Cs_f]
        const int c = (int)(i % Cs_f);
        const long rp = i / Cs_f;
        const long r = rp / P, p = rp - r * P;
        float v = 0.f;
        if (c < C) {
            if (r < N) {
                v = elem<T>::ld(dout + (r * P + p) * Cs_out + c);
                if (r < N - 1) v += elem<T>::ld(dout + ((N + r) * P + p) * Cs_out + c);
            } else {
                v = elem<T>::ld(dout + ((2L * N - 1 + (r - N)) * P + p) * Cs_out + c);
            }
        }
        elem<T>::st(dfeat + i, v);
    }
}

// patch matrix of a k x k, stride s, pad p convolution over NHWC frames [F][H][W][Cs] with C real channels:
// out[(f, oy, ox)][c*k*k + ky*k + kx] = x[f][oy*s + ky - p][ox*s + kx - p][c] (0 outside), columns >= C*k*k zero.
// The column order (c, ky, kx) is the master weight's [Cout][Cin][k][k] flattening, so the conv becomes a dense layer
// on this matrix with the master viewed as [Cout][C*k*k]. Used for the order critic's 7x7 stem (49 taps exceed the
// gather-GEMM's tap table; 3 input channels, 0.8 GFLOP).
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ x, T* __restrict__ out, long total, int H, int W, int Cs, int C, int k, int s,
                              int p, int OH, int OW, int ld) {
    GRID_STRIDE(i, total) {   // i over out [F*OH*OW][ld]
        const int j = (int)(i % ld);
        const long row = i / ld;
        float v = 0.f;
        if (j < C * k * k) {
            const int c = j / (k * k), r = j - c * k * k, ky = r / k, kx = r - ky * k;
            const int ox = (int)(row % OW), oy = (int)((row / OW) % OH);
            const long f = row / ((long)OW * OH);
            const int y = oy * s + ky - p, xx = ox * s + kx - p;
            if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) v = elem<T>::ld(x + ((f * H + y) * W + xx) * Cs + c);
        }
        elem<T>::st(out + i, v);
    }
}
// its adjoint in gather form: dx[f][y][x][c] = sum over (ky, kx) with (y + p - ky) % s == 0 ... of dcol[(f, oy, ox)][c*k*k + ky*k + kx]
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, long total, int H, int W, int Cs, int C, int k, int s,
                              int p, int OH, int OW, int ld) {
    GRID_STRIDE(i, total) {   // i over dx [F][H][W][Cs]
        const int c = (int)(i % Cs);
        const long pix = i / Cs;
        float acc = 0.f;
        if (c < C) {
            const int xx = (int)(pix % W), y = (int)((pix / W) % H);
            const long f = pix / ((long)W * H);
            for (int ky = 0; ky < k; ++ky) {
                const int ty = y + p - ky;
                if (ty < 0 || ty % s) continue;
                const int oy = ty / s;
                if (oy >= OH) continue;
                for (int kx = 0; kx < k; ++kx) {
                    const int tx = xx + p - kx;
                    if (tx < 0 || tx % s) continue;
                    const int ox = tx / s;
                    if (ox >= OW) continue;
                    acc += elem<T>::ld(dcol + ((f * OH + oy) * OW + ox) * ld + c * k * k + ky * k + kx);
                }
            }
        }
        elem<T>::st(dx + i, acc);
    }
}

template <typename T>
__global__ void mean_t_kernel(const T* __restrict__ in, T* __restrict__ out, long total, int Tn, long inner) {
    const float inv = 1.f / (float)Tn;
    GRID_STRIDE(i, total) {   // i over out [N][inner]
        const long n = i / inner, k = i % inner;
        float acc = 0.f;
        for (int t = 0; t < Tn; ++t) acc += elem<T>::ld(in + (n * Tn + t) * inner + k);
        elem<T>::st(out + i, acc * inv);
    }
}
template <typename T>
__global__ void mean_t_bwd_kernel(const T* __restrict__ dout, T* __restrict__ din, long total, int Tn, long inner) {
    const float inv = 1.f / (float)Tn;
    GRID_STRIDE(i, total) {   // i over din [N*T][inner]
        const long k = i % inner, n = i / (inner * Tn);
        elem<T>::st(din + i, elem<T>::ld(dout + n * inner + k) * inv);
    }
}

template <typename T>
__global__ void scale_by_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ alpha, float mult,
                                long n, int accumulate) {
    const float a = (alpha ? alpha[0] : 1.f) * mult;
    GRID_STRIDE(i, n) {
        const float v = elem<T>::ld(x + i) * a;
        elem<T>::st(y + i, accumulate ? elem<T>::ld(y + i) + v : v);
    }
}

}  // namespace

extern "C" int cpcsv_act_bwd(const void* dy, const void* y, void* dz, int dtype, long n, int act, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!dy || !y || !dz || ((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dz) & 15) return -1001;
    const int g = grid_for(n / (dtype == CPCSV_BF16 ? 8 : 4) + 1);
#define CPCSV_ACT_BWD(A)                                                                                                                        \
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16_t, A>), dim3(g), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dz, n); \
    else hipLaunchKernelGGL((act_bwd_kernel<float, A>), dim3(g), dim3(256), 0, s, (const float*)dy, (const float*)y, (float*)dz, n)
    switch (act) {
        case CPCSV_ACT_RELU: CPCSV_ACT_BWD(1); break;
        case CPCSV_ACT_LRELU: CPCSV_ACT_BWD(2); break;
        case CPCSV_ACT_TANH: CPCSV_ACT_BWD(3); break;
        case CPCSV_ACT_SIGMOID: CPCSV_ACT_BWD(4); break;
        default: CPCSV_ACT_BWD(0); break;
    }
#undef CPCSV_ACT_BWD
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_gate_fwd(const void* a, const void* b, void* out, int dtype, long n, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return -1001;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(gate_fwd_kernel<bf16_t>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_gate_bwd(const void* dout, const void* a, const void* b, void* da, void* db, int dtype, long n, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (((uintptr_t)dout | (uintptr_t)a | (uintptr_t)b | (uintptr_t)da | (uintptr_t)db) & 15) return -1001;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(gate_bwd_kernel<bf16_t>, dim3(grid_for(n / 8 + 1)), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)da, (bf16_t*)db, n);
    else hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, (const float*)dout, (const float*)a, (const float*)b, (float*)da, (float*)db, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_planar_to_nhwc(const void* src, int sd, void* dst, int dd, int frames, int T, long sB, long sT,
                                    long sC, int C, int HW, int Cs, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)frames * HW * Cs;
    if (Cs == 8 && C <= 8 && (long)frames * HW < (1L << 31) && ((uintptr_t)dst & 15) == 0) {
        const int npix = frames * HW, gp = (npix + 255) / 256;
        if (sd == CPCSV_F32 && dd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc8_kernel<float, float>), dim3(gp), dim3(256), 0, s, (const float*)src, (float*)dst, npix, T, sB, sT, sC, C, HW);
        else if (sd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc8_kernel<float, bf16_t>), dim3(gp), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, npix, T, sB, sT, sC, C, HW);
        else if (dd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc8_kernel<bf16_t, float>), dim3(gp), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, npix, T, sB, sT, sC, C, HW);
        else hipLaunchKernelGGL((planar_to_nhwc8_kernel<bf16_t, bf16_t>), dim3(gp), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, npix, T, sB, sT, sC, C, HW);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    if (HW <= 64 && C >= 64 && sC == HW && frames <= 65535) {
        const dim3 grid((Cs + 63) / 64, frames);
        if (sd == CPCSV_F32 && dd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc_tiled_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, T, sB, sT, C, HW, Cs);
        else if (sd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc_tiled_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst, T, sB, sT, C, HW, Cs);
        else if (dd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc_tiled_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)src, (float*)dst, T, sB, sT, C, HW, Cs);
        else hipLaunchKernelGGL((planar_to_nhwc_tiled_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, T, sB, sT, C, HW, Cs);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    const int g = grid_for(total);
    if (sd == CPCSV_F32 && dd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, total, T, sB, sT, sC, C, HW, Cs);
    else if (sd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, total, T, sB, sT, sC, C, HW, Cs);
    else if (dd == CPCSV_F32) hipLaunchKernelGGL((planar_to_nhwc_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, total, T, sB, sT, sC, C, HW, Cs);
    else hipLaunchKernelGGL((planar_to_nhwc_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, total, T, sB, sT, sC, C, HW, Cs);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_nhwc_to_planar(const void* src, int sd, void* dst, int dd, int frames, int T, long sB, long sT,
                                    long sC, int C, int HW, int Cs, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)frames * HW * C;
    const int g = grid_for(total);
    if (HW <= 64 && C >= 64 && sC == HW && frames <= 65535) {
        const dim3 grid((C + 63) / 64, frames);
        if (sd == CPCSV_F32 && dd == CPCSV_F32) hipLaunchKernelGGL((nhwc_to_planar_tiled_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, T, sB, sT, C, HW, Cs);
        else if (sd == CPCSV_F32) hipLaunchKernelGGL((nhwc_to_planar_tiled_kernel<float, bf16_t>), grid, dim3(256), 0, s, (const float*)src, (bf16_t*)dst, T, sB, sT, C, HW, Cs);
        else if (dd == CPCSV_F32) hipLaunchKernelGGL((nhwc_to_planar_tiled_kernel<bf16_t, float>), grid, dim3(256), 0, s, (const bf16_t*)src, (float*)dst, T, sB, sT, C, HW, Cs);
        else hipLaunchKernelGGL((nhwc_to_planar_tiled_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, T, sB, sT, C, HW, Cs);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    if (sd == CPCSV_F32 && dd == CPCSV_F32) hipLaunchKernelGGL((nhwc_to_planar_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, total, T, sB, sT, sC, C, HW, Cs);
    else if (sd == CPCSV_F32) hipLaunchKernelGGL((nhwc_to_planar_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, total, T, sB, sT, sC, C, HW, Cs);
    else if (dd == CPCSV_F32) hipLaunchKernelGGL((nhwc_to_planar_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, total, T, sB, sT, sC, C, HW, Cs);
    else hipLaunchKernelGGL((nhwc_to_planar_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, total, T, sB, sT, sC, C, HW, Cs);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_ingest_u8(const void* src, float* planar, void* nhwc, int nhwc_dtype, int frames, int T, long sB,
                               long sT, long sC, int C, int HW, int Cs, const float* mean, const float* stdv, void* stream) {
    if (!src || (!planar && !nhwc) || !mean || !stdv || frames <= 0 || T <= 0 || C <= 0 || C > Cs || (nhwc && Cs % 8)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const long npix = (long)frames * HW;
    const int g = grid_for(npix);
    if (nhwc_dtype == CPCSV_BF16) hipLaunchKernelGGL(ingest_u8_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const uint8_t*)src, planar, (bf16_t*)nhwc, npix, T, sB, sT, sC, C, HW, Cs, mean, stdv);
    else hipLaunchKernelGGL(ingest_u8_kernel<float>, dim3(g), dim3(256), 0, s, (const uint8_t*)src, planar, (float*)nhwc, npix, T, sB, sT, sC, C, HW, Cs, mean, stdv);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_copy2d(const void* src, int sd, long lds, int scol0, void* dst, int dd, long ldd, int dcol0,
                            long rows, int cols, int accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (rows <= 0 || cols <= 0) return 0;
    const int g = grid_for(rows * (accumulate == 2 ? ldd : (long)cols));
    if (sd == CPCSV_F32 && dd == CPCSV_F32) hipLaunchKernelGGL((copy2d_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, lds, scol0, (float*)dst, ldd, dcol0, rows, cols, accumulate);
    else if (sd == CPCSV_F32) hipLaunchKernelGGL((copy2d_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, lds, scol0, (bf16_t*)dst, ldd, dcol0, rows, cols, accumulate);
    else if (dd == CPCSV_F32) hipLaunchKernelGGL((copy2d_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, lds, scol0, (float*)dst, ldd, dcol0, rows, cols, accumulate);
    else hipLaunchKernelGGL((copy2d_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, lds, scol0, (bf16_t*)dst, ldd, dcol0, rows, cols, accumulate);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_concat_pad(const float* s0, int w0, const float* s1, int w1, const float* s2, int w2, const float* s3,
                                int w3, int nsrc, void* dst, int ddtype, long rows, int ldd, void* stream) {
    if (nsrc < 1 || nsrc > 4 || !dst) return -1001;
    ConcatSrc src;
    const float* ps[4] = {s0, s1, s2, s3};
    const int ws[4] = {w0, w1, w2, w3};
    int col = 0;
    for (int k = 0; k < 4; ++k) { src.p[k] = ps[k]; src.w[k] = k < nsrc ? ws[k] : 0; src.col0[k] = col; col += src.w[k]; }
    src.n = nsrc;
    if (col > ldd) return -1002;
    hipStream_t s = (hipStream_t)stream;
    const int g = grid_for(rows * ldd);
    if (ddtype == CPCSV_BF16) hipLaunchKernelGGL(concat_pad_kernel<bf16_t>, dim3(g), dim3(256), 0, s, src, (bf16_t*)dst, rows, ldd);
    else hipLaunchKernelGGL(concat_pad_kernel<float>, dim3(g), dim3(256), 0, s, src, (float*)dst, rows, ldd);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
// The batch preparation of a training step (reference trainer.py:254-264,287-288,303-304) in ONE launch: the motion inputs (text
// concatenated with the labels), contiguous copies of the content inputs, the per-story label presence and mean text - six torch
// launches (2 cat, 2 mean, gt, cast) plus three strided copies at the very head of every step before.
struct BatchPrep {
    const float* im_desc; long ld_imd;      // [IM][ld >= td]
    const float* im_lab;                    // [IM][L]
    const float* im_cont; long ld_imc;      // [IM][T][ld >= td]
    const float* st_desc; long ld_std;      // [ST][T][ld >= td]
    const float* st_lab;                    // [ST][T][L]
    float* im_motion;                       // [IM][td + L]
    float* im_content;                      // [IM][T][td]
    float* st_motion;                       // [ST][T][td + L]
    float* st_text;                         // [ST][T][td]
    float* st_text_mean;                    // [ST][td]
    float* chars;                           // [ST][L]: (mean over T of the labels) > 0
    int IM, ST, T, td, L;
};
__global__ void batch_prep_kernel(const BatchPrep b) {
    const long n0 = (long)b.IM * (b.td + b.L), n1 = n0 + (long)b.IM * b.T * b.td, n2 = n1 + (long)b.ST * b.T * (b.td + b.L),
               n3 = n2 + (long)b.ST * b.T * b.td, n4 = n3 + (long)b.ST * b.td, n5 = n4 + (long)b.ST * b.L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n5; i += (long)gridDim.x * blockDim.x) {
        if (i < n0) {
            const int w = b.td + b.L, r = (int)(i / w), c = (int)(i - (long)r * w);
            b.im_motion[i] = c < b.td ? b.im_desc[r * b.ld_imd + c] : b.im_lab[(long)r * b.L + c - b.td];
        } else if (i < n1) {
            const long k = i - n0;
            const long r = k / b.td;
            const int c = (int)(k - r * b.td);
            b.im_content[k] = b.im_cont[r * b.ld_imc + c];
        } else if (i < n2) {
            const long k = i - n1;
            const int w = b.td + b.L;
            const long r = k / w;
            const int c = (int)(k - r * w);
            b.st_motion[k] = c < b.td ? b.st_desc[r * b.ld_std + c] : b.st_lab[r * b.L + c - b.td];
        } else if (i < n3) {
            const long k = i - n2;
            const long r = k / b.td;
            const int c = (int)(k - r * b.td);
            b.st_text[k] = b.st_desc[r * b.ld_std + c];
        } else if (i < n4) {
            const long k = i - n3;
            const int s_ = (int)(k / b.td), c = (int)(k - (long)s_ * b.td);
            float a = 0.f;
            for (int t = 0; t < b.T; ++t) a += b.st_desc[((long)s_ * b.T + t) * b.ld_std + c];
            b.st_text_mean[k] = a / (float)b.T;                      // torch: sum / count
        } else {
            const long k = i - n4;
            const int s_ = (int)(k / b.L), c = (int)(k - (long)s_ * b.L);
            float a = 0.f;
            for (int t = 0; t < b.T; ++t) a += b.st_lab[((long)s_ * b.T + t) * b.L + c];
            b.chars[k] = (a / (float)b.T) > 0.f ? 1.f : 0.f;
        }
    }
}
extern "C" int cpcsv_batch_prep(const float* im_desc, long ld_imd, const float* im_lab, const float* im_cont, long ld_imc, const float* st_desc,
                                long ld_std, const float* st_lab, float* im_motion, float* im_content, float* st_motion, float* st_text,
                                float* st_text_mean, float* chars, int IM, int ST, int T, int td, int L, void* stream) {
    if (!im_desc || !im_lab || !im_cont || !st_desc || !st_lab || !im_motion || !im_content || !st_motion || !st_text || !st_text_mean || !chars) return -1001;
    if (IM < 1 || ST < 1 || T < 1 || td < 1 || L < 1 || ld_imd < td || ld_imc < td || ld_std < td) return -1002;
    BatchPrep b{im_desc, ld_imd, im_lab, im_cont, ld_imc, st_desc, ld_std, st_lab, im_motion, im_content, st_motion, st_text, st_text_mean,
                chars, IM, ST, T, td, L};
    const long n = (long)IM * (td + L) + (long)IM * T * td + (long)ST * T * (td + L) + (long)ST * T * td + (long)ST * td + (long)ST * L;
    hipLaunchKernelGGL(batch_prep_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, b);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_cond_concat(const void* feat, const float* cond, void* out, int dtype, int N, int P, int C,
                                 int Cs_f, int E, int Cs_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (Cs_f + E > Cs_out) return -1001;
    const long total = (long)N * P * Cs_out;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(cond_concat_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)feat, cond, (bf16_t*)out, total, P, C, Cs_f, E, Cs_out);
    else hipLaunchKernelGGL(cond_concat_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)feat, cond, (float*)out, total, P, C, Cs_f, E, Cs_out);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_cond_triplet(const void* feat, const float* cond, void* out, int dtype, int N, int P, int C, int Cs_f,
                                  int E, int Cs_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!feat || !cond || !out || N < 2 || Cs_f + E > Cs_out) return -1001;
    const long total = (3L * N - 1) * P * Cs_out;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(cond_triplet_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)feat, cond, (bf16_t*)out, total, N, P, C, Cs_f, E, Cs_out);
    else hipLaunchKernelGGL(cond_triplet_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)feat, cond, (float*)out, total, N, P, C, Cs_f, E, Cs_out);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_cond_triplet_bwd(const void* dout, void* dfeat, int dtype, int N, int P, int C, int Cs_f, int Cs_out,
                                      void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!dout || !dfeat || N < 2) return -1001;
    const long total = 2L * N * P * Cs_f;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(cond_triplet_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dout, (bf16_t*)dfeat, total, N, P, C, Cs_f, Cs_out);
    else hipLaunchKernelGGL(cond_triplet_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dout, (float*)dfeat, total, N, P, C, Cs_f, Cs_out);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_im2col(const void* x, void* out, int dtype, int F, int H, int W, int Cs, int C, int k, int s, int p, int ld,
                            int adjoint, void* stream) {
    if (!x || !out || k < 1 || s < 1 || ld < C * k * k) return -1001;
    hipStream_t st = (hipStream_t)stream;
    const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    if (!adjoint) {
        const long total = (long)F * OH * OW * ld;
        if (dtype == CPCSV_BF16) hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, total, H, W, Cs, C, k, s, p, OH, OW, ld);
        else hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (float*)out, total, H, W, Cs, C, k, s, p, OH, OW, ld);
    } else {      // x = dcol [F*OH*OW][ld], out = dx [F][H][W][Cs]
        const long total = (long)F * H * W * Cs;
        if (dtype == CPCSV_BF16) hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, total, H, W, Cs, C, k, s, p, OH, OW, ld);
        else hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x, (float*)out, total, H, W, Cs, C, k, s, p, OH, OW, ld);
    }
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_mean_t(const void* in, void* out, int dtype, int N, int T, long inner, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * inner;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(mean_t_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, total, T, inner);
    else hipLaunchKernelGGL(mean_t_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)in, (float*)out, total, T, inner);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_mean_t_bwd(const void* dout, void* din, int dtype, int N, int T, long inner, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)N * T * inner;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(mean_t_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dout, (bf16_t*)din, total, T, inner);
    else hipLaunchKernelGGL(mean_t_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dout, (float*)din, total, T, inner);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
// Zero a buffer with this library's own launch (16-byte stores, byte stores for an unaligned head / tail) instead of hipMemsetAsync:
// the runtime's memset is a __amd_rocclr_fillBufferAligned launch of its own - up to three per call for a buffer with a ragged
// size - and shows up as a foreign node in every captured graph (14 of them per step in round 4's trace).
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ p, long head, long nvec, long tail) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    u32x4* v = reinterpret_cast<u32x4*>(p + head);
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (long i = tid; i < nvec; i += stride) v[i] = z;
    if (tid < head) p[tid] = 0;
    if (tid < tail) p[head + nvec * 16 + tid] = 0;
}
extern "C" int cpcsv_fill_zero(void* p, long bytes, void* stream) {
    if (bytes <= 0) return 0;
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    long head = (long)((16 - (a & 15)) & 15);
    if (head > bytes) head = bytes;
    const long nvec = (bytes - head) / 16, tail = bytes - head - nvec * 16;
    long blocks = (nvec + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, head, nvec, tail);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
extern "C" int cpcsv_scale_by(const void* x, void* y, int dtype, const float* alpha, float mult, long n, int accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CPCSV_BF16) hipLaunchKernelGGL(scale_by_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, alpha, mult, n, accumulate);
    else hipLaunchKernelGGL(scale_by_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)x, (float*)y, alpha, mult, n, accumulate);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
