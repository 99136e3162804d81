// gemm.hip — MFMA gather-GEMMs for gfx950 (MI355X).
//
// One LDS-tiled kernel family does every dense contraction of the CP-CSV training step:
//   * gemm_nt  : C[m][n] = sum_tap sum_c A[pix(m,tap)][c] * B[n][tap*Cs+c]
//                Linear / conv forward, conv dgrad, transposed-conv dgrad phases,
//                nearest-x2 upsample folded into the gather (fwd) or into the epilogue (dgrad).
//   * wgrad_tn : dW[n][tap*Cs+c] += sum_m dY[m][n] * X[pix(m,tap)][c]   (reduction over pixels)
//
// Design (MI355X-first):
//   - 256 threads = 4 wavefronts of 64; MFMA 16x16x32 bf16 (fp32 accumulate) or the exact
//     16x16x4 f32 MFMA for the fp32-parity mode. Both dtypes share ONE byte layout: an LDS row
//     is KC 16-byte chunks (128 B) + 16 B pad (144 B stride -> conflict-free ds_read_b128), a lane
//     reads chunk (lane>>4) of row (lane&15); that is a whole bf16 fragment or 4 f32 k-steps.
//   - register-staged global->LDS with the loads of tile t+1 issued before the MFMAs of tile t
//     (guide T14); zero padding, stride and the upsample are predicates/shifts on the gather, so
//     no im2col or upsampled tensor is ever materialised.
//   - BatchNorm batch statistics come out of the epilogue as per-block column partials (no
//     atomics, deterministic), so the conv output is never re-read for the stats.
#include "common.h"
#include "../../include/cpcsv_hip.h"

namespace {

constexpr int NTHREADS = 256;
constexpr int KC = 8;                    // 16-byte chunks per LDS row per K tile
constexpr int LDS_ROW = KC * 16 + 16;    // bytes

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * LDS_ROW + chunk * 16; }

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                    __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
    }
};

// bijective XCD-aware remap (guide T1): consecutive logical tiles land on the same XCD's L2
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ------------------------------------------------------------------------------------------
// NT gather GEMM — direct-to-LDS staging (global_load_lds_dwordx4), double-buffered, one barrier
// per K tile. LDS image: rows of 128 B (8 chunks of 16 B), chunk q of row r lives in slot
// q ^ ((r>>1)&7)  (XOR swizzle -> conflict-free ds_read_b128 without padding, which the lane-linear
// LDS-DMA destination forbids). A wave instruction fills 8 rows (64 lanes x 16 B); rows that are
// outside the problem, taps that fall into the zero padding and channel tails read a zero page, so
// every lane always writes its slot and no predication or LDS clearing is needed.
// ------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) const unsigned int g_zero_page[64] = {0};

__device__ __forceinline__ int lds_sw(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, int BM, int BN, int MI, int NI, int WGN>
__device__ __forceinline__ void mma_tile_sw(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane,
                                            f32x4 (&acc)[MI][NI]) {
    constexpr int WM = MI * 16, WN = NI * 16;
#pragma unroll
    for (int s = 0; s < KC / 4; ++s) {
        u32x4 a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const u32x4*>(As + lds_sw(wm * WM + i * 16 + (lane & 15), s * 4 + (lane >> 4)));
#pragma unroll
        for (int j = 0; j < NI; ++j)
            b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_sw(wn * WN + j * 16 + (lane & 15), s * 4 + (lane >> 4)));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
    }
}

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(const cpcsv_gemm_desc d) {
    constexpr int EPC = elem<T>::per16;
    constexpr int BK = KC * EPC;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    constexpr int GA = BM / 8, GB = BN / 8;            // 8-row groups (one wave instruction each)
    constexpr int A_IT = (GA + 3) / 4, B_IT = (GB + 3) / 4;
    static_assert(WGM * WGN == 4 && BM % 16 == 0 && BN % 16 == 0, "tile");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction, but only readfirstlane makes it PROVABLY so: the LDS-DMA base goes through M0,
    // and a base the compiler thinks is divergent gets a waterfall loop around every global_load_lds (guide T20)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_n = (d.N + BN - 1) / BN;
    const bool phased = d.nphases > 1;
    const int nph = phased ? d.nphases : 1;
    // logical block id = ((tile_m * phases) + phase) * tiles_n + tile_n: the parity phases of one M tile (which read
    // the same input pixels) and its N tiles (same A panel) are neighbours, and the XCD remap keeps neighbours on
    // one XCD's L2
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_n, ph, tile_m;
    if (d.order_m_fast) {
        // weight-heavy layers (4x4 / 8x8 maps with thousands of channels): the M tiles that share one B panel are
        // neighbours instead, so the panel is fetched into one XCD's L2 once rather than once per M tile
        const int tiles_m = (d.M + BM - 1) / BM;
        tile_m = bid % tiles_m;
        ph = (bid / tiles_m) % nph;
        tile_n = bid / (tiles_m * nph);
    } else {
        tile_n = bid % tiles_n;
        ph = (bid / tiles_n) % nph;
        tile_m = bid / (tiles_n * nph);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // split-K slice = blockIdx.y
    const int tap0 = phased ? d.ph_tap0[ph] : 0;
    const int ntaps = phased ? d.ph_ntaps[ph] : d.ntaps;
    const int ooy = phased ? d.ph_ooy[ph] : d.ooy, oox = phased ? d.ph_oox[ph] : d.oox;
    const int ctiles = (d.Cs + BK - 1) / BK;
    const int nk_all = ntaps * ctiles;
    int kt0 = 0, kt1 = nk_all;
    if (d.splitk > 1) {
        const int per = (nk_all + d.splitk - 1) / d.splitk;
        kt0 = blockIdx.y * per;
        kt1 = kt0 + per < nk_all ? kt0 + per : nk_all;
        if (kt0 > kt1) kt0 = kt1;          // an empty slice still writes its (zero) partial tile: slabs are never cleared
    }

    const T* __restrict__ A = reinterpret_cast<const T*>(d.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(d.B);
    const T* zp = reinterpret_cast<const T*>(g_zero_page);

    // ---- per-lane rows: wave w stages groups w, w+4, ... ; lane -> row (lane>>3), LDS slot (lane&7) ----
    const int lrow = lane >> 3, slot = lane & 7;
    int a_pix0[A_IT], a_yx[A_IT], a_chunk[A_IT];       // pixel base, packed (y*sy, x*sx), source chunk of this slot
    bool a_ok[A_IT];
    const int BH = d.IH << d.up_shift, BW = d.IW << d.up_shift;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int g = wave + 4 * it;
        const int row = g * 8 + lrow;
        const int m = m0 + row;
        a_ok[it] = (g < GA) && (m < d.M);
        int img, y, x;
        if (d.pool_rows) {
            const int sub = m & 3, mm = m >> 2, w2 = d.MW >> 1, h2 = d.MH >> 1;
            x = 2 * (mm % w2) + (sub & 1);
            y = 2 * ((mm / w2) % h2) + (sub >> 1);
            img = mm / (w2 * h2);
        } else {
            x = m % d.MW;
            y = (m / d.MW) % d.MH;
            img = m / (d.MW * d.MH);
        }
        a_pix0[it] = img * d.IH * d.IW;
        a_yx[it] = ((y * d.sy) << 16) | (x * d.sx);
        a_chunk[it] = slot ^ ((row >> 1) & 7);
    }
    long b_off[B_IT];
    int b_chunk[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int g = wave + 4 * it;
        const int row = g * 8 + lrow;
        const int n = n0 + row;
        b_ok[it] = (g < GB) && (n < d.N);
        b_chunk[it] = slot ^ ((row >> 1) & 7);
        b_off[it] = (long)n * d.ldb + b_chunk[it] * EPC;
    }

    // ---- per-tap source pointers (all gather math once per tap) ----
    const T* a_ptr[A_IT];
    bool a_v[A_IT];
    int wtap_off = 0;
    auto set_tap = [&](int j) {
        const cpcsv_tap tap = d.taps[tap0 + j];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int iy = (a_yx[it] >> 16) + tap.oy, ix = (a_yx[it] & 0xffff) + tap.ox;
            a_v[it] = a_ok[it] && (unsigned)iy < (unsigned)BH && (unsigned)ix < (unsigned)BW;
            iy >>= d.up_shift;
            ix >>= d.up_shift;
            a_ptr[it] = A + ((long)(a_pix0[it] + iy * d.IW + ix) * d.Cs + a_chunk[it] * EPC);
        }
        wtap_off = tap.wtap * d.Cs;
    };
    // stage one K tile (channel tile ct of the current tap) into LDS buffer `buf`
    auto stage = [&](int ct, int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
        const int c0 = ct * BK;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int g = wave + 4 * it;
            if (g < GA) {
                const bool ok = a_v[it] && (c0 + a_chunk[it] * EPC < d.Cs);
                const T* p = ok ? a_ptr[it] + c0 : zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(base + g * 1024), 16, 0, 0);
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int g = wave + 4 * it;
            if (g < GB) {
                const bool ok = b_ok[it] && (c0 + b_chunk[it] * EPC < d.Cs);
                const T* p = ok ? B + b_off[it] + wtap_off + c0 : zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(base + BM * 128 + g * 1024), 16, 0, 0);
            }
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging cursor (tap pj, channel tile pct) runs one K tile ahead of the MFMAs
    int pj = kt0 / ctiles, pct = kt0 - pj * ctiles;
    if (kt0 < kt1) {
        set_tap(pj);
        stage(pct, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        if (kt + 1 < kt1) {
            if (++pct == ctiles) { pct = 0; set_tap(++pj); }
            stage(pct, cur ^ 1);
        }
        const unsigned char* base = smem + cur * TILE_BYTES;
        mma_tile_sw<T, BM, BN, MI, NI, WGN>(base, base + BM * 128, wm, wn, lane, acc);
        // LDS-DMA completion is tracked by vmcnt, ds_reads by lgkmcnt; a raw barrier with explicit counters is enough
        // for LDS hand-off inside the workgroup (no memory fences needed) and measured ~8 % faster than __syncthreads()
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }

    // ---- epilogue: raw fp32 accumulators -> LDS tile [rows][BN]; then a compact loop where a thread owns one
    // 16-byte OUTPUT chunk column: alpha / bias / activation / BN column partials / cast, and 16-byte stores of
    // whole row segments. (The MFMA layout gives a lane one column of four rows = 2-4 byte scattered stores, and
    // unrolling the activation 64x per lane made the kernel mostly epilogue code.)
    const int col_l = lane & 15, quad = lane >> 4;
    const bool split = d.splitk > 1;
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int cl = wn * WN + j * 16 + col_l;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int rl = wm * WM + i * 16 + quad * 4;
            if (d.pool_rows) {
                tile[(rl >> 2) * BN + cl] = acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[(rl + r) * BN + cl] = acc[i][j][r];
            }
        }
    }
    __syncthreads();
    const bool f32out = split || d.out_f32 || sizeof(T) == 4;     // element size of what is stored: 4 or 2 bytes
    const int epc = f32out ? 4 : 8;         // output elements per 16-byte chunk
    const int cpr = BN / epc;                                    // chunks per tile row; NTHREADS % cpr == 0
    const int ch = tid % cpr;                                    // this thread's chunk column (fixed)
    const int nb = n0 + ch * epc;
    const int rows_t = d.pool_rows ? BM / 4 : BM;
    const int ldo = split ? d.ldws : d.ldc;
    const float alpha = (d.alpha && !split) ? *d.alpha : 1.f;
    float bias8[8], cs[8], cq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bias8[e] = (d.bias && !split && e < epc && nb + e < d.N) ? d.bias[nb + e] : 0.f;
        cs[e] = cq[e] = 0.f;
    }
    if (nb < ldo) {
        unsigned char* out = reinterpret_cast<unsigned char*>(split ? (void*)(d.ws + (long)blockIdx.y * d.ws_rows * d.ldws) : d.C);
        for (int rl = tid / cpr; rl < rows_t; rl += NTHREADS / cpr) {
            long orow;
            if (d.pool_rows) {
                orow = (m0 >> 2) + rl;
                if (orow >= (d.M >> 2)) break;
            } else {
                const int m = m0 + rl;
                if (m >= d.M) break;
                orow = m;
                if (d.scatter) {
                    const int x = m % d.MW, y = (m / d.MW) % d.MH, img = m / (d.MW * d.MH);
                    orow = ((long)img * d.OH + (y * d.osy + ooy)) * d.OW + (x * d.osx + oox);
                }
            }
            const float* src = tile + rl * BN + ch * epc;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = e < epc ? src[e] : 0.f;
            if (!split) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e < epc) {
                        const bool real = nb + e < d.N;
                        float t = v[e] * alpha + bias8[e];
                        cs[e] += real ? t : 0.f;
                        cq[e] += real ? t * t : 0.f;
                        v[e] = real ? act_apply(t, d.act) : 0.f;      // channel pads of the output are zeros
                    }
                }
            }
            u32x4 pk;
            if (f32out) {
                pk = u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                *reinterpret_cast<u32x4*>(out + (orow * ldo + nb) * 4) = pk;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
                *reinterpret_cast<u32x4*>(out + (orow * ldo + nb) * 2) = pk;
            }
        }
    }

    if (d.stats && !split) {
        // column partials of this block: threads with the same chunk column combine through LDS
        __syncthreads();                                         // everyone is done reading the tile
        float* red = reinterpret_cast<float*>(smem);             // [NTHREADS][2*epc]
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < epc) { red[tid * 2 * epc + e] = cs[e]; red[tid * 2 * epc + epc + e] = cq[e]; }
        __syncthreads();
        if (tid < BN) {
            const int c_ch = tid / epc, c_e = tid - c_ch * epc;
            float sm = 0.f, q = 0.f;
            for (int t = c_ch; t < NTHREADS; t += cpr) { sm += red[t * 2 * epc + c_e]; q += red[t * 2 * epc + epc + c_e]; }
            const int n = n0 + tid;
            if (n < d.N) {
                const long part = phased ? (long)ph * (gridDim.x / (tiles_n * nph)) + tile_m : tile_m;   // one partial per (phase, M tile)
                d.stats[(part * 2 + 0) * d.ldstat + n] = sm;
                d.stats[(part * 2 + 1) * d.ldstat + n] = q;
            }
        }
    }
}

// second pass of a split-K GEMM: sum the K-slice slabs -> alpha, bias, act, cast, BN column partials.
// Block = EPI_ROWS output rows x 64 columns x 4 slab lanes (threadIdx.y): the slab sum is spread over the
// lanes and combined through LDS, then lane 0 owns one column (coalesced rows), so the column partials are plain.
constexpr int EPI_ROWS = 8;
template <typename T>
__global__ __launch_bounds__(NTHREADS) void gemm_epilogue_kernel(const float* __restrict__ ws, int ldws, int nslabs, void* C,
                                                                 int ldc, long rows, int N, const float* alpha_p,
                                                                 const float* __restrict__ bias, int act, float* stats,
                                                                 int ldstat, int out_f32) {
    __shared__ float part[4][EPI_ROWS][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + tx;
    const long r0 = (long)blockIdx.x * EPI_ROWS;
    const long slab = rows * ldws;
#pragma unroll
    for (int rr = 0; rr < EPI_ROWS; ++rr) {
        const long r = r0 + rr;
        float v = 0.f;
        if (r < rows && n < N)
            for (int k = ty; k < nslabs; k += 4) v += ws[k * slab + r * ldws + n];
        part[ty][rr][tx] = v;
    }
    __syncthreads();
    if (ty != 0 || n >= ldc) return;
    const bool pad = n >= N;                        // channel pads of the output are written as zeros
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const float b = (bias && !pad) ? bias[n] : 0.f;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int rr = 0; rr < EPI_ROWS; ++rr) {
        const long r = r0 + rr;
        if (r >= rows) break;
        float v = ((part[0][rr][tx] + part[1][rr][tx]) + (part[2][rr][tx] + part[3][rr][tx])) * alpha + b;
        s += v;
        q += v * v;
        v = pad ? 0.f : act_apply(v, act);
        if (out_f32) reinterpret_cast<float*>(C)[r * ldc + n] = v;
        else elem<T>::st(reinterpret_cast<T*>(C) + r * ldc + n, v);
    }
    if (stats && !pad) {
        stats[((long)blockIdx.x * 2 + 0) * ldstat + n] = s;
        stats[((long)blockIdx.x * 2 + 1) * ldstat + n] = q;
    }
}

// ------------------------------------------------------------------------------------------
// TN weight-gradient GEMM: both operands are pixel-major in HBM; the loader transposes them
// into the same k-contiguous LDS image the NT kernel uses.
// ------------------------------------------------------------------------------------------
// k-contiguous LDS image of the wgrad operands: row = channel, 144-B stride, 16-B chunk q of row r in
// slot q ^ ((r / EPC) & 7). The XOR spreads the transposing stores (lanes own channel groups EPC rows
// apart, which all alias to one bank in a linear image) over 8 bank groups.
template <typename T>
__device__ __forceinline__ int wg_off(int row, int kbyte) {
    constexpr int EPC = elem<T>::per16;
    const int chunk = (kbyte >> 4) ^ ((row / EPC) & 7);
    return row * LDS_ROW + (chunk << 4) + (kbyte & 15);
}

template <typename T, int ITERS>
__device__ __forceinline__ void lds_store_transposed(unsigned char* S, const u32x4 (&r)[ITERS], int og, int mgrp) {
    constexpr int EPC = elem<T>::per16;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int ml = mgrp * ITERS + it;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<uint32_t*>(S + wg_off<T>(og * EPC + e, ml * 4)) = r[it][e];
        }
    } else if constexpr (ITERS % 2 == 0) {
        // interleave pixel pairs so one 4-byte store carries (m, m+1) of one channel
#pragma unroll
        for (int it = 0; it < ITERS; it += 2) {
            const int ml = mgrp * ITERS + it;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t a = r[it][w], b = r[it + 1][w];
                const uint32_t lo = (a & 0xffffu) | (b << 16);
                const uint32_t hi = (a >> 16) | (b & 0xffff0000u);
                *reinterpret_cast<uint32_t*>(S + wg_off<T>(og * EPC + 2 * w, ml * 2)) = lo;
                *reinterpret_cast<uint32_t*>(S + wg_off<T>(og * EPC + 2 * w + 1, ml * 2)) = hi;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int ml = mgrp * ITERS + it;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                *reinterpret_cast<uint16_t*>(S + wg_off<T>(og * EPC + 2 * w, ml * 2)) = (uint16_t)(r[it][w] & 0xffffu);
                *reinterpret_cast<uint16_t*>(S + wg_off<T>(og * EPC + 2 * w + 1, ml * 2)) = (uint16_t)(r[it][w] >> 16);
            }
        }
    }
}

template <typename T, int BM, int BN, int MI, int NI, int WGN>
__device__ __forceinline__ void mma_tile_wg(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane,
                                            f32x4 (&acc)[MI][NI]) {
    constexpr int WM = MI * 16, WN = NI * 16;
#pragma unroll
    for (int s = 0; s < KC / 4; ++s) {
        u32x4 a[MI], b[NI];
        const int kb = (s * 4 + (lane >> 4)) * 16;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a[i] = *reinterpret_cast<const u32x4*>(As + wg_off<T>(wm * WM + i * 16 + (lane & 15), kb));
#pragma unroll
        for (int j = 0; j < NI; ++j)
            b[j] = *reinterpret_cast<const u32x4*>(Bs + wg_off<T>(wn * WN + j * 16 + (lane & 15), kb));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma<T>::run(a[i], b[j], acc[i][j]);
    }
}

template <typename T, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(NTHREADS) void wgrad_tn_kernel(const cpcsv_wgrad_desc d) {
    constexpr int EPC = elem<T>::per16;
    constexpr int BKM = KC * EPC;  // pixels per K tile
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int A_IT = BM * KC / NTHREADS, B_IT = BN * KC / NTHREADS;
    constexpr int OGA = BM / EPC, OGB = BN / EPC;
    static_assert(A_IT >= 1 && B_IT >= 1 && WGM * WGN == 4, "tile too small");

    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * LDS_ROW];
    unsigned char* As = smem;
    unsigned char* Bs = smem + BM * LDS_ROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_o = (d.N + BM - 1) / BM;
    const int tiles_c = (d.Cs + BN - 1) / BN;
    int bid = blockIdx.x;
    const int tile_o = bid % tiles_o; bid /= tiles_o;
    const int tile_c = bid % tiles_c; bid /= tiles_c;
    const int j = bid;  // tap
    const int o0 = tile_o * BM, c0 = tile_c * BN;
    const cpcsv_tap tap = d.taps[j];

    // pixel range of this split, aligned to the K tile
    long per = ((long)d.M + d.splits - 1) / d.splits;
    per = (per + BKM - 1) / BKM * BKM;
    const long mbeg = (long)blockIdx.y * per;
    const long mend = (mbeg + per < d.M) ? mbeg + per : d.M;
    if (mbeg >= mend) return;

    const T* __restrict__ dY = reinterpret_cast<const T*>(d.dY);
    const T* __restrict__ X = reinterpret_cast<const T*>(d.X);
    const int oga = tid % OGA, mga = tid / OGA;
    const int ogb = tid % OGB, mgb = tid / OGB;
    const int BH = d.IH << d.up_shift, BW = d.IW << d.up_shift;
    const bool a_cok = (o0 + oga * EPC) < d.ldy;          // ldy is a multiple of 8 with zero pads
    const bool b_cok = (c0 + ogb * EPC) < d.Cs;

    // pixel coordinates of this thread's gathered rows advance by BKM per K tile: carry arithmetic
    // instead of three integer divisions per load
    const int plane = d.MH * d.MW;
    const int step_img = BKM / plane, step_y = (BKM % plane) / d.MW, step_x = BKM % d.MW;
    int px[B_IT], py[B_IT], pimg[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const long m = mbeg + mgb * B_IT + it;
        px[it] = (int)(m % d.MW);
        py[it] = (int)((m / d.MW) % d.MH);
        pimg[it] = (int)(m / plane);
    }

    // same carry arithmetic for the dY rows when they are gathered (sub-pixel upsample+conv)
    int qx[A_IT], qy[A_IT], qimg[A_IT];
    const int dy_oy = tap._pad & 15, dy_ox = tap._pad >> 4;
    if (d.dy_gather) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const long m = mbeg + mga * A_IT + it;
            qx[it] = (int)(m % d.MW);
            qy[it] = (int)((m / d.MW) % d.MH);
            qimg[it] = (int)(m / plane);
        }
    }

    u32x4 areg[A_IT], breg[B_IT];
    auto gload = [&](long mt) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const long m = mt + mga * A_IT + it;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (d.dy_gather) {
                if (m < mend && a_cok) {
                    const long row = ((long)qimg[it] * d.DYH + qy[it] * d.dy_sy + dy_oy) * d.DYW + qx[it] * d.dy_sx + dy_ox;
                    v = *reinterpret_cast<const u32x4*>(dY + row * d.ldy + o0 + oga * EPC);
                }
                qx[it] += step_x;
                if (qx[it] >= d.MW) { qx[it] -= d.MW; qy[it] += 1; }
                qy[it] += step_y;
                if (qy[it] >= d.MH) { qy[it] -= d.MH; qimg[it] += 1; }
                qimg[it] += step_img;
            } else if (m < mend && a_cok) {
                v = *reinterpret_cast<const u32x4*>(dY + m * d.ldy + o0 + oga * EPC);
            }
            areg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const long m = mt + mgb * B_IT + it;
            u32x4 v = {0u, 0u, 0u, 0u};
            int iy = py[it] * d.sy + tap.oy, ix = px[it] * d.sx + tap.ox;
            if (m < mend && b_cok && (unsigned)iy < (unsigned)BH && (unsigned)ix < (unsigned)BW) {
                iy >>= d.up_shift; ix >>= d.up_shift;
                v = *reinterpret_cast<const u32x4*>(X + (((long)pimg[it] * d.IH + iy) * d.IW + ix) * d.Cs + c0 + ogb * EPC);
            }
            breg[it] = v;
            // advance to the next K tile
            px[it] += step_x;
            if (px[it] >= d.MW) { px[it] -= d.MW; py[it] += 1; }
            py[it] += step_y;
            if (py[it] >= d.MH) { py[it] -= d.MH; pimg[it] += 1; }
            pimg[it] += step_img;
        }
    };
    auto lstore = [&]() {
        lds_store_transposed<T, A_IT>(As, areg, oga, mga);
        lds_store_transposed<T, B_IT>(Bs, breg, ogb, mgb);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(mbeg);
    lstore();
    __syncthreads();
    for (long mt = mbeg; mt < mend; mt += BKM) {
        const bool more = mt + BKM < mend;
        if (more) gload(mt + BKM);
        mma_tile_wg<T, BM, BN, MI, NI, WGN>(As, Bs, wm, wn, lane, acc);
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    const int col_l = lane & 15, quad = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int c = c0 + wn * WN + jj * 16 + col_l;
        if (c >= d.Cs) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + wm * WM + i * 16 + quad * 4 + r;
                if (o >= d.N) continue;
                float* p = d.dW + (long)o * d.lddw + (long)tap.wtap * d.Cs + c;
                if (d.splits > 1) atomicAdd(p, acc[i][jj][r]);
                else *p = acc[i][jj][r];          // single slice: dW is zero on entry, a store saves the read
            }
    }
}

// ------------------------------------------------------------------------------------------
// TN weight-gradient GEMM, bf16, LDS-DMA + transposing LDS reads (the gfx950-native form).
// Both operands are pixel-major in HBM ([pixel][channel]); global_load_lds drops 4 pixel rows x 256 B per wave
// instruction into an LDS image [64 pixels][128 channels] (no register staging, no transposing stores), and
// ds_read_b64_tr_b16 hands every lane 4 consecutive PIXELS of its channel: within a 16-lane group, lane i points at
// piece (row i/4, channels (i%4)*4..+3) of a [4 pixels][16 channels] block and receives column i of that block
// (verified on hardware with tools/probe/tr_probe.py). Two such reads = the 8-deep k fragment of a 16x16x32 MFMA.
// The 32-byte channel segments of a row are XOR-swizzled with the pixel row (through the DMA source address) so the
// 4 rows of a block sit in different banks.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ int wg2_off(int row, int ch) {          // byte offset of channel ch of pixel row `row`
    const int chunk = (ch >> 3) ^ ((row & 7) << 1);
    return row * 256 + (chunk << 4) + ((ch & 7) << 1);
}

template <int WGM, int WGN>
__global__ __launch_bounds__(NTHREADS) void wgrad_tn_dma_kernel(const cpcsv_wgrad_desc d) {
    constexpr int BM = 128, BN = 128, BKM = 64;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int STAGE = 2 * BKM * 256;                       // A image + B image of one K tile (32 KB)
    constexpr int IT = BKM / 4 / 4;                            // wave instructions per operand per wave (4 rows each)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_o = (d.N + BM - 1) / BM;
    const int tiles_c = (d.Cs + BN - 1) / BN;
    int bid = blockIdx.x;
    const int tile_o = bid % tiles_o; bid /= tiles_o;
    const int tile_c = bid % tiles_c; bid /= tiles_c;
    const int j = bid;
    const int o0 = tile_o * BM, c0 = tile_c * BN;
    const cpcsv_tap tap = d.taps[j];

    long per = ((long)d.M + d.splits - 1) / d.splits;
    per = (per + BKM - 1) / BKM * BKM;
    const long mbeg = (long)blockIdx.y * per;
    const long mend = (mbeg + per < d.M) ? mbeg + per : d.M;
    if (mbeg >= mend) return;

    const bf16_t* __restrict__ dY = reinterpret_cast<const bf16_t*>(d.dY);
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(d.X);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);
    const int BH = d.IH << d.up_shift, BW = d.IW << d.up_shift;

    // lane -> (pixel row within the 4-row group, 16-byte slot); the slot's SOURCE chunk carries the swizzle
    const int lrow = lane >> 4, slot = lane & 15;
    int prow[IT], pchunk[IT];
    int px[IT], py[IT], pimg[IT];
    const int plane = d.MH * d.MW;
    const int step_img = BKM / plane, step_y = (BKM % plane) / d.MW, step_x = BKM % d.MW;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        prow[it] = 4 * (wave + 4 * it) + lrow;                 // pixel row of the tile this lane stages
        pchunk[it] = slot ^ ((prow[it] & 7) << 1);             // channel chunk (8 channels) that lands in this slot
        const long m = mbeg + prow[it];
        px[it] = (int)(m % d.MW);
        py[it] = (int)((m / d.MW) % d.MH);
        pimg[it] = (int)(m / plane);
    }
    const int dy_oy = tap._pad & 15, dy_ox = tap._pad >> 4;

    auto stage = [&](long mt, int buf) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long m = mt + prow[it];
            const bool live = m < mend;
            // dY piece
            const int oc = o0 + pchunk[it] * 8;
            const bf16_t* pa = zp;
            if (live && oc < d.ldy) {
                long row = m;
                if (d.dy_gather) row = ((long)pimg[it] * d.DYH + py[it] * d.dy_sy + dy_oy) * d.DYW + px[it] * d.dy_sx + dy_ox;
                pa = dY + row * d.ldy + oc;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                             (__attribute__((address_space(3))) void*)(base + (wave + 4 * it) * 1024), 16, 0, 0);
            // gathered X piece
            const int cc = c0 + pchunk[it] * 8;
            const bf16_t* pb = zp;
            int iy = py[it] * d.sy + tap.oy, ix = px[it] * d.sx + tap.ox;
            if (live && cc < d.Cs && (unsigned)iy < (unsigned)BH && (unsigned)ix < (unsigned)BW) {
                iy >>= d.up_shift; ix >>= d.up_shift;
                pb = X + (((long)pimg[it] * d.IH + iy) * d.IW + ix) * d.Cs + cc;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb,
                                             (__attribute__((address_space(3))) void*)(base + BKM * 256 + (wave + 4 * it) * 1024), 16, 0, 0);
            // advance this lane's pixel to the next K tile
            px[it] += step_x;
            if (px[it] >= d.MW) { px[it] -= d.MW; py[it] += 1; }
            py[it] += step_y;
            if (py[it] >= d.MH) { py[it] -= d.MH; pimg[it] += 1; }
            pimg[it] += step_img;
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane piece of a [4 pixels][16 channels] block for the transposing read
    const int gi = lane & 15, q = lane >> 4;
    const int br = gi >> 2, bc = (gi & 3) * 4;

    stage(mbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (long mt = mbeg; mt < mend; mt += BKM) {
        if (mt + BKM < mend) stage(mt + BKM, cur ^ 1);
        const unsigned char* As = smem + cur * STAGE;
        const unsigned char* Bs = As + BKM * 256;
#pragma unroll
        for (int ks = 0; ks < BKM / 32; ++ks) {
            u32x4 a[MI], b[NI];
            const int p0 = ks * 32 + q * 8 + br;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int ch = wm * WM + i * 16 + bc;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(As + wg2_off(p0, ch)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(As + wg2_off(p0 + 4, ch)));
                a[i] = u32x4{((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                             ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16)};
            }
#pragma unroll
            for (int jj = 0; jj < NI; ++jj) {
                const int ch = wn * WN + jj * 16 + bc;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + wg2_off(p0, ch)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Bs + wg2_off(p0 + 4, ch)));
                b[jj] = u32x4{((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                              ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16)};
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jj = 0; jj < NI; ++jj) Mma<bf16_t>::run(a[i], b[jj], acc[i][jj]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    const int col_l = lane & 15, quad = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int c = c0 + wn * WN + jj * 16 + col_l;
        if (c >= d.Cs) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + wm * WM + i * 16 + quad * 4 + r;
                if (o >= d.N) continue;
                float* p = d.dW + (long)o * d.lddw + (long)tap.wtap * d.Cs + c;
                if (d.splits > 1) atomicAdd(p, acc[i][jj][r]);
                else *p = acc[i][jj][r];
            }
    }
}

// ---- host-side tile selection ---------------------------------------------------------------
enum NtCfg { NT_128x128, NT_128x64, NT_128x16, NT_64x128 };
inline NtCfg pick_nt(int M, int N) {
    if (N <= 16) return NT_128x16;
    if (N <= 64) return NT_128x64;
    if (M <= 64) return NT_64x128;
    return NT_128x128;
}

inline long out_rows(const cpcsv_gemm_desc& d) {
    if (d.scatter) return (long)(d.M / (d.MH * d.MW)) * d.OH * d.OW;
    return d.pool_rows ? d.M / 4 : d.M;
}

template <typename T, int BM, int BN, int WGM, int WGN>
int launch_nt(const cpcsv_gemm_desc& d, hipStream_t s) {
    const long tiles = (long)cdiv(d.M, BM) * cdiv(d.N, BN);
    const unsigned gy = d.splitk > 1 ? d.splitk : 1, gz = 1;
    const long phases = d.nphases > 1 ? d.nphases : 1;
    hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, WGM, WGN>), dim3((unsigned)(tiles * phases), gy, gz), dim3(NTHREADS), 0, s, d);
    CPCSV_CHECK_LAUNCH();
    if (d.splitk > 1) {
        const long rows = out_rows(d);
        hipLaunchKernelGGL(gemm_epilogue_kernel<T>, dim3((unsigned)cdiv(rows, EPI_ROWS), (unsigned)cdiv(d.ldc, 64)), dim3(NTHREADS), 0, s, d.ws, d.ldws,
                           d.splitk, d.C, d.ldc, rows, d.N, d.alpha, d.bias, d.act, d.stats, d.ldstat, d.out_f32);
        CPCSV_CHECK_LAUNCH();
    }
    return 0;
}
template <typename T>
int dispatch_nt(const cpcsv_gemm_desc& d, hipStream_t s) {
    switch (pick_nt(d.M, d.N)) {
        case NT_128x16: return launch_nt<T, 128, 16, 4, 1>(d, s);
        case NT_128x64: return launch_nt<T, 128, 64, 2, 2>(d, s);
        case NT_64x128: return launch_nt<T, 64, 128, 1, 4>(d, s);
        default: return launch_nt<T, 128, 128, 2, 2>(d, s);
    }
}

template <typename T, int BM, int BN, int WGM, int WGN>
int launch_wg(const cpcsv_wgrad_desc& d, hipStream_t s) {
    const long tiles = (long)cdiv(d.N, BM) * cdiv(d.Cs, BN) * d.ntaps;
    hipLaunchKernelGGL((wgrad_tn_kernel<T, BM, BN, WGM, WGN>), dim3((unsigned)tiles, (unsigned)d.splits), dim3(NTHREADS), 0, s, d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
inline int launch_wg_dma(const cpcsv_wgrad_desc& d, hipStream_t s) {
    const long tiles = (long)cdiv(d.N, 128) * cdiv(d.Cs, 128) * d.ntaps;
    hipLaunchKernelGGL((wgrad_tn_dma_kernel<2, 2>), dim3((unsigned)tiles, (unsigned)d.splits), dim3(NTHREADS), 0, s, d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int dispatch_wg(const cpcsv_wgrad_desc& d, hipStream_t s) {
    if (sizeof(T) == 2 && d.N > 64 && d.Cs > 64 && !d.legacy) return launch_wg_dma(d, s);
    const bool rows_small = d.N <= 32, cols_small = d.Cs <= 64;
    if (rows_small) return cols_small ? launch_wg<T, 32, 64, 1, 4>(d, s) : launch_wg<T, 32, 128, 1, 4>(d, s);
    if (d.N <= 64) return cols_small ? launch_wg<T, 64, 64, 2, 2>(d, s) : launch_wg<T, 64, 128, 1, 4>(d, s);
    return cols_small ? launch_wg<T, 128, 64, 2, 2>(d, s) : launch_wg<T, 128, 128, 2, 2>(d, s);
}

}  // namespace

extern "C" int cpcsv_gemm_mtile(const cpcsv_gemm_desc* d) {
    if (d->splitk > 1) return EPI_ROWS;
    return pick_nt(d->M, d->N) == NT_64x128 ? 64 : 128;
}

extern "C" int cpcsv_gemm_nt(const cpcsv_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return -1001;
    if (d->M <= 0 || d->N <= 0 || d->ntaps <= 0 || d->ntaps > CPCSV_MAX_TAPS) return -1002;
    if (d->Cs % 8 || d->ldb % 8) return -1003;
    if (d->pool_rows && (d->scatter || d->stats || (d->M & 3))) return -1004;
    if (d->splitk > 1 && (!d->ws || d->ldws < d->N || d->ws_rows <= 0)) return -1005;
    if (d->nphases > 4 || (d->nphases > 1 && !d->scatter)) return -1006;
    if (d->stats && d->scatter && d->splitk <= 1 && d->nphases <= 1) return -1007;   // partials are indexed by (phase, M tile)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == CPCSV_BF16 ? dispatch_nt<bf16_t>(*d, s) : dispatch_nt<float>(*d, s);
}

extern "C" int cpcsv_wgrad_tn(const cpcsv_wgrad_desc* d, void* stream) {
    if (!d || !d->dY || !d->X || !d->dW) return -1001;
    if (d->M <= 0 || d->N <= 0 || d->ntaps <= 0 || d->ntaps > CPCSV_MAX_TAPS || d->splits < 1) return -1002;
    if (d->Cs % 8 || d->ldy % 8) return -1003;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return d->dtype == CPCSV_BF16 ? dispatch_wg<bf16_t>(*d, s) : dispatch_wg<float>(*d, s);
}
