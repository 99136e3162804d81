// gemm.hip — MFMA gather-GEMMs for gfx950 (MI355X).
//
// One LDS-tiled kernel family does every dense contraction of the CP-CSV training step:
//   * gemm_nt  : C[m][n] = sum_tap sum_c A[pix(m,tap)][c] * B[n][tap*Cs+c]
//                Linear / conv forward, conv dgrad, transposed-conv dgrad phases,
//                nearest-x2 upsample folded into the gather (fwd) or into the epilogue (dgrad).
//   * wgrad_tn : dW[n][tap*Cs+c] += sum_m dY[m][n] * X[pix(m,tap)][c]   (reduction over pixels)
//
// Design (MI355X-first):
//   - operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no register stage): an LDS row is one 128-byte K slice of a
//     pixel / weight row (64 bf16 or 32 fp32 channels), 16-byte chunk q of row r in slot q ^ ((r >> 1) & 7) - conflict-free
//     ds_read_b128 without padding, which the lane-linear DMA destination forbids. A lane reads chunk (lane >> 4) of row
//     (lane & 15): a whole bf16 MFMA fragment (v_mfma_f32_16x16x32_bf16) or four k-steps of the exact v_mfma_f32_16x16x4_f32.
//   - tiles: 256x128 with 8 wavefronts and a 3-stage ring (144 KB of LDS) where that yields >= 256 blocks, else 128x128 / 128x64 /
//     128x16 / 64x128 with 4 wavefronts and a double buffer; split-K into fp32 slabs + one epilogue pass for few-tile / long-K shapes.
//   - zero padding, stride and the nearest-x2 upsample are predicates / shifts on per-lane running source pointers (lanes outside
//     the problem read a zero page), so no im2col or upsampled tensor is ever materialised; the patch-resident main loop
//     (conv_patch_kernel) stages a tile's input patch once per channel tile and serves all its taps from LDS.
//   - epilogue straight from the accumulators (operands swapped: a lane holds consecutive output columns): 1/sigma, bias,
//     activation, cast, and BatchNorm batch statistics as per-block column partials (no atomics, deterministic), so the conv
//     output is never re-read for the statistics. Row groups keep several reference calls of one layer apart inside one launch.
#include "common.h"
#include <cstdlib>
#include <type_traits>
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

// Weight-gradient launches with a pixel split: which (tile, split) a block of the ONE-DIMENSIONAL grid works on. Blocks go to the 8
// XCDs round-robin by linear id; with the two-dimensional grid (tiles, splits) and a tile count that is a multiple of 8 every XCD got
// tiles of EVERY split, so each of the 8 L2s pulled the whole dY and the whole X through the fabric (8 x the operands: the family's
// PMC traffic). Here XCD x works on the splits s = x (mod 8) - for 2 / 4 splits on a quarter / half of the tiles of split x / (8 / S) -
// tile after tile of one split, so a pixel range of the operands is fetched by ONE L2. Bijective; the host picks the 1-D grid only
// when the counts divide (launch_wg_dma / launch_wg).
__device__ __forceinline__ void wg_xcd_decode(unsigned lin, int tiles, int S, int& tile, int& split) {
    const unsigned xcd = lin & 7, idx = lin >> 3;
    if ((S & 7) == 0) {
        split = (int)(idx / (unsigned)tiles) * 8 + (int)xcd;
        tile = (int)(idx % (unsigned)tiles);
    } else {
        const int g = 8 / S, tpg = tiles / g;
        split = (int)xcd / g;
        tile = ((int)xcd % g) * tpg + (int)idx;
    }
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
// cache-policy bits of the LDS-DMA loads of the A (activation) / B (weight) operands of gemm_nt (gfx940 encoding: 1 = sc0,
// 2 = nt, 16 = sc1); compile-time experiment knobs, default 0
#ifndef CPCSV_A_AUX
#define CPCSV_A_AUX 0
#endif
#ifndef CPCSV_B_AUX
#define CPCSV_B_AUX 0
#endif

__device__ __forceinline__ int lds_sw(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// GEMM row m -> (image, y, x) of its MH x MW grid. Every map of this model is a power of two wide and high: shifts and masks then
// (three run-time integer divisions per row - ~40 instructions each - sat in front of every tile's first load and in its epilogue).
struct RowGrid {
    int MW, MH, lw, lh;
    bool p2;
};
__device__ __forceinline__ RowGrid row_grid(int MW, int MH) {
    RowGrid g;
    g.MW = MW; g.MH = MH;
    g.p2 = MW > 0 && MH > 0 && (MW & (MW - 1)) == 0 && (MH & (MH - 1)) == 0;
    g.lw = g.p2 ? __builtin_ctz(MW) : 0;
    g.lh = g.p2 ? __builtin_ctz(MH) : 0;
    return g;
}
__device__ __forceinline__ void row_to_pixel(const RowGrid& g, int m, int& img, int& y, int& x) {
    if (g.p2) {
        x = m & (g.MW - 1);
        y = (m >> g.lw) & (g.MH - 1);
        img = m >> (g.lw + g.lh);
    } else {
        x = m % g.MW;
        y = (m / g.MW) % g.MH;
        img = m / (g.MW * g.MH);
    }
}

// Row groups (cpcsv_gemm_desc.ngroups): M tiles never straddle a group boundary.
__host__ __device__ __forceinline__ int m_tiles_of(const cpcsv_gemm_desc& d, int bm) {
    if (d.ngroups <= 1) return (d.M + bm - 1) / bm;
    int t = 0;
    for (int g = 0; g < d.ngroups; ++g) t += (d.grow[g + 1] - d.grow[g] + bm - 1) / bm;
    return t;
}
__device__ __forceinline__ void tile_rows(const cpcsv_gemm_desc& d, int bm, int tile_m, int& grp, int& m0, int& mlim) {
    grp = 0; m0 = tile_m * bm; mlim = d.M;
    if (d.ngroups > 1) {
        int t = tile_m;
        for (int g = 0; g < d.ngroups; ++g) {
            const int tg = (d.grow[g + 1] - d.grow[g] + bm - 1) / bm;
            if (t < tg || g == d.ngroups - 1) { grp = g; m0 = d.grow[g] + t * bm; mlim = d.grow[g + 1]; break; }
            t -= tg;
        }
    }
}

#ifndef CPCSV_PROBE
// tools/nt_cycles.py / tools/nt_ablate.py / tools/nt_clock.py build variants: 1 = no output stores, 2 = no epilogue, 4 = no K loop,
// 8 = cycle counters, 16 = no MFMAs, 32 = no fragment reads, 64 = no LDS-DMA staging behind the prologue
#define CPCSV_PROBE 0
#endif
#if CPCSV_PROBE & 8
__device__ unsigned long long g_probe[8];      // [issue, mma, wait, total, blocks] cycle sums of wave 0 of every block
#define PROBE_T() __builtin_readcyclecounter()
#else
#define PROBE_T() 0ull
#endif

// One K tile of MFMAs. `mid(s)` is called between the LDS fragment reads of k-step s and its MFMAs: the caller issues
// the next tile's LDS-DMA loads there, so the time a wave spends blocked in the (back-pressured) vector-memory issue
// overlaps its own ds_read latency and the other waves' MFMAs instead of preceding the whole tile.
template <typename T, int BM, int BN, int MI, int NI, int WGN, typename Mid>
__device__ __forceinline__ void mma_tile_sw(const unsigned char* As, const unsigned char* Bs, int wm, int wn, int lane,
                                            f32x4 (&acc)[MI][NI], Mid&& mid) {
    constexpr int WM = MI * 16, WN = NI * 16;
#pragma unroll
    for (int s = 0; s < KC / 4; ++s) {
        u32x4 a[MI], b[NI];
        if (CPCSV_PROBE & 32) {                 // ablation: no fragment reads (the MFMAs run on whatever the lane number gives)
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = u32x4{(uint32_t)lane, (uint32_t)i, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = u32x4{(uint32_t)lane, (uint32_t)j, 0x3f803f80u, 0x3f803f80u};
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const u32x4*>(As + lds_sw(wm * WM + i * 16 + (lane & 15), s * 4 + (lane >> 4)));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_sw(wn * WN + j * 16 + (lane & 15), s * 4 + (lane >> 4)));
        }
        __builtin_amdgcn_sched_barrier(0);
        mid(s);
        __builtin_amdgcn_sched_barrier(0);
        if (CPCSV_PROBE & 16) {                 // ablation: no MFMAs (the fragments are still read)
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(b[j]));
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);      // operands swapped: see the epilogue
        }
    }
}

// LDS row -> output column inside the block tile. Row = wave column w (WN rows), group g (16*CG rows), MFMA tile
// jj, quad q, r; the column is w*WN + g*16*CG + q*4*CG + jj*4 + r.
template <int WN, int CG>
__device__ __forceinline__ int col_of(int row) {
    const int hi = row / (16 * CG) * (16 * CG), in = row - hi;
    const int jj = in >> 4, q = (in >> 2) & 3, r = in & 3;
    return hi + q * 4 * CG + jj * 4 + r;
}

// ---- epilogue straight from the accumulators (shared by the streaming and the patch-resident main loops). The MFMAs run with the
// operands swapped (weight fragment as the row operand), so the 16x16 result tile is C^T and a lane holds FOUR CONSECUTIVE output
// columns of ONE row:   m = m0 + wm*WM + i*16 + (lane&15),   n = n0 + wn*WN + j*16 + (lane>>4)*4 + r
// -> one 8-byte (bf16) / 16-byte (fp32, split-K slab) store per MFMA tile and lane, no LDS round trip. (The LDS-transposed
// epilogue this replaces cost ~6 us per block - more than 16 K tiles of MFMA work.)
template <typename T, int BM, int BN, int WGM, int WGN>
__device__ __forceinline__ void nt_epilogue(const cpcsv_gemm_desc& d, f32x4 (&acc)[BM / WGM / 16][BN / WGN / 16], unsigned char* smem, int tid,
                                            int lane, int wm, int wn, int grp, int m0, int mlim, int n0, bool phased, int ph, int nph,
                                            int tile_m, int tiles_n, int ooy, int oox) {
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int CG = NI >= 4 ? 4 : (NI >= 2 ? 2 : 1), NG = NI / CG;
    const int col_l = lane & 15, quad = lane >> 4;
    const bool split = d.splitk > 1;
    if (CPCSV_PROBE & 2) {
        float t = 0.f;
        for (int i = 0; i < MI; ++i)
            for (int j = 0; j < NI; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) reinterpret_cast<float*>(d.C)[0] = t;
        return;
    }
    const bool f32out = split || d.out_f32 || sizeof(T) == 4;     // element size of what is stored: 4 or 2 bytes
    const int ldo = split ? d.ldws : d.ldc;
    const float* alpha_p = d.ngroups > 1 ? d.galpha[grp] : d.alpha;
    const float alpha = (alpha_p && !split) ? *alpha_p : 1.f;
    const float slope = d.act == CPCSV_ACT_RELU ? 0.f : (d.act == CPCSV_ACT_LRELU ? 0.2f : 1.f);
    const bool smooth_act = d.act == CPCSV_ACT_TANH || d.act == CPCSV_ACT_SIGMOID;
    const bool want_stats = d.stats && !split;
    unsigned char* out = reinterpret_cast<unsigned char*>(split ? (void*)(d.ws + (long)blockIdx.y * d.ws_rows * d.ldws) : d.C);
    // accumulator (j = g*CG + jj, r) of this lane is output column  n0 + wn*WN + g*16*CG + quad*4*CG + jj*4 + r
    const RowGrid rg = row_grid(d.MW, d.MH);
    float bias4[NI][4], cs[NI][4], cq[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn * WN + (j / CG) * 16 * CG + quad * 4 * CG + (j % CG) * 4 + r;
            bias4[j][r] = (d.bias && !split && n < d.N) ? d.bias[n] : 0.f;
            cs[j][r] = cq[j][r] = 0.f;
        }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rl = wm * WM + i * 16 + col_l;
        const int m = m0 + rl;
        bool rowok = m < mlim;
        long orow = m;
        if (d.pool_rows) {                       // the 4 rows of a 2x2 block sit in 4 neighbouring lanes
            rowok = rowok && (col_l & 3) == 0;
            orow = m >> 2;
        } else if (d.scatter) {
            int x, y, img;
            row_to_pixel(rg, m, img, y, x);
            orow = ((long)img * d.OH + (y * d.osy + ooy)) * d.OW + (x * d.osx + oox);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int nb = n0 + wn * WN + g * 16 * CG + quad * 4 * CG;     // first of 4*CG consecutive columns
            float v[4 * CG];
#pragma unroll
            for (int jj = 0; jj < CG; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jj * 4 + r] = acc[i][g * CG + jj][r];
            if (d.pool_rows) {
#pragma unroll
                for (int e = 0; e < 4 * CG; ++e) {
                    v[e] += __shfl_xor(v[e], 1);
                    v[e] += __shfl_xor(v[e], 2);
                }
            }
            if (!split) {
                float add[4 * CG];
#pragma unroll
                for (int e = 0; e < 4 * CG; ++e) add[e] = 0.f;
                if (d.addend && rowok) {                 // per-element addend [GEMM row][ldadd] fp32 (ldadd % 4 == 0, >= N)
#pragma unroll
                    for (int q4 = 0; q4 < CG; ++q4)
                        if (nb + q4 * 4 < d.ldadd) {
                            const f32x4 a4 = *reinterpret_cast<const f32x4*>(d.addend + (long)m * d.ldadd + nb + q4 * 4);
                            add[q4 * 4] = a4[0]; add[q4 * 4 + 1] = a4[1]; add[q4 * 4 + 2] = a4[2]; add[q4 * 4 + 3] = a4[3];
                        }
                }
#pragma unroll
                for (int e = 0; e < 4 * CG; ++e) {
                    const int j = g * CG + (e >> 2), r = e & 3;
                    const bool real = nb + e < d.N;
                    const float t = v[e] * alpha + add[e] + bias4[j][r];
                    const float tt = (real && rowok) ? t : 0.f;
                    cs[j][r] += tt;
                    cq[j][r] += tt * tt;
                    const float a = smooth_act ? act_apply(t, d.act) : (t > 0.f ? t : t * slope);
                    v[e] = real ? a : 0.f;                       // channel pads of the output are zeros
                }
            }
            if (rowok && !(CPCSV_PROBE & 1)) {
                // ldo is a multiple of 8: every aligned group of 4 (fp32) / 8 (bf16) columns is inside or outside as a whole
                if (f32out) {
#pragma unroll
                    for (int q4 = 0; q4 < CG; ++q4)
                        if (nb + q4 * 4 < ldo)
                            *reinterpret_cast<u32x4*>(out + (orow * ldo + nb + q4 * 4) * 4) =
                                u32x4{__float_as_uint(v[q4 * 4]), __float_as_uint(v[q4 * 4 + 1]), __float_as_uint(v[q4 * 4 + 2]),
                                      __float_as_uint(v[q4 * 4 + 3])};
                } else if (CG == 1) {
                    if (nb < ldo) {
                        u32x2 pk;
                        pk[0] = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                        pk[1] = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<u32x2*>(out + (orow * ldo + nb) * 2) = pk;
                    }
                } else {
#pragma unroll
                    for (int q8 = 0; q8 < CG / 2; ++q8)
                        if (nb + q8 * 8 < ldo) {
                            u32x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                pk[e] = (uint32_t)f32_to_bf16(v[q8 * 8 + 2 * e]) | ((uint32_t)f32_to_bf16(v[q8 * 8 + 2 * e + 1]) << 16);
                            *reinterpret_cast<u32x4*>(out + (orow * ldo + nb + q8 * 8) * 2) = pk;
                        }
                }
            }
        }
    }

    if (want_stats) {
        // column partials of this block: butterfly over the 16 lanes that hold the same columns, then the WGM waves
        // stacked along M combine through LDS (the K loop's last barrier already retired every LDS read)
        float* red = reinterpret_cast<float*>(smem);             // [WGM][2][BN]
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = cs[j][r], q = cq[j][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    a += __shfl_xor(a, o);
                    q += __shfl_xor(q, o);
                }
                if (col_l == 0) {
                    const int c = wn * WN + (j / CG) * 16 * CG + quad * 4 * CG + (j % CG) * 4 + r;
                    red[(wm * 2 + 0) * BN + c] = a;
                    red[(wm * 2 + 1) * BN + c] = q;
                }
            }
        __syncthreads();
        if (tid < BN) {
            float sm = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) { sm += red[(w * 2 + 0) * BN + tid]; q += red[(w * 2 + 1) * BN + tid]; }
            const int n = n0 + tid;
            if (n < d.N) {
                const long part = phased ? (long)ph * (gridDim.x / (tiles_n * nph)) + tile_m : tile_m;   // one partial per (phase, M tile)
                d.stats[(part * 2 + 0) * d.ldstat + n] = sm;
                d.stats[(part * 2 + 1) * d.ldstat + n] = q;
            }
        }
    }
}

// NSTAGE LDS buffers; NSTAGE-1 K tiles of LDS-DMA are in flight while one is multiplied. NSTAGE=2 is the classic
// double buffer (64 KB, two blocks per CU); NSTAGE=4 (128 KB dynamic LDS, one block per CU) keeps 96 KB of loads in
// flight per CU, which is what hides the ~1 us loaded L2/HBM latency behind the ~0.2 us of MFMA work per K tile.
// The block is WGM x WGN waves (4 or 8); with 8 waves (256x128 tile, 3 stages, 144 KB) a SIMD holds two waves of the
// same block, so one wave's ds_read latency is covered by the other's MFMAs and the tile needs 0.75 of the L1 bytes
// per FLOP that 128x128 does (the 64 B/clk/CU vector-L1 path is what bounds the 128x128 tile at the MFMA rate).
// (Round 5: the same 256x128 tile with FOUR wavefronts of 128x64 - 96 instead of 128 KB of fragment reads per K tile, 212 VGPRs + 128
// accumulators, no spills, bit-identical - is slower in the step: 13.85 against 13.42 ms. One wavefront per SIMD has nobody to hide
// its ds_read latency behind.)
template <typename T, int BM, int BN, int WGM, int WGN, int NSTAGE, bool CK = false, bool BUF = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_nt_kernel(const cpcsv_gemm_desc d) {
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int EPC = elem<T>::per16;
    constexpr int BK = KC * EPC;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    constexpr int GA = BM / 8, GB = BN / 8;            // 8-row groups (one wave instruction each)
    constexpr int A_IT = (GA + NW - 1) / NW, B_IT = (GB + NW - 1) / NW;
    constexpr int CG = NI >= 4 ? 4 : (NI >= 2 ? 2 : 1), NG = NI / CG;      // column-tile groups of the epilogue
    static_assert((NW == 4 || NW == 8) && BM % 16 == 0 && BN % 16 == 0 && NI % CG == 0 && NT >= BN, "tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // NSTAGE * TILE_BYTES
    // every wave issues exactly LPS LDS-DMA instructions per stage (needed for the partial vmcnt waits)
    constexpr int LPS = A_IT + B_IT;
    static_assert(NSTAGE == 2 || (GA % NW == 0 && GB % NW == 0 && (NSTAGE - 2) * LPS < 64), "deep pipeline needs uniform stages");

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
        const int tiles_m = m_tiles_of(d, BM);
        tile_m = bid % tiles_m;
        ph = (bid / tiles_m) % nph;
        tile_n = bid / (tiles_m * nph);
    } else {
        tile_n = bid % tiles_n;
        ph = (bid / tiles_n) % nph;
        tile_m = bid / (tiles_n * nph);
    }
    int grp, m0, mlim;                                  // row group of this tile, its first row, the group's row limit
    tile_rows(d, BM, tile_m, grp, m0, mlim);
    const int n0 = tile_n * BN;

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

    // ---- per-lane rows: wave w stages groups w, w+NW, ... ; lane -> row (lane>>3), LDS slot (lane&7) ----
    const int lrow = lane >> 3, slot = lane & 7;
    int a_pix0[A_IT], a_yx[A_IT], a_chunk[A_IT];       // pixel base, packed (y*sy, x*sx), source chunk of this slot
    bool a_ok[A_IT];
    const int BH = d.IH << d.up_shift, BW = d.IW << d.up_shift;
    const RowGrid rgp = row_grid(d.MW, d.MH);
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int g = wave + NW * it;
        const int row = g * 8 + lrow;
        const int m = m0 + row;
        a_ok[it] = (g < GA) && (m < mlim);
        int img, y, x;
        if (d.pool_rows) {
            const int sub = m & 3, mm = m >> 2, w2 = d.MW >> 1, h2 = d.MH >> 1;
            x = 2 * (mm % w2) + (sub & 1);
            y = 2 * ((mm / w2) % h2) + (sub >> 1);
            img = mm / (w2 * h2);
        } else {
            row_to_pixel(rgp, m, img, y, x);
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
        const int g = wave + NW * it;
        const int row = g * 8 + lrow;
        // LDS row `row` holds output column n0 + col_of(row): within a group of CG MFMA column tiles the columns
        // are dealt out so that a lane's accumulators of the whole group are 4*CG CONSECUTIVE columns (epilogue)
        const int n = n0 + col_of<WN, CG>(row);
        b_ok[it] = (g < GB) && (n < d.N);
        b_chunk[it] = slot ^ ((row >> 1) & 7);
        // bcol_rows: column n = (K window n / bcol_rows) of weight row n % bcol_rows (cpcsv_gemm_desc.bcol_rows)
        b_off[it] = (d.bcol_rows ? (long)(n % d.bcol_rows) * d.ldb + (long)(n / d.bcol_rows) * d.bcol_koff : (long)n * d.ldb) + b_chunk[it] * EPC;
    }

    // ---- staging cursors. All gather math happens once per tap (set_tap); staging a K tile is then, per LDS-DMA instruction, the
    // load itself and next to nothing else. (Issuing the 8 loads of a tile used to take ~1200 cycles of address arithmetic and
    // branches - more than the tile's MFMAs - and these launches are bound by issue slots: 30 more VALU instructions per K tile
    // and wavefront cost 7-15 % of a launch, profiles/r06_experiments.txt.)
    //   pointer form (BUF = false): a 64-bit pointer per staged row group; a lane outside the problem (row tail, tap in the zero
    //     padding) points at the zero page with step 0, the others walk their channel run: one 64-bit add per group and K tile.
    //   buffer form (BUF = true; operands below 2 GB): buffer_load ... lds through a resource of 2^31 records. The lane keeps a
    //     32-bit byte offset per row group and tap - 0x80000000 when it is outside the problem: out of range, the hardware
    //     delivers ZEROS (tools/probe/buf_lds.hip) - and the channel tile lives in the SCALAR offset: per K tile one scalar add,
    //     no vector instruction at all.
    // Only the last channel tile of a tap can have chunks beyond Cs (Cs % BK != 0); those lanes are fixed for the whole kernel and
    // are sent to the zero page / out of range in that tile only.
    const unsigned char* const zpb = reinterpret_cast<const unsigned char*>(g_zero_page);
    constexpr int KSTEP = BK * (int)sizeof(T);
    constexpr unsigned OOBV = 0x80000000u;
    constexpr bool ck = CK;
    static_assert(BUF || !CK, "the channel-tiles-outer order exists in the buffer form only");
    const bool has_tail = (d.Cs % BK) != 0;
    const unsigned char* a_cur[A_IT];
    const unsigned char* b_cur[B_IT];
    int a_step[A_IT], b_step[B_IT];
    unsigned a_vo[A_IT], b_vo[B_IT];                   // buffer form: byte offsets of this lane's 16 bytes (A: of the current tap)
    unsigned sa = 0, sb = 0;                           // buffer form: wave-uniform byte offsets (channel tile; B: + the tap's K window)
    bool a_tail_ok[A_IT], b_tail_ok[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) a_tail_ok[it] = (ctiles - 1) * BK + a_chunk[it] * EPC < d.Cs;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) b_tail_ok[it] = (ctiles - 1) * BK + b_chunk[it] * EPC < d.Cs;
    // ---- channel tiles OUTER, taps inner (CK): consecutive K tiles read the SAME input lines shifted by one tap, so the 4 / 9 / 16
    // reads a conv makes of every input pixel hit in L2 instead of arriving a whole channel sweep apart - with taps outer the 32
    // tiles of an XCD pull ~1.5 MB per K tile through its 4 MB L2 and a line is gone before its next tap asks for it: PMC read
    // traffic of the big-map launches was the im2col volume (884 MB for up3's data gradient: 14 x its 63 MB map), the weight panels
    // went with it. The lane's offset of K tile (tap j, channel tile ct) is its pixel's offset or "out of range" by one bit of a
    // mask made once; the tap's offset (plus a bias that keeps it non-negative: the resource starts that far below the tensor) and
    // the channel tile are scalar. Needs up_shift == 0 (the offsets separate) - the host falls back to taps outer otherwise.
    // Same order as the patch-resident loop: bit-identical to it (tests/test_gpu_ops.py).
    unsigned a_bo[A_IT], a_vm[A_IT];
    unsigned a_bias = 0;
    if (ck) {
        int lo = 0;
        for (int j = 0; j < ntaps; ++j) {
            const cpcsv_tap tap = d.taps[tap0 + j];
            const int o = tap.oy * d.IW + tap.ox;
            lo = o < lo ? o : lo;
        }
        a_bias = (unsigned)(-lo) * (unsigned)d.Cs * (unsigned)sizeof(T);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int y0 = a_yx[it] >> 16, x0 = a_yx[it] & 0xffff;
            a_bo[it] = (unsigned)(((long)(a_pix0[it] + y0 * d.IW + x0) * d.Cs + a_chunk[it] * EPC) * (long)sizeof(T));
            unsigned m = 0;
            for (int j = 0; j < ntaps; ++j) {
                const cpcsv_tap tap = d.taps[tap0 + j];
                if (a_ok[it] && (unsigned)(y0 + tap.oy) < (unsigned)BH && (unsigned)(x0 + tap.ox) < (unsigned)BW) m |= 1u << j;
            }
            a_vm[it] = m;
        }
    }
    const auto rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(A)) - a_bias, 0,
                                                          (int)OOBV, 0x00020000);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(B)), 0, (int)OOBV,
                                                          0x00020000);
    if (BUF) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it) b_vo[it] = b_ok[it] ? (unsigned)(b_off[it] * (long)sizeof(T)) : OOBV;
    }
    auto set_tap = [&](int j, int ct) {                         // position the cursors at channel tile ct of tap j
        const cpcsv_tap tap = d.taps[tap0 + j];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int iy = (a_yx[it] >> 16) + tap.oy, ix = (a_yx[it] & 0xffff) + tap.ox;
            const bool v = a_ok[it] && (unsigned)iy < (unsigned)BH && (unsigned)ix < (unsigned)BW;
            iy >>= d.up_shift;
            ix >>= d.up_shift;
            const long e = (long)(a_pix0[it] + iy * d.IW + ix) * d.Cs + a_chunk[it] * EPC;
            if (BUF) {
                a_vo[it] = v ? (unsigned)(e * (long)sizeof(T)) : OOBV;
            } else {
                a_cur[it] = v ? reinterpret_cast<const unsigned char*>(A + e + ct * BK) : zpb;
                a_step[it] = v ? KSTEP : 0;
            }
        }
        const int wtap_off = tap.wtap * (d.wstride ? d.wstride : d.Cs) + ct * BK;
        if (BUF) {
            sa = (unsigned)(ct * KSTEP);
            sb = (unsigned)wtap_off * (unsigned)sizeof(T);
        } else {
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                b_cur[it] = b_ok[it] ? reinterpret_cast<const unsigned char*>(B + b_off[it] + wtap_off) : zpb;
                b_step[it] = b_ok[it] ? KSTEP : 0;
            }
        }
    };
    // (the taps' scalar offsets sit in lanes 0 .. ntaps-1 of two registers and are fetched with v_readlane: a scalar load of
    // d.taps[j] per K tile shares its wait counter with the LDS fragment reads)
    int tab_a = 0, tab_b = 0;
    if (ck && lane < ntaps) {
        const cpcsv_tap tap = d.taps[tap0 + lane];
        tab_a = (tap.oy * d.IW + tap.ox) * d.Cs * (int)sizeof(T) + (int)a_bias;
        tab_b = tap.wtap * (d.wstride ? d.wstride : d.Cs) * (int)sizeof(T);
    }
    auto position = [&](int j, int ct) {                        // CK: offsets of K tile (tap j, channel tile ct); nothing runs
        sa = (unsigned)__builtin_amdgcn_readlane(tab_a, j) + (unsigned)(ct * KSTEP);
        sb = (unsigned)__builtin_amdgcn_readlane(tab_b, j) + (unsigned)(ct * KSTEP);
        const unsigned bit = 1u << j;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) a_vo[it] = (a_vm[it] & bit) ? a_bo[it] : OOBV;
    };
    // stage the K tile under the cursors (channel tile ct of the current tap) into LDS buffer `buf`, advance them
    // (buffer form: the tail test is a wave-uniform BRANCH around two copies of the issue loop - as a select it was one v_cndmask in
    // front of every load of every K tile)
    auto stage_a = [&](int ct, int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
        const bool tail = has_tail && ct == ctiles - 1;
        if (BUF) {
            auto issue = [&](auto tl) {
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const int g = wave + NW * it;
                    if (GA % NW == 0 || g < GA) {
                        const unsigned vo = (decltype(tl)::value && !a_tail_ok[it]) ? OOBV : a_vo[it];
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(base + g * 1024), 16, vo, sa,
                                                                 0, CPCSV_A_AUX);
                    }
                }
            };
            if (tail) issue(std::true_type{});
            else issue(std::false_type{});
            if (!ck) sa += KSTEP;
            return;
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int g = wave + NW * it;
            if (GA % NW == 0 || g < GA) {
                const unsigned char* p = (tail && !a_tail_ok[it]) ? zpb : a_cur[it];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(base + g * 1024), 16, 0, CPCSV_A_AUX);
                a_cur[it] += a_step[it];
            }
        }
    };
    auto stage_b = [&](int ct, int buf) {
        unsigned char* base = smem + buf * TILE_BYTES;
        const bool tail = has_tail && ct == ctiles - 1;
        if (BUF) {
            auto issue = [&](auto tl) {
#pragma unroll
                for (int it = 0; it < B_IT; ++it) {
                    const int g = wave + NW * it;
                    if (GB % NW == 0 || g < GB) {
                        const unsigned vo = (decltype(tl)::value && !b_tail_ok[it]) ? OOBV : b_vo[it];
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(base + BM * 128 + g * 1024), 16,
                                                                 vo, sb, 0, CPCSV_B_AUX);
                    }
                }
            };
            if (tail) issue(std::true_type{});
            else issue(std::false_type{});
            if (!ck) sb += KSTEP;
            return;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int g = wave + NW * it;
            if (GB % NW == 0 || g < GB) {
                const unsigned char* p = (tail && !b_tail_ok[it]) ? zpb : b_cur[it];
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(base + BM * 128 + g * 1024), 16, 0, CPCSV_B_AUX);
                b_cur[it] += b_step[it];
            }
        }
    };
    auto stage = [&](int ct, int buf) {
        stage_a(ct, buf);
        stage_b(ct, buf);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#if CPCSV_PROBE & 8
    unsigned long long pr_issue = 0, pr_mma = 0, pr_wait = 0;
    const unsigned long long pr_t0 = PROBE_T();
    const unsigned long long pr_r0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz: cycles / realtime = the shader clock
#endif
    // staging cursor (tap pj, channel tile pct) runs NSTAGE-1 K tiles ahead of the MFMAs
    int pj, pct;
    if (ck) { pct = kt0 / ntaps; pj = kt0 - pct * ntaps; }
    else { pj = kt0 / ctiles; pct = kt0 - pj * ctiles; }
    auto first_tile = [&]() { if (ck) position(pj, pct); else set_tap(pj, pct); };
    auto next_tile = [&]() {                                    // move the staging cursor one K tile on
        if (ck) {
            if (++pj == ntaps) { pj = 0; ++pct; }
            position(pj, pct);
        } else if (++pct == ctiles) {
            pct = 0;
            set_tap(++pj, 0);
        }
    };
    if (CPCSV_PROBE & 4) kt1 = kt0;
    if (NSTAGE == 2) {
        if (kt0 < kt1) {
            first_tile();
            stage(pct, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            const unsigned long long t0 = PROBE_T();
            // double buffer: the next tile's loads go out FIRST - they have only this tile's MFMAs to land in
            // (issuing them between the k-steps, as the deeper pipeline does, measured 20 % slower here)
            if (kt + 1 < kt1 && !(CPCSV_PROBE & 64)) {
                next_tile();
                stage(pct, cur ^ 1);
            }
            const unsigned long long t1 = PROBE_T();
            const unsigned char* base = smem + cur * TILE_BYTES;
            mma_tile_sw<T, BM, BN, MI, NI, WGN>(base, base + BM * 128, wm, wn, lane, acc, [](int) {});
            const unsigned long long t2 = PROBE_T();
            // LDS-DMA completion is tracked by vmcnt, ds_reads by lgkmcnt; a raw barrier with explicit counters is enough
            // for LDS hand-off inside the workgroup (no memory fences needed) and measured ~8 % faster than __syncthreads()
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur ^= 1;
#if CPCSV_PROBE & 8
            const unsigned long long t3 = PROBE_T();
            pr_issue += t1 - t0; pr_mma += t2 - t1; pr_wait += t3 - t2;
#endif
        }
    } else {
        // prologue: up to NSTAGE-1 tiles on their way
        int issued = 0;
        for (; issued < NSTAGE - 1 && kt0 + issued < kt1; ++issued) {
            if (issued == 0) first_tile();
            else next_tile();
            stage(pct, issued);
        }
        // tile kt0 must have landed: at most issued-1 younger stages may still be in flight
        if (issued >= 3 && NSTAGE > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (issued >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0, fill = NSTAGE - 1;                    // buffer multiplied now / buffer the next stage goes to
        for (int kt = kt0; kt < kt1; ++kt) {
            const unsigned long long t0 = PROBE_T();
            const bool more = kt + NSTAGE - 1 < kt1;        // refill the buffer that was multiplied in the previous iteration
            if (more) next_tile();
            const unsigned long long t1 = PROBE_T();
            const unsigned char* base = smem + cur * TILE_BYTES;
            mma_tile_sw<T, BM, BN, MI, NI, WGN>(base, base + BM * 128, wm, wn, lane, acc, [&](int sk) {
                if (more && !(CPCSV_PROBE & 64)) {          // (ablation 64: nothing is staged behind the prologue)
                    if (sk == 0) stage_a(pct, fill);
                    else stage_b(pct, fill);
                }
            });
            const unsigned long long t2 = PROBE_T();
            // tile kt+1 must have landed before the barrier; the `ahead` stages after it may stay in flight
            const int ahead = kt1 - 2 - kt;                // stages younger than kt+1 that exist
            if (ahead >= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSTAGE - 2) * LPS) : "memory");
            else if (ahead == 1 && NSTAGE > 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            cur = cur + 1 == NSTAGE ? 0 : cur + 1;
            fill = fill + 1 == NSTAGE ? 0 : fill + 1;
#if CPCSV_PROBE & 8
            const unsigned long long t3 = PROBE_T();
            pr_issue += t1 - t0; pr_mma += t2 - t1; pr_wait += t3 - t2;
#endif
        }
    }

#if CPCSV_PROBE & 8
    if (tid == 0) {
        atomicAdd(&g_probe[0], pr_issue); atomicAdd(&g_probe[1], pr_mma); atomicAdd(&g_probe[2], pr_wait);
        atomicAdd(&g_probe[3], PROBE_T() - pr_t0); atomicAdd(&g_probe[4], 1ull);
        atomicAdd(&g_probe[5], (unsigned long long)(kt1 - kt0));
        atomicAdd(&g_probe[6], __builtin_amdgcn_s_memrealtime() - pr_r0);
    }
#endif
    nt_epilogue<T, BM, BN, WGM, WGN>(d, acc, smem, tid, lane, wm, wn, grp, m0, mlim, n0, phased, ph, nph, tile_m, tiles_n, ooy, oox);
}

// second pass of a split-K GEMM: sum the K-slice slabs -> alpha, bias, act, cast, BN column partials.
// Block = EPI_ROWS output rows x 64 columns x 4 slab lanes (threadIdx.y): the slab sum is spread over the
// lanes and combined through LDS, then lane 0 owns one column (coalesced rows), so the column partials are plain.
constexpr int EPI_ROWS = 32;
struct EpiGroups { int n; long row[5]; const float* alpha[4]; };     // output-row groups of a split-K launch (n <= 1: none)

// Block = EPI_ROWS (32) output rows x 64 columns; thread = FOUR consecutive columns (16-byte slab loads, one 8-byte bf16 / 16-byte
// fp32 store per row) of two rows, 16 row lanes. A thread sums all K slices of its elements itself, in the order the 8-row / four
// slab-lane form of rounds 1-3 used - p_j = sum over k = j, j+4, ... ascending, then (p0 + p1) + (p2 + p3) - so the outputs are
// bit-identical to it; that form read 4 bytes per thread and load, stored 2, left three quarters of the block idle behind a barrier
// and wrote one statistics partial per 8 rows (2.1 TB/s on the 8x8x1024 map, and four times the partial rows for bn_finalize).
template <typename T, bool SMOOTH>
__global__ __launch_bounds__(NTHREADS) void gemm_epilogue_kernel(const float* __restrict__ ws, int ldws, int nslabs, void* C,
                                                                 int ldc, long rows_all, int N, const float* alpha_p,
                                                                 const float* __restrict__ bias, int act, float* stats,
                                                                 int ldstat, int out_f32, EpiGroups eg, const float* __restrict__ addend,
                                                                 int ldadd) {
    __shared__ float red[2][4][64];                   // column sums / sums of squares of the four waves
    const int cx = threadIdx.x & 15, rl = threadIdx.x >> 4;          // 16 column chunks x 16 row lanes
    const int n0 = blockIdx.y * 64 + cx * 4;
    long r0 = (long)blockIdx.x * EPI_ROWS, rows = rows_all;
    int grp = 0;
    if (eg.n > 1) {                                   // row blocks never straddle a group: block b = block b - B_g of group g
        long b = blockIdx.x;
        for (int g = 0; g < eg.n; ++g) {
            const long bg = (eg.row[g + 1] - eg.row[g] + EPI_ROWS - 1) / EPI_ROWS;
            if (b < bg || g == eg.n - 1) { r0 = eg.row[g] + b * EPI_ROWS; rows = eg.row[g + 1]; alpha_p = eg.alpha[g]; grp = g; break; }
            b -= bg;
        }
    }
    const long slab = rows_all * ldws;
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const ActPl apl = act_pl(act);
    const bool colin = n0 < ldc;                      // ldc and ldws are multiples of 4 (8): a chunk is inside or outside as a whole
    float bs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bs[e] = (bias && n0 + e < N) ? bias[n0 + e] : 0.f;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < EPI_ROWS / 16; ++h) {
        const long r = r0 + rl + 16 * h;
        if (r >= rows || !colin) continue;
        f32x4 p[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (n0 < ldws) {
            const float* src = ws + r * ldws + n0;
            for (int k = 0; k < nslabs; k += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j < nslabs) p[j] += *reinterpret_cast<const f32x4*>(src + (long)(k + j) * slab);
            }
        }
        const f32x4 tot = (p[0] + p[1]) + (p[2] + p[3]);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool pad = n0 + e >= N;             // channel pads of the output are written as zeros
            float t = pad ? 0.f : tot[e] * alpha + bs[e];
            if (addend && !pad) t += addend[r * ldadd + n0 + e];
            s[e] += t;
            q[e] += t * t;
            v[e] = pad ? 0.f : act_apply_t<SMOOTH>(t, act, apl);
        }
        if (out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + r * ldc + n0) = f32x4{v[0], v[1], v[2], v[3]};
        else if (sizeof(T) == 2) {
            u32x2 pk;
            pk[0] = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            pk[1] = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(C) + r * ldc + n0) = pk;
        } else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + r * ldc + n0) = f32x4{v[0], v[1], v[2], v[3]};
    }
    if (!stats) return;
    // column partials of the block: the four row lanes of a wave by shuffle, the four waves through LDS (fixed order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] += __shfl_xor(s[e], 16); q[e] += __shfl_xor(q[e], 16);
        s[e] += __shfl_xor(s[e], 32); q[e] += __shfl_xor(q[e], 32);
    }
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][wave][cx * 4 + e] = s[e]; red[1][wave][cx * 4 + e] = q[e]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int n = blockIdx.y * 64 + threadIdx.x;
        if (n < N) {
            const float ss = (red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + (red[0][2][threadIdx.x] + red[0][3][threadIdx.x]);
            const float qq = (red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + (red[1][2][threadIdx.x] + red[1][3][threadIdx.x]);
            stats[((long)blockIdx.x * 2 + 0) * ldstat + n] = ss;
            stats[((long)blockIdx.x * 2 + 1) * ldstat + n] = qq;
        }
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
    int bid = blockIdx.x, split = blockIdx.y;
    if (gridDim.y == 1 && d.splits > 1) wg_xcd_decode(blockIdx.x, tiles_o * tiles_c * d.ntaps, d.splits, bid, split);
    const int tile_o = bid % tiles_o; bid /= tiles_o;
    const int tile_c = bid % tiles_c; bid /= tiles_c;
    const int j = bid;  // tap
    const int o0 = tile_o * BM, c0 = tile_c * BN;
    const cpcsv_tap tap = d.taps[j];

    // pixel range of this split, aligned to the K tile
    long per = ((long)d.M + d.splits - 1) / d.splits;
    per = (per + BKM - 1) / BKM * BKM;
    const long mbeg = (long)split * per;
    const long mend = (mbeg + per < d.M) ? mbeg + per : d.M;
    if (mbeg >= mend) return;

    const T* __restrict__ dY = reinterpret_cast<const T*>(d.dY);
    const T* __restrict__ X = reinterpret_cast<const T*>(d.X);
    const int oga = tid % OGA, mga = tid / OGA;
    const int ogb = tid % OGB, mgb = tid / OGB;
    const int BH = d.IH << d.up_shift, BW = d.IW << d.up_shift;
    const int ycol0 = j * d.dy_tapstride;                 // dy_tapstride: every tap reads its own column block of dY
    const bool a_cok = (o0 + oga * EPC) < (d.dy_tapstride ? d.dy_tapstride : d.ldy);          // ldy is a multiple of 8 with zero pads
    const bool b_cok = (c0 + ogb * EPC) < d.Cs;

    // pixel coordinates of this thread's gathered rows advance by BKM per K tile: carry arithmetic
    // instead of three integer divisions per load
    const int plane = d.MH * d.MW;
    const int step_img = BKM / plane, step_y = (BKM % plane) / d.MW, step_x = BKM % d.MW;
    const RowGrid rgw = row_grid(d.MW, d.MH);
    int px[B_IT], py[B_IT], pimg[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        row_to_pixel(rgw, (int)(mbeg + mgb * B_IT + it), pimg[it], py[it], px[it]);      // (M < 2^31: launcher-checked row counts)
    }

    // same carry arithmetic for the dY rows when they are gathered (sub-pixel upsample+conv)
    int qx[A_IT], qy[A_IT], qimg[A_IT];
    const int dy_oy = tap._pad & 15, dy_ox = tap._pad >> 4;
    if (d.dy_gather) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            row_to_pixel(rgw, (int)(mbeg + mga * A_IT + it), qimg[it], qy[it], qx[it]);
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
                    v = *reinterpret_cast<const u32x4*>(dY + row * d.ldy + ycol0 + o0 + oga * EPC);
                }
                qx[it] += step_x;
                if (qx[it] >= d.MW) { qx[it] -= d.MW; qy[it] += 1; }
                qy[it] += step_y;
                if (qy[it] >= d.MH) { qy[it] -= d.MH; qimg[it] += 1; }
                qimg[it] += step_img;
            } else if (m < mend && a_cok) {
                v = *reinterpret_cast<const u32x4*>(dY + m * d.ldy + ycol0 + o0 + oga * EPC);
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
    const float wscale = d.alpha ? *d.alpha : 1.f;
    const int cmax = d.creal > 0 ? d.creal : d.Cs;       // creal: dW rows hold only the REAL input channels (master layout of a dense weight)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int c = c0 + wn * WN + jj * 16 + col_l;
        if (c >= cmax) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + wm * WM + i * 16 + quad * 4 + r;
                if (o >= d.N) continue;
                float* p = d.dW + (long)o * d.lddw + (long)tap.wtap * (d.wstride ? d.wstride : d.Cs) + c;
                const float val = acc[i][jj][r] * wscale;
                if (d.splits > 1) atomicAdd(p, val);
                else if (d.accumulate) atomicAdd(p, val);   // deferred update: earlier calls of this step are already in there.
                // One writer per address and launch, so still one fixed order; the no-return atomic does not stall the
                // epilogue on a load the way `*p += val` did (the weight-gradient launches had become 2x slower)
                else *p = val;                     // single slice: dW is zero on entry, a store saves the read
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

// LIN = linear staging: when the output grid is a power of two wide and at most 64 (every conv map of this model: 4..64; dense
// layers: 1x1) a K tile of 64 pixels is a whole number of grid rows, so a lane keeps its x for the whole kernel, its global
// row index advances by a constant, and - images being contiguous - so do BOTH source addresses (X gathered through any
// stride, dY through the sub-pixel gather): staging a piece is then one predicated pointer select + one 64-bit add. The
// general form below spends ~45 VALU/SALU instructions and two divergent branches per piece (380 per K tile against 32 MFMAs).
template <int WGM, int WGN, int BKM = 64, bool LIN = false>
__global__ __launch_bounds__(NTHREADS) void wgrad_tn_dma_kernel(const cpcsv_wgrad_desc d) {
    constexpr int BM = 128, BN = 128;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int STAGE = 2 * BKM * 256;                       // A image + B image of one K tile (32 KB)
    constexpr int IT = BKM / 4 / 4;                            // wave instructions per operand per wave (4 rows each)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_o = (d.N + BM - 1) / BM;
    const int tiles_c = (d.Cs + BN - 1) / BN;
    int bid = blockIdx.x, split = blockIdx.y;
    if (gridDim.y == 1 && d.splits > 1) wg_xcd_decode(blockIdx.x, tiles_o * tiles_c * d.ntaps, d.splits, bid, split);
    const int tile_o = bid % tiles_o; bid /= tiles_o;
    const int tile_c = bid % tiles_c; bid /= tiles_c;
    const int j = bid;
    const int o0 = tile_o * BM, c0 = tile_c * BN;
    const cpcsv_tap tap = d.taps[j];

    long per = ((long)d.M + d.splits - 1) / d.splits;
    per = (per + BKM - 1) / BKM * BKM;
    const long mbeg = (long)split * per;
    const long mend = (mbeg + per < d.M) ? mbeg + per : d.M;
    if (mbeg >= mend) return;

    const bf16_t* __restrict__ dY = reinterpret_cast<const bf16_t*>(d.dY);
    const bf16_t* __restrict__ X = reinterpret_cast<const bf16_t*>(d.X);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);
    const int BH = d.IH << d.up_shift, BW = d.IW << d.up_shift;
    // second pass of the same layer (rows >= M1): its base pointers, moved back by the first pass's extent so that the
    // running image / row indices address it directly
    const int n1 = d.M1 ? d.M1 / (d.MH * d.MW) : 0;
    const bf16_t* __restrict__ dYb = d.M1 ? reinterpret_cast<const bf16_t*>(d.dY2) - (d.dy_gather ? (long)n1 * d.DYH * d.DYW : (long)d.M1) * d.ldy : dY;
    const bf16_t* __restrict__ Xb = d.M1 ? reinterpret_cast<const bf16_t*>(d.X2) - (long)n1 * d.IH * d.IW * d.Cs : X;

    // lane -> (pixel row within the 4-row group, 16-byte slot); the slot's SOURCE chunk carries the swizzle
    const int lrow = lane >> 4, slot = lane & 15;
    int prow[IT], pchunk[IT];
    int px[IT], py[IT], pimg[IT];
    const int plane = d.MH * d.MW;
    const int step_img = BKM / plane, step_y = (BKM % plane) / d.MW, step_x = BKM % d.MW;
    const RowGrid rgw = row_grid(d.MW, d.MH);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        prow[it] = 4 * (wave + 4 * it) + lrow;                 // pixel row of the tile this lane stages
        pchunk[it] = slot ^ ((prow[it] & 7) << 1);             // channel chunk (8 channels) that lands in this slot
        row_to_pixel(rgw, (int)(mbeg + prow[it]), pimg[it], py[it], px[it]);
    }
    const int dy_oy = tap._pad & 15, dy_ox = tap._pad >> 4;
    const int ycol0 = j * d.dy_tapstride, ylim = d.dy_tapstride ? d.dy_tapstride : d.ldy;    // dy_tapstride: per-tap column block of dY

    auto stage = [&](long mt, int buf) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long m = mt + prow[it];
            const bool live = m < mend;
            const bool second = d.M1 && m >= d.M1;
            // dY piece
            const int oc = o0 + pchunk[it] * 8;
            const bf16_t* pa = zp;
            if (live && oc < ylim) {
                long row = m;
                if (d.dy_gather) row = ((long)pimg[it] * d.DYH + py[it] * d.dy_sy + dy_oy) * d.DYW + px[it] * d.dy_sx + dy_ox;
                pa = (second ? dYb : dY) + row * d.ldy + ycol0 + oc;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                             (__attribute__((address_space(3))) void*)(base + (wave + 4 * it) * 1024), 16, 0, 0);
            // gathered X piece
            const int cc = c0 + pchunk[it] * 8;
            const bf16_t* pb = zp;
            int iy = py[it] * d.sy + tap.oy, ix = px[it] * d.sx + tap.ox;
            if (live && cc < d.Cs && (unsigned)iy < (unsigned)BH && (unsigned)ix < (unsigned)BW) {
                iy >>= d.up_shift; ix >>= d.up_shift;
                pb = (second ? Xb : X) + (((long)pimg[it] * d.IH + iy) * d.IW + ix) * d.Cs + cc;
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

    // ---- linear staging state (LIN): running byte pointers, constant steps, the grid row (for the zero-padding test)
    const unsigned char* la_cur[IT];
    const unsigned char* lb_cur[IT];
    int ly[IT];
    bool la_ok[IT], lb_xok[IT];
    long la_step = 0, lb_step = 0;
    int ly_step = 0;
    const unsigned char* const zpb = reinterpret_cast<const unsigned char*>(g_zero_page);
    if (LIN) {
        const int lw = __builtin_ctz(d.MW);
        const int rows_per_tile = BKM >> lw;                       // grid rows a K tile advances (BKM % MW == 0)
        ly_step = rows_per_tile & (d.MH - 1);
        la_step = (d.dy_gather ? (long)d.dy_sy * rows_per_tile * d.DYW : (long)BKM) * d.ldy * 2;
        lb_step = (long)d.sy * rows_per_tile * d.IW * d.Cs * 2;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long m = mbeg + prow[it];
            const int x = (int)(m & (d.MW - 1));
            const long gy = m >> lw;                               // global grid row = img * MH + y
            ly[it] = (int)(gy & (d.MH - 1));
            const int oc = o0 + pchunk[it] * 8, cc = c0 + pchunk[it] * 8;
            la_ok[it] = oc < ylim;
            const long arow = d.dy_gather ? ((long)d.dy_sy * gy + dy_oy) * d.DYW + x * d.dy_sx + dy_ox : m;
            la_cur[it] = reinterpret_cast<const unsigned char*>(dY + arow * d.ldy + ycol0 + oc);
            const int ix = x * d.sx + tap.ox;
            lb_xok[it] = cc < d.Cs && (unsigned)ix < (unsigned)d.IW;
            lb_cur[it] = reinterpret_cast<const unsigned char*>(X + (((long)d.sy * gy + tap.oy) * d.IW + ix) * d.Cs + cc);
        }
    }
    // the zero page's address as an OPAQUE per-lane value: with a visible constant the compiler turns every "valid ? cursor :
    // zero page" select into a divergent branch with a scalar-addressed load on one side (800 instructions per K tile)
    unsigned long long zp_v = reinterpret_cast<unsigned long long>(zpb);
    asm volatile("" : "+v"(zp_v));
    // `rem`: pixels of this K tile that exist (BKM except in the last tile of a split)
    auto stage_lin = [&](int buf, int rem) {
        unsigned char* base = smem + buf * STAGE;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool live = prow[it] < rem;
            const bool aok = live & la_ok[it];
            const unsigned long long pa = aok ? reinterpret_cast<unsigned long long>(la_cur[it]) : zp_v;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                             (__attribute__((address_space(3))) void*)(base + (wave + 4 * it) * 1024), 16, 0, 0);
            const bool bok = live & lb_xok[it] & ((unsigned)(ly[it] * d.sy + tap.oy) < (unsigned)d.IH);
            const unsigned long long pb = bok ? reinterpret_cast<unsigned long long>(lb_cur[it]) : zp_v;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb,
                                             (__attribute__((address_space(3))) void*)(base + BKM * 256 + (wave + 4 * it) * 1024), 16, 0, 0);
            la_cur[it] += la_step;
            lb_cur[it] += lb_step;
            ly[it] = (ly[it] + ly_step) & (d.MH - 1);
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

    auto stage_any = [&](long mt, int buf) {
        if (LIN) {
            const long left = mend - mt;
            stage_lin(buf, left < BKM ? (int)left : BKM);
        } else {
            stage(mt, buf);
        }
    };
    stage_any(mbeg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (long mt = mbeg; mt < mend; mt += BKM) {
        if (mt + BKM < mend) stage_any(mt + BKM, cur ^ 1);
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
    const float wscale = d.alpha ? *d.alpha : 1.f;
    const int cmax = d.creal > 0 ? d.creal : d.Cs;       // creal: dW rows hold only the REAL input channels (master layout of a dense weight)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int c = c0 + wn * WN + jj * 16 + col_l;
        if (c >= cmax) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + wm * WM + i * 16 + quad * 4 + r;
                if (o >= d.N) continue;
                float* p = d.dW + (long)o * d.lddw + (long)tap.wtap * (d.wstride ? d.wstride : d.Cs) + c;
                const float val = acc[i][jj][r] * wscale;
                if (d.splits > 1) atomicAdd(p, val);
                else if (d.accumulate) atomicAdd(p, val);
                else *p = val;
            }
    }
}

// ------------------------------------------------------------------------------------------
// Patch-resident main loop (bf16): the INPUT PATCH of a 256-row tile - its pixels plus the halo, one 64-channel slice - is
// staged in LDS ONCE per channel tile (stride 2: once per channel tile and input-parity class), and the 4 taps that read it are
// served from LDS by shifted row addressing. The streaming loop above re-stages the tile's rows for every tap (every input
// pixel 4x: per phase of the sub-pixel upsample+conv and of the transposed-conv data gradient, per parity class of a 4x4
// stride-2 conv): per K tile it fills 32 KB (A) + 16 KB (B) through the L2 -> LDS path that bounds it; here the A side is
// patch/4 ~ 10 KB. K order: channel tiles outer, taps inner (cpcsv_gemm_desc.korder = 1 in the streaming kernel gives the same
// order: bit-identical results).
//
// Geometry (host-checked, patch_geometry_ok()): no upsample / pooling / split-K; MW a power of two in [16, 64], MH a power of two;
//   S = 1: grid == input grid, every tap offset in [-1, 1], every phase 4 taps (sub-pixel upsample+conv forward, transposed-conv
//          data gradient: reference model.py:26-34, 502-513);
//   S = 2: input = 2 MH x 2 MW, one phase of 16 taps with offsets in [-1, 2], ordered by input-parity class ((oy+1)&1, (ox+1)&1)
//          in four runs of 4, one class per run (4x4 stride-2 pad-1 convs: the critics' towers forward, the data gradient of the sub-pixel form).
// A tile is 256 consecutive rows = R = 256/MW grid rows of ONE image (MH*MW > 256) or 256/(MH*MW) whole images ("segments").
// Patch pixel (seg, yy, xx) is input pixel (S*(y0+yy) - 1 + cy, S*xx - 1 + cx) of the segment's image, (cy, cx) the parity class
// (0 for S = 1); patch row index = (seg*PR + yy)*PW + xx with PR = R + 3 - S, PW = MW + 3 - S. A tap (oy, ox) reads patch row
// (y + dy, x + dx), dy = oy + 1 (S = 1) or (oy + 1) >> 1 (S = 2). Pixels outside the image and rows of images beyond the row
// group read the zero page. LDS row = 128 B, chunk q of patch row r in slot q ^ (r & 7): any 16 CONSECUTIVE rows are
// conflict-free for ds_read_b128 (the fragment rows of a tap start at an arbitrary patch row).
// Pipeline: B tiles in a 3-stage ring like the streaming 256x128 kernel; the next patch is issued in pieces during the first 3
// taps of the current one, into the other patch buffer. Every wave issues the same number of LDS-DMA instructions per tap
// (surplus instructions re-fill the last row group with the same bytes), so the partial vmcnt waits are constants.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_pw(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <int BN, int S, int PA_IT>
__global__ __launch_bounds__(512) void conv_patch_kernel(const cpcsv_gemm_desc d) {
    using T = bf16_t;
    constexpr int BM = 256, WGM = 4, WGN = 2, NW = 8, NSTAGE = 3, NTAP = 4, NCLS = S * S;
    constexpr int EPC = 8, BK = 64;
    constexpr int WM = BM / WGM, WN = BN / WGN, MI = WM / 16, NI = WN / 16;
    constexpr int GB = BN / 8, B_IT = GB / NW;
    constexpr int CG = NI >= 4 ? 4 : (NI >= 2 ? 2 : 1);
    constexpr int B_BYTES = BN * 128;
    static_assert(GB % NW == 0 && (S == 1 || S == 2) && PA_IT >= 5 && PA_IT <= 7, "patch kernel shape");
    // patch pieces issued during tap j of a patch (none in the last tap: its barrier must see the whole next patch landed)
    constexpr int P0 = PA_IT - 4, P1 = 2, P2 = 2;            // 7 -> {3,2,2,0}, 6 -> {2,2,2,0}, 5 -> {1,2,2,0}

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [2 patch buffers][3 B stages]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int tiles_n = (d.N + BN - 1) / BN;
    const bool phased = d.nphases > 1;
    const int nph = phased ? d.nphases : 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int tile_n, ph, tile_m;
    if (d.order_m_fast) {
        const int tiles_m = m_tiles_of(d, BM);
        tile_m = bid % tiles_m;
        ph = (bid / tiles_m) % nph;
        tile_n = bid / (tiles_m * nph);
    } else {
        tile_n = bid % tiles_n;
        ph = (bid / tiles_n) % nph;
        tile_m = bid / (tiles_n * nph);
    }
    int grp, m0, mlim;
    tile_rows(d, BM, tile_m, grp, m0, mlim);
    const int n0 = tile_n * BN;
    const int tap0 = phased ? d.ph_tap0[ph] : 0;
    const int ooy = phased ? d.ph_ooy[ph] : d.ooy, oox = phased ? d.ph_oox[ph] : d.oox;
    const int ctiles = (d.Cs + BK - 1) / BK;
    const int npg = ctiles * NCLS;                      // patches: (channel tile, parity class), class inner
    const int nk = NTAP * npg;
    const int wslice = d.wstride ? d.wstride : d.Cs;

    // ---- patch geometry ----
    const int lw = __builtin_ctz(d.MW);
    const int plane = d.MH * d.MW;
    const bool whole = plane <= BM;                     // tile = whole images
    const int R = whole ? d.MH : (BM >> lw);            // grid rows per segment
    const int lr = __builtin_ctz(R);
    const int nseg = whole ? BM / plane : 1;
    const int PW = d.MW + 3 - S, PR = R + 3 - S;
    const int P = nseg * PR * PW;                       // patch rows (pixels)
    const int Q = (P + 7) >> 3;                         // 8-row groups = LDS-DMA wave instructions per patch
    const int PATCH_BYTES = Q * 1024;
    unsigned char* const patch0 = smem;
    unsigned char* const bring = smem + 2 * PATCH_BYTES;

    const T* __restrict__ A = reinterpret_cast<const T*>(d.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(d.B);
    const unsigned char* const zpb = reinterpret_cast<const unsigned char*>(g_zero_page);
    unsigned long long zp_v = reinterpret_cast<unsigned long long>(zpb);
    asm volatile("" : "+v"(zp_v));                      // opaque: keeps "valid ? cursor : zero page" a select (see the wgrad kernel)
    const bool has_tail = (d.Cs % BK) != 0;
    const int lrow = lane >> 3, slot = lane & 7;

    // ---- patch staging: instruction `it` of this wave fills row group q = min(wave + 8*it, Q-1) ----
    const int img0 = m0 / plane;                        // first image of the tile
    const int y0 = whole ? 0 : (m0 - img0 * plane) >> lw;
    const unsigned magic_seg = 0xFFFFFFFFu / (unsigned)(PR * PW) + 1u, magic_pw = 0xFFFFFFFFu / (unsigned)PW + 1u;
    const unsigned char* pa_base[PA_IT];                // source of class (0,0), channel tile 0
    int pa_mask[PA_IT], pa_q[PA_IT];                    // bit c: the pixel of parity class c exists
    bool pa_tail_ok[PA_IT];
#pragma unroll
    for (int it = 0; it < PA_IT; ++it) {
        int q = wave + NW * it;
        q = q < Q ? q : Q - 1;
        pa_q[it] = q;
        const int prow = q * 8 + lrow;
        // (prow < 2^16 and the divisors < 2^16: n / d == umulhi(n, ceil(2^32 / d)) exactly; two run-time divisions per piece and
        // lane were ~500 instructions in front of the block's first load)
        const int seg = (int)__umulhi((unsigned)prow, magic_seg);
        const int rem = prow - seg * PR * PW;
        const int yy = (int)__umulhi((unsigned)rem, magic_pw), xx = rem - yy * PW;
        const int img = img0 + seg, iy = S * (y0 + yy) - 1, ix = S * xx - 1;
        const int chunk = slot ^ (prow & 7);
        const bool live = prow < P && (long)img * plane < mlim;
        int mask = 0;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            const int cy = c >> 1, cx = c & 1;
            if (live && (unsigned)(iy + cy) < (unsigned)d.IH && (unsigned)(ix + cx) < (unsigned)d.IW) mask |= 1 << c;
        }
        pa_mask[it] = mask;
        pa_tail_ok[it] = (ctiles - 1) * BK + chunk * EPC < d.Cs;
        pa_base[it] = reinterpret_cast<const unsigned char*>(A + (((long)img * d.IH + iy) * d.IW + ix) * d.Cs + chunk * EPC);
    }
    // stage piece `it` of patch pg = (channel tile, class) into patch buffer pb
    auto stage_patch = [&](int it, int pg, int pb) {
        const int ct = pg / NCLS, run = pg - ct * NCLS;
        // parity class of this run of 4 taps (S = 2): from its first tap
        const int c = S == 1 ? 0 : ((((d.taps[tap0 + 4 * run].oy + 1) & 1) << 1) | ((d.taps[tap0 + 4 * run].ox + 1) & 1));
        const long off = ((long)(c >> 1) * d.IW + (c & 1)) * d.Cs * (long)sizeof(T) + (long)ct * BK * (long)sizeof(T);
        const bool ok = ((pa_mask[it] >> c) & 1) && !(has_tail && ct == ctiles - 1 && !pa_tail_ok[it]);
        const unsigned long long p = ok ? reinterpret_cast<unsigned long long>(pa_base[it] + off) : zp_v;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(patch0 + pb * PATCH_BYTES + pa_q[it] * 1024), 16, 0, 0);
    };

    // ---- B staging (same image as the streaming kernel: rows in col_of order, slot = chunk ^ ((row>>1)&7)) ----
    long b_off[B_IT];
    bool b_ok[B_IT], b_tail_ok[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int g = wave + NW * it;
        const int row = g * 8 + lrow;
        const int n = n0 + col_of<WN, CG>(row);
        const int chunk = slot ^ ((row >> 1) & 7);
        b_ok[it] = n < d.N;
        b_tail_ok[it] = (ctiles - 1) * BK + chunk * EPC < d.Cs;
        b_off[it] = (long)n * d.ldb + chunk * EPC;
    }
    auto stage_b = [&](int kt, int buf) {               // K tile kt = (channel tile, tap), channel tile outer
        const int ct = kt / (NTAP * NCLS), j = kt - ct * (NTAP * NCLS);
        const int wtap_off = d.taps[tap0 + j].wtap * wslice + ct * BK;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int g = wave + NW * it;
            const bool ok = b_ok[it] && !(has_tail && ct == ctiles - 1 && !b_tail_ok[it]);
            const unsigned long long p = ok ? reinterpret_cast<unsigned long long>(B + b_off[it] + wtap_off) : zp_v;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(bring + buf * B_BYTES + g * 1024), 16, 0, 0);
        }
    };

    // ---- per-lane fragment rows in the patch (the tap's (dy, dx) is added per tap) ----
    int pr0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int rl = wm * WM + i * 16 + (lane & 15);
        const int x = rl & (d.MW - 1), yl = rl >> lw;
        const int seg = yl >> lr, y = yl & (R - 1);
        pr0[i] = (seg * PR + y) * PW + x;
    }
    auto tap_delta = [&](int j) {                       // j = tap index within the phase (0 .. 4*NCLS)
        const cpcsv_tap t = d.taps[tap0 + j];
        const int dy = S == 1 ? t.oy + 1 : (t.oy + 1) >> 1, dx = S == 1 ? t.ox + 1 : (t.ox + 1) >> 1;
        return dy * PW + dx;
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the whole first patch, B tiles 0 and 1 ----
#pragma unroll
    for (int it = 0; it < PA_IT; ++it) stage_patch(it, 0, 0);
    stage_b(0, 0);
    stage_b(1, 1);                                      // (nk >= 4)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_IT) : "memory");       // B tile 1 may still be in flight
    __builtin_amdgcn_s_barrier();

    auto mma = [&](const unsigned char* Ap, const unsigned char* Bs, int delta) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const u32x4*>(Ap + lds_pw(pr0[i] + delta, s * 4 + (lane >> 4)));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[j] = *reinterpret_cast<const u32x4*>(Bs + lds_sw(wn * WN + j * 16 + (lane & 15), s * 4 + (lane >> 4)));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::run(b[j], a[i], acc[i][j]);
        }
    };

    int bcur = 0, bfill = 2;
    for (int pg = 0; pg < npg; ++pg) {
        const int pb = pg & 1;
        const bool next_pg = pg + 1 < npg;
        const int jbase = (pg % NCLS) * NTAP;           // first tap of this patch's class within the phase
        // one tap = one K tile; J is a compile-time tap index (the vmcnt immediates differ per tap)
        auto tap_step = [&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int PBEG = j == 0 ? 0 : (j == 1 ? P0 : (j == 2 ? P0 + P1 : P0 + P1 + P2));
            constexpr int PCNT = j == 0 ? P0 : (j == 1 ? P1 : (j == 2 ? P2 : 0));
            const int kt = pg * NTAP + j;
            // pieces of the next patch, then the B tile two iterations ahead
            if (next_pg) {
#pragma unroll
                for (int k = 0; k < PCNT; ++k) stage_patch(PBEG + k, pg + 1, pb ^ 1);
            }
            const bool more_b = kt + 2 < nk;
            if (more_b) stage_b(kt + 2, bfill);
            mma(patch0 + pb * PATCH_BYTES, bring + bcur * B_BYTES, tap_delta(jbase + j));
            // B tile kt+1 (and, behind the last tap, the whole next patch) must have landed; what THIS iteration issued may fly on
            if (more_b && next_pg) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PCNT + B_IT) : "memory");
            else if (more_b) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B_IT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            bcur = bcur + 1 == NSTAGE ? 0 : bcur + 1;
            bfill = bfill + 1 == NSTAGE ? 0 : bfill + 1;
        };
        tap_step(std::integral_constant<int, 0>{});
        tap_step(std::integral_constant<int, 1>{});
        tap_step(std::integral_constant<int, 2>{});
        tap_step(std::integral_constant<int, 3>{});
    }
    nt_epilogue<T, BM, BN, WGM, WGN>(d, acc, smem, tid, lane, wm, wn, grp, m0, mlim, n0, phased, ph, nph, tile_m, tiles_n, ooy, oox);
}

// ---- host-side tile selection ---------------------------------------------------------------
enum NtCfg { NT_128x128, NT_128x64, NT_128x16, NT_64x128, NT_256x128, NT_64x64 };
// CPCSV_NT_BIG=0 keeps every shape on the 4-wave kernels (A/B timing)
static const int g_nt_big = [] { const char* e = getenv("CPCSV_NT_BIG"); return e ? atoi(e) : 1; }();
static const int g_nt_force = [] { const char* e = getenv("CPCSV_NT_FORCE"); return e ? atoi(e) : -1; }();   // sweeps only
// Tile choice by how many blocks each candidate yields (measured on the layer shapes, tools/gemm_sweep.py): the
// largest tile that still gives every CU work; mid-size problems take 128x64 tiles rather than split-K (the fp32
// slabs and the second launch cost more than the narrower tile); only short-M / long-K shapes are left to split-K.
static const int g_nt_t256 = [] { const char* e = getenv("CPCSV_NT_T256"); return e ? atoi(e) : 256; }();   // tile-count thresholds (sweeps)
// (128x128 from 160 tiles on, round 5: 13.37 against 13.50 ms per step over three alternating pairs - the critics' head conv, 184 such
// tiles, leaves the 128x64 tile; 128: 13.40)
static const int g_nt_t128 = [] { const char* e = getenv("CPCSV_NT_T128"); return e ? atoi(e) : 160; }();
static const int g_nt_t64 = [] { const char* e = getenv("CPCSV_NT_T64"); return e ? atoi(e) : 0; }();
inline NtCfg pick_nt(int M, int N, int phases) {
    if (g_nt_force >= 0) return (NtCfg)g_nt_force;
    if (N <= 16) return NT_128x16;
    if (N <= 64) return NT_128x64;
    if (M <= 64) return NT_64x128;
    const long ph = phases > 1 ? phases : 1;
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 128) * ph, t128 = (long)cdiv(M, 128) * cdiv(N, 128) * ph;
    if (g_nt_big && t256 >= g_nt_t256) return NT_256x128;
    if (t128 >= g_nt_t128) return NT_128x128;
    // (experiment knob) 64x64 tiles - 32 KB of LDS, up to five blocks per CU - when even 128x64 leaves at most one block per CU
    if (g_nt_t64 > 0 && (long)cdiv(M, 128) * cdiv(N, 64) * ph <= g_nt_t64) return NT_64x64;
    return NT_128x64;       // few tiles: narrow tiles (+ a modest split-K for long K) beat wide tiles with a deep split
}

inline long out_row_of(const cpcsv_gemm_desc& d, long m) {       // output row of GEMM row m at an image boundary
    if (d.scatter) return (m / (d.MH * d.MW)) * d.OH * d.OW;
    return d.pool_rows ? m / 4 : m;
}
inline long out_rows(const cpcsv_gemm_desc& d) { return out_row_of(d, d.M); }

template <typename T, int BM, int BN, int WGM, int WGN, int NSTAGE = 2>
int launch_nt(const cpcsv_gemm_desc& d, hipStream_t s) {
    const long tiles = (long)m_tiles_of(d, BM) * cdiv(d.N, BN);
    const unsigned gy = d.splitk > 1 ? d.splitk : 1, gz = 1;
    const long phases = d.nphases > 1 ? d.nphases : 1;
    constexpr int lds = NSTAGE * (BM + BN) * 128;
    // d.korder here is the INTERNAL form cpcsv_gemm_nt resolved: bit 0 = channel tiles outer, bit 1 = buffer-resource staging
    using Kern = void (*)(const cpcsv_gemm_desc);
    static const Kern kerns[3] = {gemm_nt_kernel<T, BM, BN, WGM, WGN, NSTAGE, false, false>, gemm_nt_kernel<T, BM, BN, WGM, WGN, NSTAGE, false, true>,
                                  gemm_nt_kernel<T, BM, BN, WGM, WGN, NSTAGE, true, true>};
    const Kern kern = kerns[(d.korder & 1) ? 2 : ((d.korder & 2) ? 1 : 0)];
    if (lds > 64 * 1024) {                                 // more than the default dynamic-LDS cap: raise it once (every variant)
        static const hipError_t once = [] {
            hipError_t r = hipSuccess;
            for (const Kern k : kerns) {
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) r = e;
            }
            return r;
        }();
        if (once != hipSuccess) return -1100 - (int)once;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * phases), gy, gz), dim3(WGM * WGN * 64), lds, s, d);
    CPCSV_CHECK_LAUNCH();
    if (d.splitk > 1 && !d.slabs_only) {
        const long rows = out_rows(d);
        EpiGroups eg;
        eg.n = d.ngroups > 1 ? d.ngroups : 0;
        long blocks = cdiv(rows, EPI_ROWS);
        if (eg.n) {
            blocks = 0;
            for (int g = 0; g <= eg.n; ++g) eg.row[g] = out_row_of(d, d.grow[g]);
            for (int g = 0; g < eg.n; ++g) { eg.alpha[g] = d.galpha[g]; blocks += cdiv(eg.row[g + 1] - eg.row[g], EPI_ROWS); }
        }
        if (d.act >= CPCSV_ACT_TANH) hipLaunchKernelGGL((gemm_epilogue_kernel<T, true>), dim3((unsigned)blocks, (unsigned)cdiv(d.ldc, 64)), dim3(NTHREADS), 0, s, d.ws, d.ldws,
                           d.splitk, d.C, d.ldc, rows, d.N, d.alpha, d.bias, d.act, d.stats, d.ldstat, d.out_f32, eg, d.addend, d.ldadd);
        else hipLaunchKernelGGL((gemm_epilogue_kernel<T, false>), dim3((unsigned)blocks, (unsigned)cdiv(d.ldc, 64)), dim3(NTHREADS), 0, s, d.ws, d.ldws,
                           d.splitk, d.C, d.ldc, rows, d.N, d.alpha, d.bias, d.act, d.stats, d.ldstat, d.out_f32, eg, d.addend, d.ldadd);
        CPCSV_CHECK_LAUNCH();
    }
    return 0;
}
// CPCSV_NT_DEEP (experiment knob, default off = 2): K-tile pipeline depth of the 128x64 / 128x128 tiles when the launch has
// at most ~one block per CU. Alone, such launches (the critics' 8x8 / 16x16 layers: 240-480 blocks, 64+ K tiles of ~0.13 us
// of MFMA work each) are bound by the ~1 us global->LDS latency of every K tile, and 3-4 stages would keep 2-3 tiles in
// flight. Measured in the step: 21.5 ms (2 stages) -> 22.4 (3) -> 23.1 (4): the 96 KB of LDS per block evict the blocks of
// the other streams' kernels from the CU, and it is that cross-stream co-residency that hides the latency today.
// CPCSV_NT_BIG_STAGES=2: the 256x128 tile with a double buffer (96 KB of LDS instead of 144: a 64 KB block of another stream's
// kernel still fits beside it on the CU); experiment knob
constexpr int g_nt_big_stages = 3;       // (the 2-stage form of the 256x128 tile, CPCSV_NT_BIG_STAGES=2: +0.3 ms per step in round 4; knob retired)
constexpr int g_nt_deep = 2;             // (3- / 4-stage 128-wide tiles, CPCSV_NT_DEEP: slower in rounds 2, 4 and 5; knob retired)
constexpr int g_nt_deep_tiles = 520;

// ---- patch-resident main loop: eligibility + launch ----
// CPCSV_PATCH=0: never (A/B runs); cpcsv_gemm_desc.patch = -1 / 1 overrides per call
static const int g_patch = [] { const char* e = getenv("CPCSV_PATCH"); return e ? atoi(e) : 2; }();
constexpr int g_patch_min_blocks = 128;  // (64 / 256: no difference at round 6, profiles/r06_knob_sweep.txt; knob retired)
inline int patch_stride(const cpcsv_gemm_desc& d) {        // 0: not eligible; 1 / 2: the kernel's S
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (d.dtype != CPCSV_BF16 || d.splitk > 1 || d.pool_rows || d.up_shift || d.sy != d.sx || (d.sy != 1 && d.sy != 2) || d.bcol_rows) return 0;
    const int S = d.sy;
    if (d.IH != S * d.MH || d.IW != S * d.MW || !pow2(d.MW) || !pow2(d.MH) || d.MW < 16 || d.MW > 64 || d.Cs < 64) return 0;
    if (d.M % (d.MH * d.MW)) return 0;
    const int nph = d.nphases > 1 ? d.nphases : 1;
    if (S == 2 && (nph != 1 || d.ntaps != 16)) return 0;
    for (int p = 0; p < nph; ++p) {
        const int t0 = d.nphases > 1 ? d.ph_tap0[p] : 0, nt = d.nphases > 1 ? d.ph_ntaps[p] : d.ntaps;
        if (nt != 4 * S * S) return 0;
        for (int j = 0; j < nt; ++j) {
            const int oy = d.taps[t0 + j].oy, ox = d.taps[t0 + j].ox;
            if (oy < -1 || oy > S || ox < -1 || ox > S) return 0;
        }
        if (S == 2) {      // taps come in four runs of 4, one input-parity class ((oy+1)&1, (ox+1)&1) per run, every class once
            int seen = 0;
            for (int r = 0; r < 4; ++r) {
                const int c = (((d.taps[t0 + 4 * r].oy + 1) & 1) << 1) | ((d.taps[t0 + 4 * r].ox + 1) & 1);
                for (int j = 1; j < 4; ++j)
                    if (((((d.taps[t0 + 4 * r + j].oy + 1) & 1) << 1) | ((d.taps[t0 + 4 * r + j].ox + 1) & 1)) != c) return 0;
                seen |= 1 << c;
            }
            if (seen != 15) return 0;
        }
    }
    return S;
}
inline bool patch_geometry_ok(const cpcsv_gemm_desc& d) { return patch_stride(d) != 0; }
inline bool use_patch(const cpcsv_gemm_desc& d) {
    if (d.patch < 0 || !patch_geometry_ok(d)) return false;
    if (d.patch > 0) return true;
    // CPCSV_PATCH: 0 off; 1 the stride-1 phase launches with more than 64 output columns - alone they run 1.00-1.10x the streaming
    // kernel (profiles/r04_patch_probe.txt) with a quarter of its A-side L2 -> LDS traffic; 2 (default since round 6) also the
    // stride-2 windows (neutral in the step, profiles/r06_experiments.txt, a fifth of the operand traffic; their parity-class tap order
    // is no longer a numerics caveat: DESIGN.md section 2); 3: also the 64-column tile (up4_seg forward: 170 against 118 us in the step)
    if (!g_patch || (g_patch < 2 && patch_stride(d) == 2) || (g_patch < 3 && d.N <= 64)) return false;
    const long blocks = (long)m_tiles_of(d, 256) * cdiv(d.N, d.N <= 64 ? 64 : 128) * (d.nphases > 1 ? d.nphases : 1);
    return blocks >= g_patch_min_blocks;
}
template <int BN, int S, int PA_IT>
int launch_patch_t(const cpcsv_gemm_desc& d, hipStream_t s, int lds) {
    auto kern = conv_patch_kernel<BN, S, PA_IT>;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (once != hipSuccess) return -1100 - (int)once;
    const long tiles = (long)m_tiles_of(d, 256) * cdiv(d.N, BN) * (d.nphases > 1 ? d.nphases : 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, s, d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
template <int BN, int S>
int launch_patch_s(const cpcsv_gemm_desc& d, hipStream_t s, int Q, int lds) {
    if (Q <= 40) return launch_patch_t<BN, S, 5>(d, s, lds);
    if (Q <= 48) return launch_patch_t<BN, S, 6>(d, s, lds);
    return launch_patch_t<BN, S, 7>(d, s, lds);
}
inline int launch_patch(const cpcsv_gemm_desc& d, hipStream_t s) {
    const int S = patch_stride(d);
    const int plane = d.MH * d.MW;
    const bool whole = plane <= 256;
    const int R = whole ? d.MH : 256 / d.MW, nseg = whole ? 256 / plane : 1;
    const int P = nseg * (R + 3 - S) * (d.MW + 3 - S), Q = (P + 7) / 8;
    const int bn = d.N <= 64 ? 64 : 128;
    const int lds = 2 * Q * 1024 + 3 * bn * 128;
    if (S == 0 || Q > 56 || lds > 160 * 1024) return -1010;
    if (S == 1) return bn == 64 ? launch_patch_s<64, 1>(d, s, Q, lds) : launch_patch_s<128, 1>(d, s, Q, lds);
    return bn == 64 ? launch_patch_s<64, 2>(d, s, Q, lds) : launch_patch_s<128, 2>(d, s, Q, lds);
}

template <typename T>
int dispatch_nt(const cpcsv_gemm_desc& d, hipStream_t s) {
    if (sizeof(T) == 2 && use_patch(d)) return launch_patch(d, s);
    const NtCfg cfg = pick_nt(d.M, d.N, d.nphases);
    if (g_nt_deep > 2 && (cfg == NT_128x64 || cfg == NT_128x128)) {
        const int bn = cfg == NT_128x64 ? 64 : 128;
        const long blocks = (long)cdiv(d.M, 128) * cdiv(d.N, bn) * (d.nphases > 1 ? d.nphases : 1) * (d.splitk > 1 ? d.splitk : 1);
        if (blocks <= g_nt_deep_tiles) {
            if (cfg == NT_128x64) return g_nt_deep >= 4 ? launch_nt<T, 128, 64, 2, 2, 4>(d, s) : launch_nt<T, 128, 64, 2, 2, 3>(d, s);
            return launch_nt<T, 128, 128, 2, 2, 3>(d, s);
        }
    }
    switch (cfg) {
        case NT_128x16: return launch_nt<T, 128, 16, 4, 1>(d, s);
        case NT_128x64: return launch_nt<T, 128, 64, 2, 2>(d, s);
        case NT_64x128: return launch_nt<T, 64, 128, 1, 4>(d, s);
        case NT_256x128: return g_nt_big_stages == 2 ? launch_nt<T, 256, 128, 4, 2, 2>(d, s) : launch_nt<T, 256, 128, 4, 2, 3>(d, s);
        case NT_64x64: return launch_nt<T, 64, 64, 2, 2>(d, s);
        default: return launch_nt<T, 128, 128, 2, 2, 2>(d, s);
    }
}

// CPCSV_WG_XCD=0: the two-dimensional (tiles, splits) grid of rounds 1-5 (A/B; see wg_xcd_decode)
static const int g_wg_xcd = [] { const char* e = getenv("CPCSV_WG_XCD"); return e ? atoi(e) : 1; }();
inline dim3 wg_grid(long tiles, int splits) {
    const long total = tiles * splits;
    const bool ok = g_wg_xcd && splits > 1 && (total & 7) == 0 && total < (1l << 31) &&
                    ((splits & 7) == 0 || ((splits == 2 || splits == 4) && tiles % (8 / splits) == 0));
    return ok ? dim3((unsigned)total, 1u) : dim3((unsigned)tiles, (unsigned)splits);
}
template <typename T, int BM, int BN, int WGM, int WGN>
int launch_wg(const cpcsv_wgrad_desc& d, hipStream_t s) {
    const long tiles = (long)cdiv(d.N, BM) * cdiv(d.Cs, BN) * d.ntaps;
    hipLaunchKernelGGL((wgrad_tn_kernel<T, BM, BN, WGM, WGN>), wg_grid(tiles, d.splits), dim3(NTHREADS), 0, s, d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
// CPCSV_WG_BKM=32: pixels per K tile of the LDS-DMA weight-gradient kernel (32 KB of LDS per block instead of 64: more
// blocks of OTHER kernels stay resident beside it; experiment knob)
constexpr int g_wg_bkm = 64;             // (32 pixels per K tile, CPCSV_WG_BKM=32: slower in round 2; knob retired)
static int g_wg_lin = [] { const char* e = getenv("CPCSV_WG_LIN"); return e ? atoi(e) : 1; }();      // A/B switch (cpcsv_set_wgrad_linear)
inline bool wg_linear_ok(const cpcsv_wgrad_desc& d) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    if (!g_wg_lin || d.M1 || d.up_shift || !pow2(d.MW) || !pow2(d.MH) || d.MW > 64) return false;
    if (d.IH != d.MH * d.sy || d.IW != d.MW * d.sx) return false;                   // input rows of consecutive images are contiguous in the row index
    if (d.dy_gather && (d.DYH != d.MH * d.dy_sy || d.DYW != d.MW * d.dy_sx)) return false;
    return true;
}
inline int launch_wg_dma(const cpcsv_wgrad_desc& d, hipStream_t s) {
    const long tiles = (long)cdiv(d.N, 128) * cdiv(d.Cs, 128) * d.ntaps;
    if (g_wg_bkm != 32 && wg_linear_ok(d)) {
        hipLaunchKernelGGL((wgrad_tn_dma_kernel<2, 2, 64, true>), wg_grid(tiles, d.splits), dim3(NTHREADS), 0, s, d);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    if (g_wg_bkm == 32) {
        hipLaunchKernelGGL((wgrad_tn_dma_kernel<2, 2, 32>), wg_grid(tiles, d.splits), dim3(NTHREADS), 0, s, d);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL((wgrad_tn_dma_kernel<2, 2>), wg_grid(tiles, d.splits), dim3(NTHREADS), 0, s, d);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int dispatch_wg(const cpcsv_wgrad_desc& d, hipStream_t s) {
    if (d.M1 && !(sizeof(T) == 2 && d.N > 64 && d.Cs > 64 && !d.legacy)) return -1006;      // two-pass form: LDS-DMA kernel only
    if (sizeof(T) == 2 && d.N > 64 && d.Cs > 64 && !d.legacy) return launch_wg_dma(d, s);
    const bool rows_small = d.N <= 32, cols_small = d.Cs <= 64;
    if (rows_small) return cols_small ? launch_wg<T, 32, 64, 1, 4>(d, s) : launch_wg<T, 32, 128, 1, 4>(d, s);
    if (d.N <= 64) return cols_small ? launch_wg<T, 64, 64, 2, 2>(d, s) : launch_wg<T, 64, 128, 1, 4>(d, s);
    return cols_small ? launch_wg<T, 128, 64, 2, 2>(d, s) : launch_wg<T, 128, 128, 2, 2>(d, s);
}

}  // namespace

#if CPCSV_PROBE & 8
// tools/nt_cycles.py (a -DCPCSV_PROBE=8 build of this file): [issue, mma, wait, total, blocks, K tiles] sums of wave 0 of every block
extern "C" int cpcsv_probe_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#endif

extern "C" int cpcsv_set_wgrad_linear(int on) {
    const int was = g_wg_lin;
    g_wg_lin = on ? 1 : 0;
    return was;
}

extern "C" int cpcsv_gemm_mtile(const cpcsv_gemm_desc* d) {
    if (d->splitk > 1) return EPI_ROWS;
    if (use_patch(*d)) return 256;
    const NtCfg c = pick_nt(d->M, d->N, d->nphases);
    return (c == NT_64x128 || c == NT_64x64) ? 64 : (c == NT_256x128 ? 256 : 128);
}

extern "C" int cpcsv_gemm_ntile(const cpcsv_gemm_desc* d) {
    const NtCfg c = pick_nt(d->M, d->N, d->nphases);
    return c == NT_128x16 ? 16 : ((c == NT_128x64 || c == NT_64x64) ? 64 : 128);
}

extern "C" int cpcsv_gemm_nt(const cpcsv_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return -1001;
    if (d->M <= 0 || d->N <= 0 || d->ntaps <= 0 || d->ntaps > CPCSV_MAX_TAPS) return -1002;
    if (d->Cs % 8 || d->ldb % 8) return -1003;
    if (d->pool_rows && (d->scatter || d->stats || (d->M & 3))) return -1004;
    if (d->splitk > 1 && (!d->ws || d->ldws < d->N || d->ws_rows <= 0)) return -1005;
    if (d->nphases > 4 || (d->nphases > 1 && !d->scatter)) return -1006;
    if (d->stats && d->scatter && d->splitk <= 1 && d->nphases <= 1) return -1007;   // partials are indexed by (phase, M tile)
    if (d->patch > 0 && !patch_geometry_ok(*d)) return -1010;
    if (d->korder < 0 || d->korder > 2 || (d->korder == 1 && d->up_shift)) return -1011;
    if (d->addend && (d->pool_rows || d->scatter || d->ldadd < d->N || (d->ldadd & 3))) return -1009;
    if (d->slabs_only && d->splitk <= 1) return -1012;
    if (d->bcol_rows < 0 || (d->bcol_rows && (d->bcol_koff % 8 || d->scatter || d->pool_rows || d->ntaps != 1))) return -1013;
    if (d->ngroups > 4) return -1008;
    if (d->ngroups > 1) {
        if (d->grow[0] != 0 || d->grow[d->ngroups] != d->M) return -1008;
        for (int g = 0; g < d->ngroups; ++g)
            if (d->grow[g + 1] <= d->grow[g] || d->grow[g + 1] % (d->MH * d->MW) || (d->pool_rows && (d->grow[g + 1] & 3))) return -1008;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // K order of the streaming main loop (the patch-resident one always walks channel tiles outer): 0 = the library's choice =
    // taps outer. Channel tiles outer (1; CPCSV_KORDER=1 makes it the choice wherever up_shift == 0 and there is more than one tap)
    // takes 40 % off the family's fabric reads (60.7 -> 36.2 GB per 9 steps; up3's data gradient 884 -> 270 MB per launch) and
    // costs +0.2 ... +0.3 ms per step even with the buffer-form cursor (13.34-13.43 against 13.09-13.15): measured, not the
    // default (profiles/r06_experiments.txt, profiles/r06_korder_traffic.txt)
    static const int g_korder = [] { const char* e = getenv("CPCSV_KORDER"); return e ? atoi(e) : 0; }();
    // CPCSV_NT_BUF=0: the pointer form of the staging cursors (rounds 2-5) everywhere (A/B)
    static const int g_nt_buf = [] { const char* e = getenv("CPCSV_NT_BUF"); return e ? atoi(e) : 1; }();
    cpcsv_gemm_desc e = *d;
    const long esz = d->dtype == CPCSV_BF16 ? 2 : 4;
    const long a_bytes = ((long)d->M / (d->MH * d->MW > 0 ? d->MH * d->MW : 1) + 1) * d->IH * d->IW * d->Cs * esz;
    const long b_bytes = (long)(d->bcol_rows ? d->bcol_rows : d->N) * d->ldb * esz + (d->bcol_rows ? (long)(d->N / d->bcol_rows + 1) * d->bcol_koff * esz : 0);
    const bool buf = g_nt_buf && a_bytes < (1l << 31) - (1l << 24) && b_bytes < (1l << 31) - (1l << 24);
    int ko = d->korder ? d->korder : g_korder;
    if (ko == 0) ko = 2;
    if (ko == 1 && (d->up_shift || d->ntaps == 1 || !buf)) ko = 2;
    e.korder = (ko == 1 ? 1 : 0) | (buf ? 2 : 0);
    return e.dtype == CPCSV_BF16 ? dispatch_nt<bf16_t>(e, s) : dispatch_nt<float>(e, s);
}

extern "C" int cpcsv_wgrad_tn(const cpcsv_wgrad_desc* d, void* stream) {
    if (!d || !d->dY || !d->X || !d->dW) return -1001;
    if (d->M <= 0 || d->N <= 0 || d->ntaps <= 0 || d->ntaps > CPCSV_MAX_TAPS || d->splits < 1) return -1002;
    if (d->Cs % 8 || d->ldy % 8) return -1003;
    if (d->wstride < 0 || d->wstride % 8 || (d->wstride && d->wstride < d->Cs)) return -1007;
    if (d->dy_tapstride < 0 || d->dy_tapstride % 8 || (d->dy_tapstride && (d->dy_tapstride < d->N || (long)d->ntaps * d->dy_tapstride > d->ldy || d->dy_gather || d->M1)))
        return -1008;
    if (d->M1 && (!d->dY2 || !d->X2 || d->M1 % 64 || d->M1 % (d->MH * d->MW) || d->M1 >= d->M)) return -1005;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_cpcsv_deterministic && d->splits > 1) {       // one block walks all pixels of its tile: plain stores, one order
        cpcsv_wgrad_desc one = *d;
        one.splits = 1;
        return one.dtype == CPCSV_BF16 ? dispatch_wg<bf16_t>(one, s) : dispatch_wg<float>(one, s);
    }
    return d->dtype == CPCSV_BF16 ? dispatch_wg<bf16_t>(*d, s) : dispatch_wg<float>(*d, s);
}
