// thin.hip — streaming convolutions whose GEMM has a degenerate dimension (gfx950, bf16).
//
// The generator's output convs (StoryGAN.img 128->3 + tanh, img_seg 64->1 + tanh: reference model.py:272-274,298-300)
// and the critics' first conv (encode_img.0, 3|1->124 k4 s2 p1 + LeakyReLU: model.py:499-500,541-542,583-584) move tens
// of MB per call for ~1 GFLOP: they are HBM-bound (AI 9-44 FLOP/B), and run through the general
// gather-GEMM they re-gather the input once per tap through the vector L1 (5-17 % of the HBM roofline in round 1).
// Here the wide tensor crosses HBM->LDS exactly ONCE per tile (LDS-DMA, XOR-swizzled so the MFMA fragment reads are
// conflict-free), all taps are served from LDS, and the matrix cores do the arithmetic with the thin dimension padded
// to one 16-wide MFMA tile (the MFMA rate is 16x the VALU rate, so the padding is free while memory-bound).
//
//   thin3x3_fwd    y[p][o]  = act( sum_{tap,c} x[p+tap][c] w[o][tap][c] )          64-wide images: taps-as-rows kernel (x streamed
//                                                                                   into MFMA fragments) / rolling-row kernel; else halo tiles
//   thin3x3_dgrad  dx[p][c] = sum_{tap,o} dz[p-tap][o] w[o][tap][c]                 K = 9 taps x 8 stored channels
//   thin3x3_wgrad  G[o][tap][c] += sum_q dz[q-tap][o] x[q][c]                       64-wide: x unshifted, (tap, o) pairs as MFMA rows
//   thin4x4s2_fwd  y[p][o]  = act( alpha * sum_{tap,c<8} x[2p+tap][c] w[o][tap][c] ) K = 16 taps x 8 stored channels
//   thin4x4s2_wgrad G[o][tap*8+c] += sum_p dz[p][o] x[2p+tap][c]                    all 16 taps x 8 channels as one MFMA dimension
#include "common.h"
#include <cstdlib>
#include "../../include/cpcsv_hip.h"

namespace {

__device__ __attribute__((aligned(256))) const unsigned int t_zero_page[64] = {0};
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& rows, const u32x4& cols, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, rows), __builtin_bit_cast(bf16x8_t, cols), c, 0, 0, 0);
}
// tanh through the hardware exponential: 1 - 2/(1 + e^{2x}); |error| < 3e-7, far below the bf16 output step
__device__ __forceinline__ float fast_tanh(float v) {
    const float e = __expf(2.f * fminf(fmaxf(v, -15.f), 15.f));
    return 1.f - 2.f * __frcp_rn(1.f + e);
}
__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }

// ---- LDS image of a halo tile: [pixel][CS] rows of PIXB = 2*CS bytes, 16-byte chunk q of pixel column pc stored in slot
// q ^ key(pc). MODE 0 (plain fragment reads, ds_read_b128 of 16 consecutive pixel columns at one chunk): key = pc & 15 for
// 256-byte pixels, (pc >> 1) & 7 for 128-byte pixels. MODE 1 (transposing reads, 4 consecutive pixel columns x 32-byte
// channel segments): key keeps the 32-byte pairs together and rotates them by the pixel column.
template <int CS, int MODE>
__device__ __forceinline__ int swz(int chunk, int pc) {
    constexpr int NCH = CS / 8;
    if (MODE == 0) return NCH == 16 ? (chunk ^ (pc & 15)) : (chunk ^ ((pc >> 1) & 7));
    return NCH == 16 ? (chunk ^ ((pc & 7) << 1)) : (chunk ^ (((pc >> 1) & 3) << 1));
}

// Stage the (R+2) x (TW+2) halo tile of image `img`, rows r0-1.., columns c0-1.. into LDS (zero outside the image).
template <int CS, int MODE>
__device__ __forceinline__ void stage_halo(unsigned char* smem, const bf16_t* __restrict__ x, int img, int r0, int c0, int R, int TW,
                                           int H, int W, int wave, int lane) {
    constexpr int NCH = CS / 8;
    const int npix = (R + 2) * (TW + 2);
    const int ninstr = (npix * NCH + 63) / 64;
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(t_zero_page);
    for (int i = wave; i < ninstr; i += 4) {
        const int ci = i * 64 + lane;
        const int pi = ci / NCH, phys = ci - pi * NCH;
        const int prow = pi / (TW + 2), pcol = pi - prow * (TW + 2);
        const int y = r0 - 1 + prow, xx = c0 - 1 + pcol;
        const unsigned char* src = zp;
        if (pi < npix && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W)
            src = reinterpret_cast<const unsigned char*>(x + (((long)img * H + y) * W + xx) * CS + swz<CS, MODE>(phys, pcol) * 8);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward: block = 4 wavefronts, tile = R rows x TW columns of one image; a wavefront owns 16-pixel row segments.
// ------------------------------------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(256) void thin3x3_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                          int H, int W, int Cout, int act, int R, int TW, int probe, long ntiles) {
    constexpr int PIXB = CS * 2, KC = CS / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / TW, tiles_y = (H + R - 1) / R;

    // weight fragments (rows = output channels, zero beyond Cout) stay in registers for the whole (persistent) block:
    // loading them per tile cost as much vector-L1 traffic as the tile itself
    const int n = lane & 15, quad = lane >> 4;
    u32x4 bw[9][KC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
        {   // every lane loads (a clamped row), the pad rows are zeroed afterwards: a predicated load per fragment compiles to
            // 36 exec-masked blocks, 1.4 us of prologue
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + (long)(n < Cout ? n : 0) * 9 * CS + t * CS + kc * 32 + quad * 8);
            bw[t][kc] = n < Cout ? v : u32x4{0u, 0u, 0u, 0u};
        }

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    long b = tile;
    const int tx = (int)(b % tiles_x); b /= tiles_x;
    const int ty = (int)(b % tiles_y); b /= tiles_y;
    const int img = (int)b, r0 = ty * R, c0 = tx * TW;
    __syncthreads();                                                   // the previous tile's fragment reads are done
    if (!(probe & 2)) stage_halo<CS, 0>(smem, x, img, r0, c0, R, TW, H, W, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // A wavefront owns column group `wave` (TW = 64: four groups of 16 pixels) and walks the tile's rows in PAIRS: two
    // rows x two k-halves = four independent accumulator chains, and the fragment reads of a whole tap (2*KC ds_read_b128)
    // are issued before its MFMAs - a single dependent chain of 9*KC MFMAs, each waiting for its own LDS read, took
    // ~120 cycles per MFMA.
    const int gpr = (probe & 1) ? 0 : TW / 16;
    for (int cg = wave; cg < gpr; cg += 4) {
        for (int r = 0; r < R; r += 2) {
            if (r0 + r >= H) break;
            const bool two = (r + 1 < R) && (r0 + r + 1 < H);
            f32x4 acc[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[i][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int pcol = cg * 16 + n + t % 3;
                const unsigned char* b0 = smem + ((r + t / 3) * (TW + 2) + pcol) * PIXB;
                const unsigned char* b1 = two ? b0 + (TW + 2) * PIXB : b0;   // next output row
                u32x4 a0[KC], a1[KC];
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    const int off = swz<CS, 0>(kc * 4 + quad, pcol) * 16;
                    a0[kc] = *reinterpret_cast<const u32x4*>(b0 + off);
                    a1[kc] = *reinterpret_cast<const u32x4*>(b1 + off);
                }
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    acc[0][kc & 1] = mfma_bf16(bw[t][kc], a0[kc], acc[0][kc & 1]);
                    acc[1][kc & 1] = mfma_bf16(bw[t][kc], a1[kc], acc[1][kc & 1]);
                }
            }
            // lane holds output channels quad*4 .. +3 of pixel n: quad 0 carries the real channels, quad 1 the zero pads
            if (quad < 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 1 && !two) break;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sum = acc[i][0][e] + acc[i][1][e];
                        v[e] = (quad == 0 && e < Cout) ? (act == CPCSV_ACT_TANH ? fast_tanh(sum) : act_apply(sum, act)) : 0.f;
                    }
                    u32x2 pk = {pack2(v[0], v[1]), pack2(v[2], v[3])};
                    *reinterpret_cast<u32x2*>(y + ((((long)img * H + r0 + r + i) * W + c0 + cg * 16 + n) * 8 + quad * 4)) = pk;
                }
            }
        }
    }
    }   // tiles
}

// ------------------------------------------------------------------------------------------------------------------
// forward, full-width images (W == 64): ROLLING rows. A block owns SPAN output rows of one image and keeps a ring of NR = 8
// input rows in LDS (no column halo: columns -1 and 64 are the zero padding, served from a zero pixel). While rows r-1..r+2
// feed the MFMAs of output rows r, r+1, the LDS-DMA loads of rows r+3..r+6 are in flight: every input row crosses
// HBM->LDS once (+2 halo rows per span) and the loads never drain. (The tile kernel above stages a (R+2)-row tile, waits,
// computes: with R = 2 it reads every row twice and overlaps nothing inside a block.)
// ------------------------------------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(512) void thin3x3_fwd_roll_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                               int H, int Cout, int act, int SPAN, long long* dbg) {
    constexpr int W = 64, PIXB = CS * 2, KC = CS / 32, NCH = CS / 8, NR = 8, ROWB = W * PIXB, NW = 8;
    int dbi = 0;
    auto stamp = [&]() { if (dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && dbi < 60) dbg[(blockIdx.x ? 64 : 0) + dbi++] = wall_clock64(); };
    stamp();
    constexpr int IPW = W * NCH / 64 / NW;                            // DMA instructions per wavefront and row (2 | 1)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [NR][W][PIXB] ring + one zero pixel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int spans = H / SPAN;
    const int img = blockIdx.x / spans, row0 = (blockIdx.x % spans) * SPAN;
    const int n = lane & 15, quad = lane >> 4;
    unsigned char* zpix = smem + NR * ROWB;
    if (tid < PIXB / 4) reinterpret_cast<uint32_t*>(zpix)[tid] = 0u;

    u32x4 bw[9][KC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
        {   // every lane loads (a clamped row), the pad rows are zeroed afterwards: a predicated load per fragment compiles to
            // 36 exec-masked blocks, 1.4 us of prologue
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + (long)(n < Cout ? n : 0) * 9 * CS + t * CS + kc * 32 + quad * 8);
            bw[t][kc] = n < Cout ? v : u32x4{0u, 0u, 0u, 0u};
        }

    // this lane's IPW chunks of a row: instruction i = wave + NW*k moves chunks i*64 .. i*64+63 (lane-linear LDS destination)
    int src_off[IPW];
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        const int ci = (wave + NW * k) * 64 + lane;
        const int pc = ci / NCH, phys = ci - pc * NCH;
        src_off[k] = pc * CS + swz<CS, 0>(phys, pc) * 8;
    }
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(t_zero_page);
    const int last_needed = row0 + SPAN;                              // input rows row0-1 .. row0+SPAN
    auto issue_row = [&](int row) {
        const bool real = (unsigned)row < (unsigned)H && row <= last_needed;
        const bf16_t* rb = x + ((long)img * H + row) * W * CS;
        unsigned char* dst = smem + ((row + 1) & (NR - 1)) * ROWB;
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            const unsigned char* src = real ? reinterpret_cast<const unsigned char*>(rb + src_off[k]) : zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + (wave + NW * k) * 1024), 16, 0, 0);
        }
    };
    stamp();
#pragma unroll 1
    for (int row = row0 - 1; row <= row0 + 4; ++row) issue_row(row);
    stamp();

    // eight wavefronts (two per SIMD, so that one's LDS reads hide behind the other's MFMAs): wavefront = 16-pixel column
    // group (wave & 3) of output row r + (wave >> 2) of the current row pair
    const int cg = wave & 3, ri = wave >> 2;
#pragma unroll 1
    for (int r = row0; r < row0 + SPAN; r += 2) {
        // rows <= r+2 have landed once at most the two rows issued last (r+3, r+4) are outstanding. (Output stores issued in
        // between also count in vmcnt and retire in order, which only makes this wait stricter.)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
        __syncthreads();                                              // ... for every wavefront; and rows r-3, r-2 are free
        stamp();
        issue_row(r + 5);
        issue_row(r + 6);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int pcol = cg * 16 + n + t % 3 - 1;
            const bool inside = (unsigned)pcol < (unsigned)W;
            const int pcs = inside ? pcol : 0;
            const unsigned char* b0 = inside ? smem + ((r + ri + t / 3) & (NR - 1)) * ROWB + pcs * PIXB : zpix;   // input row r+ri + t/3 - 1
            u32x4 a0[KC];
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                a0[kc] = *reinterpret_cast<const u32x4*>(b0 + (inside ? swz<CS, 0>(kc * 4 + quad, pcs) : (kc * 4 + quad)) * 16);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) acc[kc & 1] = mfma_bf16(bw[t][kc], a0[kc], acc[kc & 1]);
        }
        if (quad < 2) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sum = acc[0][e] + acc[1][e];
                v[e] = (quad == 0 && e < Cout) ? (act == CPCSV_ACT_TANH ? fast_tanh(sum) : act_apply(sum, act)) : 0.f;
            }
            u32x2 pk = {pack2(v[0], v[1]), pack2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(y + ((((long)img * H + r + ri) * W + cg * 16 + n) * 8 + quad * 4)) = pk;
        }
        stamp();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the look-ahead rows of the last iterations
    stamp();
}

// ------------------------------------------------------------------------------------------------------------------
// forward, 64-wide images, taps as an MFMA dimension:  P[q][(tap, o)] = sum_c x[q][c] w[o][tap][c]  for every input pixel q,
// then  y[p][o] = act( sum_tap P[p + tap][(tap, o)] ).  x is consumed UNSHIFTED, so its MFMA fragment (pixel n, channels
// quad*8..+7 of a 32-channel chunk) is one 16-byte global load per lane - no LDS staging, no LDS-DMA (whose issue rate per
// CU bounds the rolling kernel above), PF input rows in flight in registers. The (tap, o) pairs are the MFMA rows (27 of 32
// for RGB, 9 of 16 for the segmentation layer): 2*KC MFMAs per 16 pixels instead of 9*KC. P (32 floats per pixel) goes
// through a 4-row ring in LDS, from which the nine shifted terms of an output pixel are summed.
// ------------------------------------------------------------------------------------------------------------------
template <int CS, int SPAN, int PF>
__global__ __launch_bounds__(256) void thin3x3_fwd_taps_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                               int H, int Cout, int act) {
    constexpr int W = 64, KC = CS / 32, PP = 32;                   // PP: floats per pixel in the P ring
    __shared__ __attribute__((aligned(16))) float Ps[8 * W * PP];  // [slot][pixel][PP], 8-row ring
    const int lane = threadIdx.x & 63;
    const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 16-pixel column group of this wavefront
    const int spans = H / SPAN;
    const int img = blockIdx.x / spans, r0 = (blockIdx.x % spans) * SPAN;
    const int n = lane & 15, quad = lane >> 4;
    const int nrows = 9 * Cout, RT = (nrows + 15) / 16;

    u32x4 wr[2][KC];                                               // rows (tap, o) of the two row tiles
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int idx = rt * 16 + n;
        const bool ok = idx < nrows;
        const int tap = ok ? idx / Cout : 0, o = ok ? idx - tap * Cout : 0;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + ((long)o * 9 + tap) * CS + kc * 32 + quad * 8);
            wr[rt][kc] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // Everything a step needs beyond compile-time constants is computed here once: the kernel is VALU-issue-bound (one
    // wavefront per SIMD), the first version spent ~250 instructions per row on addresses and selects, this one ~60.
    const int px = cg * 16 + n;
    const bf16_t* xb = x + (((long)img * H + r0) * W + px) * CS + quad * 8;        // row r0; + (i - r0) * W * CS + kc * 32
    u32x4 xf[PF][KC];
    auto fetch = [&](int s_, int slot) {                           // input row r0 - 1 + s_; only the first and the last step of a
        const int di = s_ - 1;                                     // span can fall outside the image (compile-time s_)
        const bool edge = s_ == 0 || s_ == SPAN + 1;
        const bool in = !edge || (unsigned)(r0 + di) < (unsigned)H;
        const long ro = (long)(in ? di : 0) * W * CS;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xb + ro + kc * 32);
            xf[slot][kc] = (!edge || in) ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) fetch(s, s);

    const int oc = quad < Cout ? quad : 0;
    unsigned poff[3];                                              // byte offset of P[pixel px + dx - 1][tap (0, dx), channel oc] in a ring row
    float pm[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int q = px + dx - 1, qc = q < 0 ? 0 : (q > W - 1 ? W - 1 : q);
        poff[dx] = (unsigned)((qc * PP + dx * Cout + oc) * 4);
        pm[dx] = (q == qc && quad < Cout) ? 1.f : 0.f;
    }
    const unsigned tapstep = (unsigned)(3 * Cout * 4);             // one kernel row (dy) further in the (tap, o) index
    const unsigned pwr = (unsigned)((px * PP + quad * 4) * 4);
    const bool even = (quad & 1) == 0;
    bf16_t* yb = y + (((long)img * H + r0) * W + px) * 8 + (even ? quad : quad + 3);
    const unsigned char* Pb = reinterpret_cast<const unsigned char*>(Ps);
    static_assert(SPAN % 4 == 0, "ring slots are compile-time constants only if spans start at a multiple of 4");

#pragma unroll
    for (int s = 0; s < SPAN + 3; ++s) {
        // step s: MFMAs of input row r0 - 1 + s into ring slot s & 7, and - one step BEHIND, so that its LDS reads do not wait
        // for this step's writes and barrier - the output row r0 + s - 3 from the P rows of steps s-3, s-2, s-1
        float sum = 0.f;
        if (s >= 3) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    sum += pm[dx] * *reinterpret_cast<const float*>(Pb + ((s - 3 + dy) & 7) * (W * PP * 4) + dy * tapstep + poff[dx]);
        }
        if (s < SPAN + 2) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                acc[0] = mfma_bf16(wr[0][kc], xf[s % PF][kc], acc[0]);
                if (RT > 1) acc[1] = mfma_bf16(wr[1][kc], xf[s % PF][kc], acc[1]);
            }
            if (s + PF < SPAN + 2) fetch(s + PF, s % PF);          // refill the slot just consumed
            unsigned char* prow = const_cast<unsigned char*>(Pb) + (s & 7) * (W * PP * 4) + pwr;
            *reinterpret_cast<f32x4*>(prow) = acc[0];
            *reinterpret_cast<f32x4*>(prow + 64) = acc[1];
        }
        if (s >= 3) {
            // lane (pixel n, quad) holds output channel o = quad; neighbouring quads swap so that every lane stores two channels
            // (4 bytes): quads 0 / 2 hold (o0, o1) / (o2, o3), quads 1 / 3 the zero pads
            if (act == CPCSV_ACT_TANH) sum = fast_tanh(sum);
            else sum = act_apply(sum, act);
            sum *= pm[1];                                          // pad channels stay zero whatever the activation maps 0 to
            const float other = __shfl_xor(sum, 16);               // quad ^ 1
            // even quad q stores channels (q, q+1); odd quad q stores the pad channels: 0->0,1  2->2,3  1->4,5  3->6,7
            *reinterpret_cast<uint32_t*>(yb + (long)(s - 3) * W * 8) = even ? pack2(sum, other) : 0u;
        }
        if (s < SPAN + 2) __syncthreads();                         // row s of P is complete for every column group
    }
}

// ------------------------------------------------------------------------------------------------------------------
// data gradient: K = 9 taps x 8 stored channels of dz (3 MFMA k-steps of 4 taps), all Cin per wavefront, 16-byte stores
// ------------------------------------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(256) void thin3x3_dgrad_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx,
                                                            int H, int W, long ngroups) {
    constexpr int NT = CS / 16;                        // column tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, quad = lane >> 4;


    // weight fragments: fetched once per block into LDS, copied to registers per wavefront (see thin4x4s2_fwd_kernel)
    __shared__ u32x4 wsm[NT * 3 * 64];
    for (int e = threadIdx.x; e < NT * 3 * 64; e += blockDim.x) {
        const int ln = e & 63, f = e / 64, j = f / 3, ks = f - j * 3;
        const int nn = ln & 15, qq = ln >> 4;
        const int ch = ((j >> 1) * 4 + (nn >> 2)) * 8 + (j & 1) * 4 + (nn & 3);    // see thin4x4s2_fwd_kernel: 64-byte store runs
        const int t = ks * 4 + qq;
        wsm[e] = t < 9 ? *reinterpret_cast<const u32x4*>(wb + (long)ch * 72 + t * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    u32x4 bw[NT][3];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) bw[j][ks] = wsm[(j * 3 + ks) * 64 + lane];
    const int gpr = W / 16;
    // (wave-uniform 32-bit index arithmetic: as 64-bit divisions on a VGPR loop variable these three lines were ~350 instructions per
    // trip, against ~100 of loads, MFMAs and stores)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int gi = blockIdx.x * 4 + wave_u; gi < (int)ngroups; gi += gridDim.x * 4) {
        const int cg = gi % gpr;
        const long rowi = gi / gpr;                     // img*H + y
        const int yy = (int)((gi / gpr) % H);
        const int xx = cg * 16 + n;
        u32x4 a[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int t = ks * 4 + quad;
            const int sy = yy + 1 - t / 3, sx = xx + 1 - t % 3;
            a[ks] = (t < 9 && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
                        ? *reinterpret_cast<const u32x4*>(dz + ((rowi - yy + sy) * W + sx) * 8) : u32x4{0u, 0u, 0u, 0u};
        }
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) acc[j] = mfma_bf16(bw[j][ks], a[ks], acc[j]);
        }
        bf16_t* out = dx + (rowi * W + xx) * CS;
#pragma unroll
        for (int j = 0; j < NT; j += 2) {
            u32x4 pk = {pack2(acc[j][0], acc[j][1]), pack2(acc[j][2], acc[j][3]), pack2(acc[j + 1][0], acc[j + 1][1]),
                        pack2(acc[j + 1][2], acc[j + 1][3])};
            *reinterpret_cast<u32x4*>(out + ((j >> 1) * 4 + quad) * 8) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient: reduction over pixels. Persistent blocks walk tiles; x halo tile in LDS (transposing reads give every
// lane 4 consecutive pixels of its channel), dz tile in LDS; a wavefront owns NT/4 channel tiles x 9 taps of accumulators;
// every block writes ONE partial [Cout][9*CS] slab, a second kernel adds the slabs into G in a fixed order.
// ------------------------------------------------------------------------------------------------------------------
template <int CS>
__global__ __launch_bounds__(256) void thin3x3_wgrad_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x, float* __restrict__ slabs,
                                                            int N, int H, int W, int Cout, int R, int TW) {
    constexpr int PIXB = CS * 2, NT = CS / 16, TPW = NT / 4;      // channel tiles per wavefront
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int halo_bytes = (((R + 2) * (TW + 2) * PIXB + 1023) / 1024) * 1024;
    unsigned char* dzs = smem + halo_bytes;                       // [R*TW][16 bytes]
    const int tiles_x = W / TW, tiles_y = (H + R - 1) / R;
    const long ntiles = (long)N * tiles_y * tiles_x;
    const int gi = lane & 15, quad = lane >> 4;
    const int br = gi >> 2, bc = (gi & 3) * 4;                    // piece of a [4 pixels][16 channels] block this lane addresses

    f32x4 acc[9][TPW];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < TPW; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long b = tile;
        const int tx = (int)(b % tiles_x); b /= tiles_x;
        const int ty = (int)(b % tiles_y); b /= tiles_y;
        const int img = (int)b, r0 = ty * R, c0 = tx * TW;
        __syncthreads();                                           // previous tile's reads are done
        stage_halo<CS, 1>(smem, x, img, r0, c0, R, TW, H, W, wave, lane);
        for (int p = tid; p < R * TW; p += 256) {
            const int r = p / TW, cc = p - r * TW;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r0 + r < H) v = *reinterpret_cast<const u32x4*>(dz + (((long)img * H + r0 + r) * W + c0 + cc) * 8);
            *reinterpret_cast<u32x4*>(dzs + p * 16) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int ksteps = R * TW / 32;
        for (int ks = 0; ks < ksteps; ++ks) {
            const int r = (ks * 32) / TW, x0 = (ks * 32) % TW + quad * 8;       // this lane's 8 pixels: row r, columns x0..x0+7
            // dz^T fragment: row = output channel gi (< 8 stored), 8 consecutive pixels
            u32x4 dfrag = {0u, 0u, 0u, 0u};
            if (gi < 8) {
                const unsigned char* p = dzs + (r * TW + x0) * 16 + gi * 2;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dfrag[e] = (uint32_t)*reinterpret_cast<const uint16_t*>(p + (2 * e) * 16) |
                               ((uint32_t)*reinterpret_cast<const uint16_t*>(p + (2 * e + 1) * 16) << 16);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int prow = r + t / 3, pc0 = x0 + t % 3 + br;          // halo coordinates of pixel (x0 + br) under this tap
                const unsigned char* rowb = smem + (prow * (TW + 2)) * PIXB;
#pragma unroll
                for (int c = 0; c < TPW; ++c) {
                    const int ch = (wave * TPW + c) * 16 + bc;
                    const int pca = pc0, pcb = pc0 + 4;
                    const unsigned char* pa = rowb + pca * PIXB + swz<CS, 1>(ch >> 3, pca) * 16 + (ch & 7) * 2;
                    const unsigned char* pb = rowb + pcb * PIXB + swz<CS, 1>(ch >> 3, pcb) * 16 + (ch & 7) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)pa);
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)pb);
                    const u32x4 xf = {((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                                      ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16)};
                    acc[t][c] = mfma_bf16(dfrag, xf, acc[t][c]);
                }
            }
        }
    }
    // lane holds output channels o = quad*4 + e (only quad 0 is real), input channel (wave*TPW + c)*16 + gi
    if (quad == 0) {
        float* slab = slabs + (long)blockIdx.x * Cout * 9 * CS;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < TPW; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < Cout) slab[((long)e * 9 + t) * CS + (wave * TPW + c) * 16 + gi] = acc[t][c][e];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// weight gradient, 64-wide images, second formulation:  G[o][tap][c] = sum_q dz[q - tap][o] x[q][c].  The WIDE tensor x is
// the unshifted MFMA column operand - staged without a halo, every pixel read from HBM and from LDS exactly once per
// k-step - and the shifts move to the narrow dz: the MFMA rows are the (tap, o) pairs (27 of 32 rows for the RGB layer, 9
// of 16 for the segmentation layer) whose k-fragments are 2-byte gathers from a 16-byte-per-pixel dz tile with a zero
// halo. 2*8 MFMAs and 16 transposing reads per 32 pixels instead of 9*8 and 144. Persistent blocks, one slab per block.
// ------------------------------------------------------------------------------------------------------------------
constexpr int DZ_PITCH = 75;                                  // 66 columns (+-1 halo) + one pad slot per 8 (bank skew)
__device__ __forceinline__ int dz_slot(int q) { return q + (q >> 3); }

template <int CS>
__global__ __launch_bounds__(256) void thin3x3_wgrad_rows_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x, float* __restrict__ slabs,
                                                                 int N, int H, int Cout, int RB) {
    constexpr int W = 64, PIXB = CS * 2, NCH = CS / 8, NT = CS / 16, TPW = NT / 4, ROWB = W * PIXB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* xs = smem;                                     // [RB][64 px][PIXB], chunks swizzled for transposing reads
    unsigned char* dzs = smem + RB * ROWB;                        // [RB + 2][DZ_PITCH][16 B], zero halo
    const int units_y = (H + RB - 1) / RB;
    const long units = (long)N * units_y;
    const int gi = lane & 15, quad = lane >> 4;
    const int br = gi >> 2, bc = (gi & 3) * 4;
    const int nrows = 9 * Cout, RT = (nrows + 15) / 16;           // (tap, o) row tiles: 2 for Cout = 3, 1 for Cout = 1

    f32x4 acc[3][TPW];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int ct = 0; ct < TPW; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's row of each row tile: (tap, o) -> byte offset of dz[-tap][o] relative to the pixel's own halo position
    int rsy[3], rsx[3], rch[3];
    bool rok[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        const int idx = rt * 16 + gi;
        rok[rt] = rt < RT && idx < nrows;
        const int tap = rok[rt] ? idx / Cout : 0;
        rch[rt] = rok[rt] ? idx - tap * Cout : 0;
        rsy[rt] = 2 - tap / 3;                                     // halo coordinates of dz[q - tap]: (row + 2 - dy, column + 2 - dx)
        rsx[rt] = 2 - tap % 3;
    }

    for (long u = blockIdx.x; u < units; u += gridDim.x) {
        const int img = (int)(u / units_y), r0 = (int)(u % units_y) * RB;
        __syncthreads();                                           // the previous unit's reads are done
        for (int i = wave; i < RB * NCH; i += 4) {                 // x rows by LDS-DMA (NCH instructions per row)
            const int ci = i * 64 + lane;
            const int pi = ci / NCH, phys = ci - pi * NCH;
            const int row = pi >> 6, col = pi & 63;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(t_zero_page);
            if (r0 + row < H)
                src = reinterpret_cast<const unsigned char*>(x + (((long)img * H + r0 + row) * W + col) * CS + swz<CS, 1>(phys, col) * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xs + i * 1024), 16, 0, 0);
        }
        for (int e = tid; e < (RB + 2) * DZ_PITCH; e += 256) {     // dz rows r0-1 .. r0+RB with the zero halo
            const int r = e / DZ_PITCH, sl = e - r * DZ_PITCH;
            const int q = sl - sl / 9;                             // inverse of dz_slot (pad slots fail the check below)
            u32x4 v = {0u, 0u, 0u, 0u};
            const int y = r0 - 1 + r, px = q - 1;
            if (dz_slot(q) == sl && (unsigned)y < (unsigned)H && (unsigned)px < (unsigned)W)
                v = *reinterpret_cast<const u32x4*>(dz + (((long)img * H + y) * W + px) * 8);
            reinterpret_cast<u32x4*>(dzs)[e] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ks = 0; ks < RB * 2; ++ks) {                      // 32 pixels per k-step: half a row
            const int r = ks >> 1, x0 = (ks & 1) * 32 + quad * 8;  // this lane's 8 pixels: row r0 + r, columns x0 .. x0+7
            if (r0 + r >= H) break;
            u32x4 bf[TPW];
#pragma unroll
            for (int ct = 0; ct < TPW; ++ct) {
                const int ch = (wave * TPW + ct) * 16 + bc;
                const int pa = x0 + br, pb = pa + 4;
                const unsigned char* rowb = xs + r * ROWB;
                const unsigned char* a0 = rowb + pa * PIXB + swz<CS, 1>(ch >> 3, pa) * 16 + (ch & 7) * 2;
                const unsigned char* a1 = rowb + pb * PIXB + swz<CS, 1>(ch >> 3, pb) * 16 + (ch & 7) * 2;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a0);
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a1);
                bf[ct] = u32x4{((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                               ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16)};
            }
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                if (rt >= RT) break;
                // row (tap, o): dz at pixel q - tap -> halo coordinates (r + 2 - dy, column + 2 - dx)
                const int sx = rsx[rt];
                const unsigned char* dr = dzs + ((r + rsy[rt]) * DZ_PITCH) * 16 + rch[rt] * 2;
                u32x4 af = {0u, 0u, 0u, 0u};
                if (rok[rt]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c0 = x0 + 2 * e + sx, c1 = c0 + 1;
                        af[e] = (uint32_t)*reinterpret_cast<const uint16_t*>(dr + dz_slot(c0) * 16) |
                                ((uint32_t)*reinterpret_cast<const uint16_t*>(dr + dz_slot(c1) * 16) << 16);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < TPW; ++ct) acc[rt][ct] = mfma_bf16(af, bf[ct], acc[rt][ct]);
            }
        }
    }
    // lane holds rows quad*4 + e of its row tiles ((tap, o) = divmod(rt*16 + quad*4 + e, Cout)), column = channel gi of the tile
    float* slab = slabs + (long)blockIdx.x * Cout * 9 * CS;
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
        if (rt >= RT) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = rt * 16 + quad * 4 + e;
            if (idx >= nrows) continue;
            const int tap = idx / Cout, o = idx - tap * Cout;
#pragma unroll
            for (int ct = 0; ct < TPW; ++ct) slab[((long)o * 9 + tap) * CS + (wave * TPW + ct) * 16 + gi] = acc[rt][ct][e];
        }
    }
}

// G[i] += sum_b slabs[b][i] in a FIXED order: block = 32 elements x 8 slab lanes (lane k adds slabs k, k+8, ...), the 8
// partial sums are combined in lane order through LDS.
__global__ __launch_bounds__(256) void thin_slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, float* __restrict__ G, int n) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, k = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    float s = 0.f;
    if (i < n)
        for (int b = k; b < nslabs; b += 64) {        // eight slabs in flight per lane (one per trip: 60-120 us for 512 slabs)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = b + 8 * j < nslabs ? slabs[(long)(b + 8 * j) * n + i] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
    part[k][e] = s;
    __syncthreads();
    if (k == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][e];
        G[i] += t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// critic first conv: 4x4, stride 2, pad 1 over an 8-stored-channel image; K = 16 taps x 8 = 4 MFMA k-steps (k-step = kernel
// row ky, lane quad = kx). Input pixels are 16 bytes: the fragment loads go straight to the vector L1 (no LDS).
// ------------------------------------------------------------------------------------------------------------------
template <int CSO, bool SMOOTH>
__global__ __launch_bounds__(256) void thin4x4s2_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                            const float* __restrict__ alpha_p, int H, int W, int Cout, int act, long ngroups) {
    constexpr int NT = CSO / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, quad = lane >> 4;
    const int OH = H / 2, OW = W / 2;
    // weight row n of column tile j <-> output channel ((j>>1)*4 + (n>>2))*8 + (j&1)*4 + (n&3): the pair of tiles (2s, 2s+1)
    // then gives lane quad q the 8 consecutive channels of 16-byte piece s*4+q, so ONE store instruction writes a
    // contiguous 64-byte run per pixel
    // the block fetches the 32 KB of weight fragments ONCE into LDS (fragment-major, lane-linear) and every wavefront copies
    // them to its registers from there: loading them per wavefront straight from L2 was 4x the bytes of the whole input
    __shared__ u32x4 wsm[NT * 4 * 64];
    for (int e = threadIdx.x; e < NT * 4 * 64; e += blockDim.x) {
        const int ln = e & 63, f = e >> 6, j = f >> 2, ks = f & 3;
        const int nn = ln & 15, qq = ln >> 4;
        const int ch = ((j >> 1) * 4 + (nn >> 2)) * 8 + (j & 1) * 4 + (nn & 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(w + (long)(ch < Cout ? ch : 0) * 128 + (ks * 4 + qq) * 8);
        wsm[e] = ch < Cout ? v : u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    u32x4 bw[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bw[j][ks] = wsm[(j * 4 + ks) * 64 + lane];
    const float alpha = alpha_p ? *alpha_p : 1.f;
    const ActPl apl = act_pl(act);                     // (no run-time activation switch in the element loop)
    const int gpr = OW / 16;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // (wave-uniform 32-bit index arithmetic, see thin3x3_dgrad_kernel)
    for (int gi = blockIdx.x * 4 + wave_u; gi < (int)ngroups; gi += gridDim.x * 4) {
        const int cg = gi % gpr;
        const int rowo_i = gi / gpr;
        const long rowo = rowo_i;                       // img*OH + oy
        const int oy = rowo_i % OH;
        const long img = rowo_i / OH;
        const int ox = cg * 16 + n;
        u32x4 a[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int sy = 2 * oy + ks - 1, sx = 2 * ox + quad - 1;
            a[ks] = ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
                        ? *reinterpret_cast<const u32x4*>(x + ((img * H + sy) * W + sx) * 8) : u32x4{0u, 0u, 0u, 0u};
        }
        bf16_t* out = y + (rowo * OW + ox) * CSO;
#pragma unroll
        for (int j = 0; j < NT; j += 2) {
            const int chb = ((j >> 1) * 4 + quad) * 8;                    // first of this lane's 8 channels in tile pair j
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                c0 = mfma_bf16(bw[j][ks], a[ks], c0);
                c1 = mfma_bf16(bw[j + 1][ks], a[ks], c1);
            }
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = (chb + e < Cout) ? act_apply_t<SMOOTH>(c0[e] * alpha, act, apl) : 0.f;
                v[4 + e] = (chb + 4 + e < Cout) ? act_apply_t<SMOOTH>(c1[e] * alpha, act, apl) : 0.f;
            }
            u32x4 pk = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            *reinterpret_cast<u32x4*>(out + chb) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// critic first conv, weight gradient: G[o][tap*8 + c] += sum_p dz[p][o] x[2p + tap][c]   (64-wide images: an output row is
// exactly one 32-pixel MFMA k-step). Through the gather-GEMM this ran 16 taps x (Cout x 8) tiles that each re-read dz:
// 28 TFLOP/s. Here a block owns RB output rows of one image: its dz rows go to LDS once (LDS-DMA, transposing reads give
// the k-contiguous o-fragments), the 2*RB+2 input rows (16 bytes per pixel) too, and ALL 16 taps x 8 channels form the
// MFMA row dimension (8 row tiles of two taps each), so dz is read once. One fp32 slab [Cout][128] per block; the slabs
// are summed in a fixed order by thin_slab_reduce_kernel (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------------------------
constexpr int X4_PITCH = 70;                                  // 66 pixel columns (+-1 halo) + one pad slot per 16 (bank skew)
__device__ __forceinline__ int x4_slot(int q) { return q + (q >> 4); }

__global__ __launch_bounds__(256) void thin4x4s2_wgrad_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x, float* __restrict__ slabs,
                                                              int H, int Cout, int RB) {
    constexpr int W = 64, OW = 32, CS = 128, PIXB = CS * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int OH = H / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int units = OH / RB;
    const int img = blockIdx.x / units, oy0 = (blockIdx.x % units) * RB;
    unsigned char* dzs = smem;                                    // [RB][32 px][256 B], chunks swizzled for transposing reads
    unsigned char* xs = smem + RB * OW * PIXB;                    // [2*RB + 2][X4_PITCH][16 B], zero borders
    const int xrows = 2 * RB + 2;

    // ---- stage: dz rows by LDS-DMA (RB * 8 instructions), x rows through registers (zero outside the image)
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(t_zero_page);
    for (int i = wave; i < RB * 8; i += 4) {
        const int ci = i * 64 + lane;
        const int pi = ci >> 4, phys = ci & 15;                   // pixel of the tile, physical 16-byte slot
        const int row = pi >> 5, col = pi & 31;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(
            dz + (((long)img * OH + oy0 + row) * OW + col) * CS + swz<CS, 1>(phys, col) * 8);
        (void)zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dzs + i * 1024), 16, 0, 0);
    }
    for (int e = tid; e < xrows * X4_PITCH; e += 256) reinterpret_cast<u32x4*>(xs)[e] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int e = tid; e < xrows * W; e += 256) {
        const int r = e >> 6, px = e & 63;
        const int y = 2 * oy0 - 1 + r;
        if ((unsigned)y < (unsigned)H)
            reinterpret_cast<u32x4*>(xs)[r * X4_PITCH + x4_slot(px + 1)] =
                *reinterpret_cast<const u32x4*>(x + (((long)img * H + y) * W + px) * 8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int gi = lane & 15, quad = lane >> 4;
    const int br = gi >> 2, bc = (gi & 3) * 4;                    // transposing-read piece: pixel br of 4, channels bc..bc+3 of 16
    const int tsel = gi >> 3, c = gi & 7;                         // row operand: row gi = (tap parity, input channel)
    f32x4 acc[8][2];
#pragma unroll
    for (int tp = 0; tp < 8; ++tp)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) acc[tp][ot] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int r = 0; r < RB; ++r) {
        // column operand: dz^T fragments of this wavefront's two 16-channel tiles, pixels quad*8 .. +7 of output row r
        u32x4 bf[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const int ch = (wave * 2 + ot) * 16 + bc;
            const int pa = quad * 8 + br, pb = pa + 4;
            const unsigned char* rowb = dzs + r * OW * PIXB;
            const unsigned char* a0 = rowb + pa * PIXB + swz<CS, 1>(ch >> 3, pa) * 16 + (ch & 7) * 2;
            const unsigned char* a1 = rowb + pb * PIXB + swz<CS, 1>(ch >> 3, pb) * 16 + (ch & 7) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a0);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a1);
            bf[ot] = u32x4{((uint32_t)(uint16_t)lo[0]) | ((uint32_t)(uint16_t)lo[1] << 16), ((uint32_t)(uint16_t)lo[2]) | ((uint32_t)(uint16_t)lo[3] << 16),
                           ((uint32_t)(uint16_t)hi[0]) | ((uint32_t)(uint16_t)hi[1] << 16), ((uint32_t)(uint16_t)hi[2]) | ((uint32_t)(uint16_t)hi[3] << 16)};
        }
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) {
            // row operand: row gi = (tap 2*tp + tsel, channel c), 8 consecutive output pixels -> every second input pixel
            const int tap = 2 * tp + tsel, ky = tap >> 2, kx = tap & 3;
            const unsigned char* xr = xs + ((2 * r + ky) * X4_PITCH) * 16 + c * 2;
            u32x4 af;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q0 = 16 * quad + 2 * (2 * e) + kx, q1 = q0 + 2;          // q = 2*ox + kx (halo offset +1 folded in)
                af[e] = (uint32_t)*reinterpret_cast<const uint16_t*>(xr + x4_slot(q0) * 16) |
                        ((uint32_t)*reinterpret_cast<const uint16_t*>(xr + x4_slot(q1) * 16) << 16);
            }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) acc[tp][ot] = mfma_bf16(af, bf[ot], acc[tp][ot]);
        }
    }
    // lane holds rows quad*4 + e (tap 2*tp + (row >> 3), channel row & 7) of column gi = output channel of the tile
    float* slab = slabs + (long)blockIdx.x * Cout * 128;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        const int o = (wave * 2 + ot) * 16 + gi;
        if (o >= Cout) continue;
#pragma unroll
        for (int tp = 0; tp < 8; ++tp)
            *reinterpret_cast<f32x4*>(slab + (long)o * 128 + tp * 16 + quad * 4) = acc[tp][ot];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// critic first conv, data gradient (the generator step's d loss / d fake image; 64-wide images):
//   dx[y][x][c] = alpha * sum over the 4 taps (ky, kx) with y+1-ky, x+1-kx even, and o:  dz[(y+1-ky)/2][(x+1-kx)/2][o] w[o][ky*4+kx][c]
// A wavefront owns one pixel-parity class (y&1, x&1) - its 4 taps x 128 channels are K = 512, its weight fragments (rows = the
// 8 stored input channels) stay in registers - and a block stages RA+2 rows of dz (with a zero halo) in LDS once for the
// 2*RA output rows they reach: dz crosses HBM -> LDS once instead of once per tap through the gather-GEMM's vector L1 path.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void thin4x4s2_dgrad_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ wb, bf16_t* __restrict__ dx,
                                                              const float* __restrict__ alpha_p, int H, int RA) {
    constexpr int OW = 32, CS = 128, PIXB = CS * 2, TWH = OW + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [(RA + 2)][34][PIXB], stage_halo<CS, 0> image
    const int OH = H / 2, W = 2 * OW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int units = (OH + RA - 1) / RA;
    const int img = blockIdx.x / units, a0 = (blockIdx.x % units) * RA;
    stage_halo<CS, 0>(smem, dz, img, a0, 0, RA, OW, OH, OW, wave, lane);
    const int n = lane & 15, quad = lane >> 4;
    const int py = wave >> 1, px = wave & 1;
    u32x4 aw[4][4];                                               // [tap (j, i)][32-channel chunk]: rows = input channel n (< 8)
#pragma unroll
    for (int ji = 0; ji < 4; ++ji) {
        const int tap = (1 - py + 2 * (ji >> 1)) * 4 + (1 - px + 2 * (ji & 1));
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(wb + (long)(n & 7) * (16 * CS) + tap * CS + kc * 32 + quad * 8);
            aw[ji][kc] = n < 8 ? v : u32x4{0u, 0u, 0u, 0u};
        }
    }
    const float alpha = alpha_p ? *alpha_p : 1.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int a = 0; a < RA; ++a) {
        if (a0 + a >= OH) break;
#pragma unroll
        for (int bg = 0; bg < 2; ++bg) {                          // 16 output pixels of this class: b = bg*16 + n
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ji = 0; ji < 4; ++ji) {
                const int hr = a + py - (ji >> 1) + 1, hc = bg * 16 + n + px - (ji & 1) + 1;     // halo coordinates of the dz pixel
                const unsigned char* pb = smem + (hr * TWH + hc) * PIXB;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const u32x4 b = *reinterpret_cast<const u32x4*>(pb + swz<CS, 0>(kc * 4 + quad, hc) * 16);
                    acc[kc & 1] = mfma_bf16(aw[ji][kc], b, acc[kc & 1]);
                }
            }
            if (quad < 2) {                                       // rows quad*4 + e = input channels; 8 stored channels per pixel
                const int y = 2 * (a0 + a) + py, xx = 2 * (bg * 16 + n) + px;
                u32x2 pk = {pack2((acc[0][0] + acc[1][0]) * alpha, (acc[0][1] + acc[1][1]) * alpha),
                            pack2((acc[0][2] + acc[1][2]) * alpha, (acc[0][3] + acc[1][3]) * alpha)};
                *reinterpret_cast<u32x2*>(dx + (((long)img * H + y) * W + xx) * 8 + quad * 4) = pk;
            }
        }
    }
}

// tile rows per block: as many as fit in ~72 KB of LDS (two blocks per CU), at least 1
inline int rows_for(int Cs, int TW, int extra_per_row) {
    int R = 8;
    while (R > 1 && (R + 2) * (TW + 2) * Cs * 2 + R * extra_per_row > 72 * 1024) --R;
    return R;
}

}  // namespace

extern "C" int cpcsv_thin_supported(int kind, int Cs, int Cout, int H, int W) {
    if (kind == 0) return (Cs == 64 || Cs == 128) && Cout >= 1 && Cout <= 4 && W % 32 == 0 && H >= 1;     // 3x3 s1 p1
    // 4x4 s2 p1: the kernels hard-code a 128-channel stored row, so pad8(Cout) must BE 128 (Cout 121..128): with Cout 113..120 the
    // caller's buffers have a 120-channel stride and the kernels would write out of bounds
    if (kind == 1) return Cs == 8 && ((Cout + 7) & ~7) == 128 && H % 2 == 0 && W % 32 == 0;
    return 0;
}

extern "C" int cpcsv_thin3x3_fwd(const void* x, const void* w_fwd, void* y, int N, int H, int W, int Cs, int Cout, int act,
                                 void* stream) {
    if (!x || !w_fwd || !y || !cpcsv_thin_supported(0, Cs, Cout, H, W)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    static const int probe = [] { const char* e = getenv("CPCSV_THIN_PROBE"); return e ? atoi(e) : 0; }();     // tools only
    static const int force_r = [] { const char* e = getenv("CPCSV_THIN_R"); return e ? atoi(e) : 0; }();
    static const int force_tw = [] { const char* e = getenv("CPCSV_THIN_TW"); return e ? atoi(e) : 0; }();
    // 128-channel input: taps-as-rows kernel (21.8 us at N=60 against 24.6 for the rolling kernel); 64-channel input: the
    // rolling kernel (13.1 against 17.8 - the per-row P exchange costs the same for half the bytes). CPCSV_THIN_TAPS=0/1/2: A/B.
    static const int taps_form = [] { const char* e = getenv("CPCSV_THIN_TAPS"); return e ? atoi(e) : 1; }();
    if (taps_form && W == 64 && H % 16 == 0 && (Cs == 128 || taps_form == 2) && !force_r && !force_tw && !probe) {
        const unsigned grid = (unsigned)((long)N * (H / 16));
        if (Cs == 128) hipLaunchKernelGGL((thin3x3_fwd_taps_kernel<128, 16, 8>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, H, Cout, act);
        else hipLaunchKernelGGL((thin3x3_fwd_taps_kernel<64, 16, 8>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, H, Cout, act);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    static const int roll = [] { const char* e = getenv("CPCSV_THIN_ROLL"); return e ? atoi(e) : 16; }();      // span; 0: tile kernel
    if (roll > 0 && W == 64 && H % roll == 0 && roll % 2 == 0 && !force_r && !force_tw && !probe) {
        const int lds = 8 * 64 * Cs * 2 + Cs * 2;
        const unsigned grid = (unsigned)((long)N * (H / roll));
        static long long* dbgp = [] { const char* e = getenv("CPCSV_THIN_DBG_PTR"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();   // tools only
        if (Cs == 128) {
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_fwd_roll_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (once != hipSuccess) return -1100 - (int)once;
            hipLaunchKernelGGL(thin3x3_fwd_roll_kernel<128>, dim3(grid), dim3(512), lds, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, H, Cout, act, roll, dbgp);
        } else {
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_fwd_roll_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (once != hipSuccess) return -1100 - (int)once;
            hipLaunchKernelGGL(thin3x3_fwd_roll_kernel<64>, dim3(grid), dim3(512), lds, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, H, Cout, act, roll, dbgp);
        }
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    const int TW = force_tw ? force_tw : (W < 64 ? W : 64);
    if (W % TW) return -1002;
    const int R = force_r ? force_r : rows_for(Cs, TW, 0);
    const int lds = (((R + 2) * (TW + 2) * Cs * 2 + 1023) / 1024) * 1024;
    const long ntiles = (long)N * ((H + R - 1) / R) * (W / TW);
    constexpr int force_grid = 0;
    const long cap = force_grid ? force_grid : 512;                   // two persistent blocks per CU
    const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    if (Cs == 128) {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_fwd_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (once != hipSuccess) return -1100 - (int)once;
        hipLaunchKernelGGL(thin3x3_fwd_kernel<128>, dim3(grid), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, H, W, Cout, act, R, TW, probe, ntiles);
    } else {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (once != hipSuccess) return -1100 - (int)once;
        hipLaunchKernelGGL(thin3x3_fwd_kernel<64>, dim3(grid), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, H, W, Cout, act, R, TW, probe, ntiles);
    }
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_thin3x3_dgrad(const void* dz, const void* w_bwd, void* dx, int N, int H, int W, int Cs, int Cout, void* stream) {
    if (!dz || !w_bwd || !dx || !cpcsv_thin_supported(0, Cs, Cout, H, W)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const long ngroups = (long)N * H * W / 16;
    if (ngroups >= (1L << 31)) return -1001;            // (32-bit group index in the kernel)
    const unsigned grid = (unsigned)(ngroups / 4 < 512 ? (ngroups + 3) / 4 : 512);     // persistent: the weight fragments load once per block
    if (Cs == 128) hipLaunchKernelGGL(thin3x3_dgrad_kernel<128>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)w_bwd, (bf16_t*)dx, H, W, ngroups);
    else hipLaunchKernelGGL(thin3x3_dgrad_kernel<64>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)w_bwd, (bf16_t*)dx, H, W, ngroups);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

static inline int rows_rb(int Cs) { return Cs == 128 ? 4 : 8; }      // x rows per unit of the rows-formulation kernel: 64 KB of LDS
static const int g_wgrad_rows = [] { const char* e = getenv("CPCSV_THIN_WGRAD_ROWS"); return e ? atoi(e) : 1; }();   // A/B switch

extern "C" int cpcsv_thin3x3_wgrad_slabs(int N, int H, int W, int Cs) {
    if (W == 64 && g_wgrad_rows) {
        const long units = (long)N * ((H + rows_rb(Cs) - 1) / rows_rb(Cs));
        return (int)(units < 512 ? units : 512);               // two persistent blocks per CU
    }
    const int TW = W < 64 ? W : 64;
    const int R = rows_for(Cs, TW, TW * 16);
    const long ntiles = (long)N * ((H + R - 1) / R) * (W / TW);
    return (int)(ntiles < 256 ? ntiles : 256);      // one persistent block per CU
}

extern "C" int cpcsv_thin3x3_wgrad(const void* dz, const void* x, float* G, float* slabs, int N, int H, int W, int Cs, int Cout,
                                   void* stream) {
    if (!dz || !x || !G || !slabs || !cpcsv_thin_supported(0, Cs, Cout, H, W)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    if (W == 64 && g_wgrad_rows) {
        const int RB = rows_rb(Cs), nslabs = cpcsv_thin3x3_wgrad_slabs(N, H, W, Cs);
        const int lds = RB * 64 * Cs * 2 + (RB + 2) * DZ_PITCH * 16;
        if (Cs == 128) {
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_wgrad_rows_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (once != hipSuccess) return -1100 - (int)once;
            hipLaunchKernelGGL(thin3x3_wgrad_rows_kernel<128>, dim3(nslabs), dim3(256), lds, s, (const bf16_t*)dz, (const bf16_t*)x, slabs, N, H, Cout, RB);
        } else {
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_wgrad_rows_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (once != hipSuccess) return -1100 - (int)once;
            hipLaunchKernelGGL(thin3x3_wgrad_rows_kernel<64>, dim3(nslabs), dim3(256), lds, s, (const bf16_t*)dz, (const bf16_t*)x, slabs, N, H, Cout, RB);
        }
        CPCSV_CHECK_LAUNCH();
        const int n = Cout * 9 * Cs;
        hipLaunchKernelGGL(thin_slab_reduce_kernel, dim3(cdiv(n, 32)), dim3(256), 0, s, slabs, nslabs, G, n);
        CPCSV_CHECK_LAUNCH();
        return 0;
    }
    const int TW = W < 64 ? W : 64;
    if (W % TW || TW % 32) return -1002;
    const int R = rows_for(Cs, TW, TW * 16);
    const int lds = (((R + 2) * (TW + 2) * Cs * 2 + 1023) / 1024) * 1024 + R * TW * 16;
    const int nslabs = cpcsv_thin3x3_wgrad_slabs(N, H, W, Cs);
    if (Cs == 128) {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_wgrad_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (once != hipSuccess) return -1100 - (int)once;
        hipLaunchKernelGGL(thin3x3_wgrad_kernel<128>, dim3(nslabs), dim3(256), lds, s, (const bf16_t*)dz, (const bf16_t*)x, slabs, N, H, W, Cout, R, TW);
    } else {
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin3x3_wgrad_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (once != hipSuccess) return -1100 - (int)once;
        hipLaunchKernelGGL(thin3x3_wgrad_kernel<64>, dim3(nslabs), dim3(256), lds, s, (const bf16_t*)dz, (const bf16_t*)x, slabs, N, H, W, Cout, R, TW);
    }
    CPCSV_CHECK_LAUNCH();
    const int n = Cout * 9 * Cs;
    hipLaunchKernelGGL(thin_slab_reduce_kernel, dim3(cdiv(n, 32)), dim3(256), 0, s, slabs, nslabs, G, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_thin4x4s2_dgrad(const void* dz, const void* w_bwd, void* dx, const float* alpha, int N, int H, int W, void* stream) {
    if (!dz || !w_bwd || !dx || W != 64 || H % 2) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const int RA = 6, OH = H / 2;
    const int lds = (((RA + 2) * 34 * 256 + 1023) / 1024) * 1024;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin4x4s2_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (once != hipSuccess) return -1100 - (int)once;
    hipLaunchKernelGGL(thin4x4s2_dgrad_kernel, dim3((unsigned)((long)N * ((OH + RA - 1) / RA))), dim3(256), lds, s, (const bf16_t*)dz, (const bf16_t*)w_bwd,
                       (bf16_t*)dx, alpha, H, RA);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_thin4x4s2_wgrad_slabs(int N, int H, int W) {
    if (W != 64 || H % 16) return 0;                              // 0: shape not served, use cpcsv_wgrad_tn
    return N * ((H / 2) / 8);
}

extern "C" int cpcsv_thin4x4s2_wgrad(const void* dz, const void* x, float* G, float* slabs, int N, int H, int W, int Cout, void* stream) {
    if (!dz || !x || !G || !slabs || !cpcsv_thin_supported(1, 8, Cout, H, W) || cpcsv_thin4x4s2_wgrad_slabs(N, H, W) <= 0) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const int RB = 8;
    const int nslabs = cpcsv_thin4x4s2_wgrad_slabs(N, H, W);
    const int lds = RB * 32 * 256 + (2 * RB + 2) * X4_PITCH * 16;
    static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(thin4x4s2_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (once != hipSuccess) return -1100 - (int)once;
    hipLaunchKernelGGL(thin4x4s2_wgrad_kernel, dim3(nslabs), dim3(256), lds, s, (const bf16_t*)dz, (const bf16_t*)x, slabs, H, Cout, RB);
    CPCSV_CHECK_LAUNCH();
    const int n = Cout * 128;
    hipLaunchKernelGGL(thin_slab_reduce_kernel, dim3(cdiv(n, 32)), dim3(256), 0, s, slabs, nslabs, G, n);
    CPCSV_CHECK_LAUNCH();
    return 0;
}

extern "C" int cpcsv_thin4x4s2_fwd(const void* x, const void* w_fwd, void* y, const float* alpha, int N, int H, int W, int Cout,
                                   int act, void* stream) {
    if (!x || !w_fwd || !y || !cpcsv_thin_supported(1, 8, Cout, H, W)) return -1001;
    hipStream_t s = (hipStream_t)stream;
    const long ngroups = (long)N * (H / 2) * (W / 2) / 16;
    if (ngroups >= (1L << 31)) return -1001;            // (32-bit group index in the kernel)
    constexpr int g4 = 512;      // sweeps: 256 21.8 us, 512 15.7, 768 18.4 (knob retired)
    const unsigned grid = (unsigned)(ngroups / 4 < g4 ? (ngroups + 3) / 4 : g4);     // persistent: 128 registers of weight fragments per lane
    if (act >= CPCSV_ACT_TANH)
        hipLaunchKernelGGL((thin4x4s2_fwd_kernel<128, true>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, alpha, H, W,
                           Cout, act, ngroups);
    else
        hipLaunchKernelGGL((thin4x4s2_fwd_kernel<128, false>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w_fwd, (bf16_t*)y, alpha, H, W,
                           Cout, act, ngroups);
    CPCSV_CHECK_LAUNCH();
    return 0;
}
