/* cpcsv_hip.h — C ABI of libcpcsv_hip.so: the MI355X (gfx950) kernels behind the CP-CSV
 * story-GAN training step.
 *
 * The reference (basiclab/CPCStoryVisualization-Pytorch) has NO native code and no FFI: every
 * device op is whatever PyTorch dispatches to. This library therefore replaces
 * LIBRARY CALLS made by the reference's Python; each entry point cites the reference call site
 * whose device work it performs. Binding stub: INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - plain pointers + ints only; all pointers are DEVICE pointers owned by the caller
 *     (PyTorch allocations); kernels never allocate, free or retain them.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream and
 *     never synchronises the device.
 *   - return 0 on success, negative on error (-(hipError_t) for launch errors, -1000-x for
 *     argument errors); never throws.
 *   - dtype: 0 = fp32, 1 = bf16 (storage of activations / packed weights; accumulation, batch
 *     statistics, losses, master weights and optimiser state are always fp32).
 *   - activations are NHWC with the channel count padded to a multiple of 8 ("Cs"); pad channels
 *     hold zeros. Matrices are row-major with a leading dimension that is a multiple of 8.
 */
#ifndef CPCSV_HIP_H
#define CPCSV_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPCSV_MAX_TAPS 16

/* activation codes (epilogues and BN-apply) */
#define CPCSV_ACT_NONE 0
#define CPCSV_ACT_RELU 1    /* nn.ReLU        model.py:33           */
#define CPCSV_ACT_LRELU 2   /* LeakyReLU(0.2) model.py:78,500        */
#define CPCSV_ACT_TANH 3    /* nn.Tanh        model.py:257,274,300   */
#define CPCSV_ACT_SIGMOID 4 /* nn.Sigmoid     model.py:80            */

typedef struct cpcsv_tap {
    int8_t oy, ox;  /* input offset of this tap relative to (y*sy, x*sx)          */
    uint8_t wtap;   /* which K-slice of the packed weight this tap multiplies      */
    uint8_t _pad;
} cpcsv_tap;

/* Gathered "NT" GEMM:  C[m][n] = sum_j sum_c A[pix(m, tap j)][c] * B[n][wtap_j*Cs + c]
 * One descriptor covers: Linear forward/dgrad (ntaps=1, 1x1 grid), conv forward (any kernel /
 * stride / padding, optional fused nearest x2 upsample of the input), conv dgrad (negated taps),
 * transposed-conv dgrad phases (scattered C), and the fused 2x2 sum-pool of an upsample's dgrad. */
typedef struct cpcsv_gemm_desc {
    const void* A;     /* activations, NHWC [imgs][IH][IW][Cs], dtype                      */
    const void* B;     /* packed weights [N][ldb], dtype, K-slices of Cs per tap           */
    void* C;           /* output rows, dtype (or fp32 if out_f32)                          */
    int dtype;
    int M, N;          /* output rows (imgs*MH*MW) and columns                             */
    int Cs;            /* stored channels per input pixel = K extent of one tap            */
    int ldb, ldc;
    int ntaps;
    cpcsv_tap taps[CPCSV_MAX_TAPS];
    int MH, MW;        /* per-image grid the rows m enumerate                              */
    int IH, IW;        /* stored input height/width                                        */
    int sy, sx;        /* input stride per grid step                                       */
    int up_shift;      /* 1: input is read through a nearest x2 upsample (model.py:29)     */
    int pool_rows;     /* 1: rows are ordered (img,y2,x2,dy,dx) and the 4 rows of a 2x2
                             block are summed into ONE output row (dgrad of the upsample)  */
    int scatter;       /* 1: C row for m is pixel (y*osy+ooy, x*osx+oox) of an OHxOW image */
    int OH, OW, osy, osx, ooy, oox;
    const float* alpha;/* NULL or device scalar: accumulator is multiplied by *alpha before
                          bias (1/sigma of a spectral-normed weight: conv(x,W/s) = conv(x,W)/s) */
    const float* bias; /* [N] or NULL                                                      */
    int act;           /* CPCSV_ACT_*  applied after bias                                  */
    float* stats;      /* NULL or [Mtiles][2][ldstat] per-block column sum / sum of squares
                          of the pre-activation values (BatchNorm batch statistics)        */
    int ldstat;
    int out_f32;       /* 1: C is fp32 regardless of dtype                                 */
    int splitk;        /* >1: the K tiles are split over blockIdx.y; partial products are added
                          (fp32 atomics) into ws, then one epilogue pass applies alpha/bias/act,
                          casts into C and emits the BN partials. For few-tile / long-K shapes. */
    float* ws;         /* [splitk][ws_rows][ldws] fp32 workspace: K slice s writes its partial tile into
                          slab s with plain stores; no zeroing needed, deterministic summation order */
    int ldws;
    long ws_rows;      /* rows of one slab = output rows of the GEMM                           */
    int nphases;       /* >1: transposed-conv dgrad parity phases batched over blockIdx.z:
                          phase p uses taps[ph_tap0[p] .. +ph_ntaps[p]) and writes output pixel
                          (y*osy + ph_ooy[p], x*osx + ph_oox[p]); all phases share MH x MW    */
    int ph_tap0[4], ph_ntaps[4], ph_ooy[4], ph_oox[4];
    int order_m_fast;  /* block order: 0 = N tiles of one M tile adjacent (A panel shared in L2),
                          1 = M tiles of one N tile adjacent (B panel shared; weight-heavy layers)  */
    /* Row groups: several passes of ONE layer (same weights) in one launch - the real and the fake batch of a critic
     * tower, the real / wrong / fake batches of its head (miscc/utils.py:70-84), the story half and the image half of a
     * generator pass (model.py:348,426). The rows [grow[g], grow[g+1]) of A / C belong to pass g (whole images); no M tile
     * straddles a boundary (tile t of the launch = tile t - T_g of group g, T_g = sum of ceil(rows_h / tile) over h < g), so
     * every BatchNorm partial belongs to exactly one pass, and pass g is scaled by *galpha[g] (its own 1/sigma: the power
     * iteration advances between the passes) instead of *alpha. ngroups <= 1: one group [0, M). */
    int ngroups;
    int grow[5];
    const float* galpha[4];
    /* Per-element addend: NULL, or fp32 [output rows][ldadd] added to the accumulator after alpha and before bias / BatchNorm
     * statistics / activation (not with splitk > 1 on the slabs: the split-K epilogue pass adds it). D_GET_LOGITS (model.py:89-92)
     * tiles the condition vector over the 4x4 map: those 489 input channels are spatially constant, so their share of the 3x3
     * conv is a small dense product per sample and tap; cpcsv_cond_head_fwd is the production form of that factorisation. */
    const float* addend;
    int ldadd;
    /* K-loop order of the streaming main loop: 0 = the library's choice (taps outer), 1 = channel tiles outer / taps inner (returns
     * -1011 with up_shift != 0: the tap offsets do not separate from the pixel there), 2 = taps outer / channel tiles inner.
     * The patch-resident main loop (input patch of one channel tile staged in LDS once, all taps of the phase served from it)
     * always walks K in order 1: the two are bit-identical in that order. In order 1 consecutive K tiles re-read the same input
     * lines one tap further, so the 4 / 9 / 16 reads of every input pixel hit in L2 instead of arriving a channel sweep apart:
     * 40 % less fabric read traffic for the family and +0.2 ... +0.3 ms per step - measured, not the default. */
    int korder;
    int wstride;       /* 0, or the element distance between the K slices of consecutive weight taps in B when it is not Cs: the A
                          operand then holds only the first Cs channels of a wider layer (the feature channels of D_GET_LOGITS'
                          concatenated input, model.py:89-92) while B keeps the layer's full packed rows */
    int patch;         /* 0: the library picks the patch-resident main loop where the geometry allows it; -1: never (A/B runs,
                          bit-identity tests); 1: require it (returns -1010 if the geometry does not allow it) */
    int slabs_only;    /* 1 (with splitk > 1): stop after the K-slice slabs ws[splitk][ws_rows][ldws] are written - no epilogue pass,
                          C / alpha / bias / act / stats unused; the caller's own kernel sums the slabs (cpcsv_cond_head_fwd) */
    int bcol_rows;     /* 0, or: output column n multiplies B row (n % bcol_rows) at the extra K offset (n / bcol_rows) * bcol_koff
                          elements - ONE launch then computes the products of A with several K windows of the same weight rows side
                          by side (the 9 taps' condition-channel slices of D_GET_LOGITS' 3x3 conv: column (tap, o), N = 9 * Cout) */
    int bcol_koff;
} cpcsv_gemm_desc;

/* rows covered by one stats partial (the kernel's M tile, or the epilogue pass's row tile when
 * splitk > 1); Mtiles = ceil(out_rows / this) */
int cpcsv_gemm_mtile(const cpcsv_gemm_desc* d);
/* columns of the block tile the kernel will use for this shape (for split-K planning on the host) */
int cpcsv_gemm_ntile(const cpcsv_gemm_desc* d);
/* replaces: F.linear / F.conv2d forward + cudnn dgrad behind model.py:16-34,44,75-80,250-308,
 * 499-520 and their autograd backward-data passes. */
int cpcsv_gemm_nt(const cpcsv_gemm_desc* d, void* stream);

/* Weight gradient ("TN" GEMM, reduction over pixels, fp32 atomics for the pixel splits):
 *   dW[n][j*Cs + c] += sum_m dY[m][n] * X[pix(m, tap j)][c]      for j in [0, ntaps)
 * dW must be zero on entry (with splits == 1 the result is simply stored). Same gather geometry as the forward conv.
 * replaces: cudnn/MKL-DNN backward-weights behind every Conv2d/Linear on the path. */
typedef struct cpcsv_wgrad_desc {
    const void* dY;    /* [M][ldy] dtype                                                   */
    const void* X;     /* NHWC input of the forward conv, dtype                            */
    float* dW;         /* [N][lddw] fp32, lddw >= ntaps*Cs                                 */
    int dtype;
    int M, N, Cs, ldy, lddw;
    int ntaps;
    cpcsv_tap taps[CPCSV_MAX_TAPS];
    int MH, MW, IH, IW, sy, sx, up_shift;
    int splits;        /* number of pixel-range splits (>=1)                               */
    int dy_gather;     /* 0: dY row of pixel m is row m. 1: pixel m=(img,y,x) of the MHxMW grid reads dY
                          pixel (y*dy_sy + (tap._pad & 15), x*dy_sx + (tap._pad >> 4)) of a DYH x DYW map
                          (sub-pixel form of upsample+conv: each output parity is its own 2x2 conv) */
    int DYH, DYW, dy_sy, dy_sx;
    int legacy;        /* diagnostics: 1 = force the register-staged kernel instead of the LDS-DMA one */
    int accumulate;    /* 1: dW may already hold earlier calls' sums (deferred update, cpcsv_layer_update): always add */
    const float* alpha; /* device scalar multiplied into this call's contribution (1/sigma of a spectral-normed layer,
                           so that calls with different sigma can share one accumulator) or NULL */
    /* Two passes of the same layer in ONE launch (the story half and the image half of a generator pass, the real and the
     * fake pass of a critic tower): rows [0, M1) come from (dY, X), rows [M1, M) from (dY2, X2), same geometry, summed in
     * the accumulators - no second launch, no read-modify-write of dW. M1 = 0: single pass. bf16 LDS-DMA kernel only;
     * M1 must be a whole number of images and a multiple of 64 rows. */
    const void* dY2;
    const void* X2;
    int M1;
    int creal;         /* 0, or the number of REAL input channels (<= Cs): columns c >= creal of a tap are not written, so dW may
                          be a master-layout dense weight gradient [N][creal] (lddw = creal, ntaps = 1): no unpack launch */
    int wstride;       /* 0, or the element distance between the column blocks of consecutive weight taps in dW when it is not Cs:
                          X then holds only Cs channels of a wider layer whose accumulator keeps its full rows */
    int dy_tapstride;  /* 0, or: tap j reads the dY columns [j * dy_tapstride, j * dy_tapstride + N) of its rows instead of
                          [0, N) (ldy >= ntaps * dy_tapstride): every tap has its own pre-reduced dY (cpcsv_cond_head_bwd) */
} cpcsv_wgrad_desc;
int cpcsv_wgrad_tn(const cpcsv_wgrad_desc* d, void* stream);
/* tests / A-B timing: 0 = the bf16 LDS-DMA weight-gradient kernel always uses its general staging (per-piece gather
 * arithmetic); 1 (default) = linear running-pointer staging wherever the geometry allows. Returns the previous setting. */
int cpcsv_set_wgrad_linear(int on);

/* ---- weight packing (fp32 master [Cout][Cin][taps] -> operand layouts) ------------------- */
/* The packed K axis is S slices of Cin_s channels; slice sl takes master tap tapmap[sl] (NULL =
 * identity, S = taps; -1 = an all-zero slice). Outputs (each may be NULL), cast to dtype, pads zero:
 *   dst_fwd [Cout][S*Cin_s]      B operand of conv/linear forward and the layout wgrad writes
 *   dst_bwd [Cin][S*Cout_s]      B operand of conv dgrad (taps negated in the gather)
 *   dst_lin [S*Cin_s][Cout_s]    B operand of the dgrad of a full-window conv run as a Linear over
 *                                the flattened NHWC input (cate_classify model.py:520, outlogits.3 :79)
 * Spectral-normed layers pack W_orig; 1/sigma rides in the GEMM epilogue (alpha). */
int cpcsv_pack_weight(const float* w, void* dst_fwd, void* dst_bwd, void* dst_lin, int dtype, int Cout,
                      int Cin, int taps, int S, const int8_t* tapmap, int Cin_s, int Cout_s, void* stream);
/* inverse for gradients: dW master [Cout][Cin][taps] = G[Cout][S*Cin_s] * (1/sigma)
 *                                                     - coef * u[o] * v[i*taps+t]   (if sigma)
 * with coef = *gw_dot / sigma^2 (gw_dot = sum(G .* W), device scalar). u, v and gw_dot may all be NULL with sigma
 * set: the rank-1 term is then taken as 0 (exact behind a train-mode BatchNorm, where sum(G .* W) = 0).
 * Master taps that no slice maps to receive 0. accumulate!=0 adds into dw. */
int cpcsv_unpack_wgrad(float* G, float* dw, const float* sigma, const float* u, const float* v,
                       const float* gw_dot, int Cout, int Cin, int taps, int S, const int8_t* tapmap,
                       int Cin_s, int accumulate, int rezero, void* stream);
/* dw[o][k] -= (*gw / sigma[0]^2) * u[o] * v[k]  (dw = master weight gradient viewed [rows][cols]): the rank-1 term of ONE
 * call of a spectral-normed layer when the accumulator already holds contributions divided by that call's sigma
 * (cpcsv_bn_bwd_apply folds 1/sigma into dz; several calls with different sigma then share one cpcsv_unpack_wgrad). */
int cpcsv_rank1_sub(float* dw, const float* gw, const float* sigma, const float* u, const float* v, long rows, long cols,
                    void* stream);
/* rezero != 0: every entry of G that is read is written back as 0, so a persistent accumulator is
 * clean for the next cpcsv_wgrad_tn without a memset */
/* Summed-tap variants for the sub-pixel form of nearest-x2 upsample + 3x3 conv (model.py:26-34): slice sl of
 * the packed K axis holds the SUM of the master taps in bit-mask masks[sl] (bit t = tap t), because those taps
 * read the same low-resolution pixel; the weight gradient of master tap t is the sum of the slices containing it. */
int cpcsv_pack_weight_sum(const float* w, void* dst_fwd, void* dst_bwd, int dtype, int Cout, int Cin, int taps,
                          int S, const uint16_t* masks, int Cin_s, int Cout_s, void* stream);
int cpcsv_unpack_wgrad_sum(float* G, float* dw, int Cout, int Cin, int taps, int S, const uint16_t* masks,
                           int Cin_s, int accumulate, int rezero, void* stream);
/* gw_dot[0] = sum_{o,i,t} G[o][sl(t)*Cin_s+i] * w[o][i][t]   (fp32, zeroed by the call) */
int cpcsv_wgrad_dot(const float* G, const float* w, float* gw_dot, int Cout, int Cin, int taps, int S,
                    const int8_t* tapmap, int Cin_s, void* stream);

/* ---- spectral norm power iteration (torch.nn.utils.spectral_norm, model.py:5,19,79) ------- */
/* W viewed as [rows][cols]. If iterate: v <- normalize(W^T u); u <- normalize(W v) (eps 1e-12),
 * both updated in place; always: out[0] = sigma = u . (W v), out[1] = 1/sigma. If snapshot, the u and v this call
 * ended with are also copied to out[2 .. 2+rows) and out[2+rows .. 2+rows+cols) (the backward pass of THIS call
 * needs them; the layer's own u/v move on with the next call).
 * work: rows+cols+2 floats owned by the layer, zero-filled once at allocation; every call leaves it zero (the
 * one-block finishing kernel clears the accumulators it consumed), so no memset launches are needed. */
int cpcsv_spectral_sigma(const float* w, float* u, float* v, float* out, float* work, int rows,
                         int cols, int iterate, int snapshot, void* stream);
/* The same power iteration for MANY layers in one launch triple (all spectral-normed layers of the critics; the per-layer
 * form is 3 launches x 54 evaluations per step). jobs: DEVICE array; job k: w [rows][cols] fp32 master, u [rows], v [cols]
 * (updated in place when iterate), work = rows + cols + 2 floats that are zero on entry and left zero, out = 2 + rows + cols
 * floats: sigma, 1/sigma, u and v snapshots of THIS call (dL/dW_orig needs them). start1 / start2: DEVICE prefix sums
 * [njobs + 1] of cpcsv_sn_multi_blocks(rows, cols, 1 | 2) over the jobs; nblk1 / nblk2 their totals. Jobs of one launch
 * must be distinct layers (calls of the same layer are sequential launches: iteration k+1 starts from u of k). */
typedef struct cpcsv_sn_job {
    const float* w;
    float* u;
    float* v;
    float* work;
    float* out;
    int rows, cols;
} cpcsv_sn_job;
int cpcsv_sn_multi_blocks(int rows, int cols, int pass);
int cpcsv_spectral_sigma_multi(const cpcsv_sn_job* jobs, int njobs, const int* start1, int nblk1, const int* start2, int nblk2,
                               int iterate, void* stream);
/* The same iteration with ONE pass over every W (round 6; the two-pass form reads the fp32 masters twice per iteration: 2 GB of a
 * step's HBM traffic): a block owns 32 columns and all rows of a job in registers, forms its columns' W^T u completely and its
 * share of W (W^T u) from the same registers; the shares (part: per job  nblk_j x rows_j + nblk_j  floats at part_off[j], DEVICE
 * arrays) are added up in a fixed order by a second small launch, then the same finishing kernel. start: DEVICE prefix sums
 * [njobs + 1] of cpcsv_sn_multi_blocks(rows, cols, 3); nblk their total; max_rows = the largest rows of the jobs (<= 1024, else
 * -1002). Not in the reproducible mode (-1003: it keeps the two-pass form's summation orders). Always iterates. */
int cpcsv_spectral_sigma_multi1(const cpcsv_sn_job* jobs, int njobs, const int* start, int nblk, float* part,
                                const long long* part_off, int max_rows, void* stream);

/* ---- BatchNorm (train mode; nn.BatchNorm1d/2d at model.py:32,77,252,256,262,...) ----------- */
/* Row groups of a BatchNorm call (optional last argument of the four entry points; NULL = one group): several passes of
 * ONE layer whose rows are concatenated - real | fake batch of a critic, story | image half of a generator pass - each
 * normalised with its OWN batch statistics, exactly as the reference's separate calls are (miscc/utils.py:70-84,
 * model.py:348,426). Per-channel vectors of group g (mean, invstd, scale, shift, sums, bwd_sums) live at the passed
 * pointer + g*pstride floats; gamma / beta / running statistics / dgamma / dbeta are the layer's own (shared). */
typedef struct cpcsv_bn_groups {
    int n;             /* 1..4 groups                                                                        */
    long row[5];       /* cumulative row offsets: group g = rows [row[g], row[g+1]) of x / y / dy / dx          */
    long pstride;
    int tile[5];       /* cpcsv_bn_finalize: cumulative counts of the GEMM's statistics partials per group: group g owns
                          partial rows p*TM + [tile[g], tile[g+1]) for every phase p < nph (no M tile of cpcsv_gemm_nt
                          straddles a group, see cpcsv_gemm_desc.ngroups). `row` may be in any unit that divides the
                          sample count (e.g. low-resolution rows of the sub-pixel form): count_g = count * rows_g / row[n];
                          the running statistics are updated group after group, in order                        */
    int nph, TM;       /* phases (1, or 4 for the sub-pixel upsample+conv) and partial rows per phase               */
    const float* sigma[4]; /* cpcsv_bn_bwd_apply: {sigma, 1/sigma} of the spectral-normed conv in front, per group (NULL:
                          none). dx of that group is multiplied by 1/sigma - the conv's data- and weight-gradient GEMMs
                          then need no per-pass scale - and gw_out[g] receives the group's <G, W>              */
} cpcsv_bn_groups;
/* reduce per-block partials from cpcsv_gemm_nt into mean / biased var, build scale/shift, update
 * running stats (momentum 0.1, unbiased var), all fp32. scale/shift have Cs entries (pads = 0).
 * The affine map is pinned: scale = fl(gamma * invstd), shift = fma(-mean, scale, beta), y = act(fma(x, scale, shift)) - every
 * function below that recomputes the pre-activation from (mean, invstd, gamma, beta) uses exactly these roundings, so the
 * backward pass's activation mask is the forward's mask bit for bit (reference model.py:31-33 takes it from the stored output). */
int cpcsv_bn_finalize(const float* partials, int mtiles, int ldstat, long count, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float* mean,
                      float* invstd, float* scale, float* shift, int C, int Cs, float eps,
                      float momentum, int update_running, float* bwd_sums, const cpcsv_bn_groups* groups, void* stream);
/* bwd_sums: NULL, or the [2][Cs] fp32 accumulator of the coming backward pass: it is zeroed here for free */
/* y = act(x*scale[c] + shift[c]);  x,y [rows][Cs] dtype */
int cpcsv_bn_apply(const void* x, void* y, int dtype, const float* scale, const float* shift,
                   long rows, int C, int Cs, int act, const cpcsv_bn_groups* groups, void* stream);
/* cpcsv_bn_finalize + cpcsv_bn_apply in ONE launch when a call has only a handful of statistics partial rows
 * (groups->tile[n] * nph <= 64: dense layers over few statistics tiles - the generator's fc / fc_seg with their 32768 / 16384
 * BatchNorm1d features, the text encoders' BatchNorm1d layers, the critics' 4x4 maps): every block sums the rows of its own
 * channels in double, in a fixed order (bit-reproducible), no cpcsv_bn_finalize launch. `groups` is required and carries
 * tile / nph / TM like the one cpcsv_bn_finalize takes, with `row` in rows of x (as for cpcsv_bn_apply). */
int cpcsv_bn_apply_partials(const void* x, void* y, int dtype, const float* partials, int ldstat, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float* stat_out, float* bwd_sums, long rows, int C, int Cs,
                            int act, float eps, float momentum, const cpcsv_bn_groups* groups, void* stream);
/* backward pass 1: sums[k][0][c] += sum dz, sums[k][1][c] += sum dz*xhat with dz = dy*act'(z), z = fma(x, scale, shift) as in the
 * forward pass, recomputed from x (the activation output is not re-read); sums fp32 [CPCSV_BN_SUM_COPIES][2][Cs], zero on entry
 * (cpcsv_bn_finalize clears its bwd_sums argument, which has this shape). The row slabs of the launch spread their
 * atomics over the copies k; pass 2 adds the copies up. */
#define CPCSV_BN_SUM_COPIES 8
int cpcsv_bn_bwd_reduce(const void* dy, const void* x, int dtype, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, float* sums, long rows, int C, int Cs, int act,
                        const cpcsv_bn_groups* groups, void* stream);
/* backward pass 2: dx = gamma*invstd*(dz - sums0/rows - xhat*sums1/rows) [* 1/sigma]; dgamma/dbeta (+)= sums (of all groups,
 * added in group order). gw_out (optional, with sigma and the BatchNorm eps): receives sum(dL/dW_eff .* W_orig)
 * of the spectral-normed conv whose output x is, = sigma * sum_c gamma_c*sums1_c*eps*invstd_c^2 (the rank-1 term of
 * cpcsv_unpack_wgrad needs it; in closed form because BN removes the mean and, up to eps, the scale of x); one value per
 * group. `sigma` (single group, groups == NULL) = that pass's {sigma, 1/sigma}: dx is multiplied by 1/sigma. */
int cpcsv_bn_bwd_apply(const void* dy, const void* x, void* dx, int dtype, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, const float* sums,
                       float* dgamma, float* dbeta, long rows, int C, int Cs, int act, int accumulate,
                       float* gw_out, const float* sigma, float eps, const cpcsv_bn_groups* groups, void* stream);
/* out[c] += sum_r x[r][c], c < C  (bias gradients; out is fp32 and is accumulated into) */
int cpcsv_colsum(const void* x, int dtype, float* out, long rows, int C, int Cs, void* stream);

/* ---- elementwise / layout ------------------------------------------------------------------ */
/* dz = dy * act'(y) for activations applied in a GEMM epilogue (tanh / sigmoid / lrelu) */
int cpcsv_act_bwd(const void* dy, const void* y, void* dz, int dtype, long n, int act, void* stream);
/* out = a*b + b  (model.py:383,387) and its backward da = dout*b, db = dout*(a+1) */
int cpcsv_gate_fwd(const void* a, const void* b, void* out, int dtype, long n, void* stream);
int cpcsv_gate_bwd(const void* dout, const void* a, const void* b, void* da, void* db, int dtype, long n, void* stream);
/* channel-planar images <-> NHWC [frames][HW][Cs] (pads written as zero). Frame f = (b, t) with
 * b = f / T, t = f % T lives at element offset b*sB + t*sT, channel c at + c*sC, its HW pixels are
 * contiguous. Plain NCHW: T=1, sB=C*HW, sC=HW. A story (B,C,T,H,W) (model.py:611-613): sB=C*T*HW,
 * sT=HW, sC=T*HW. sdtype/ddtype: 0 fp32, 1 bf16 */
int cpcsv_planar_to_nhwc(const void* src, int sdtype, void* dst, int ddtype, int frames, int T, long sB,
                         long sT, long sC, int C, int HW, int Cs, void* stream);
int cpcsv_nhwc_to_planar(const void* src, int sdtype, void* dst, int ddtype, int frames, int T, long sB,
                         long sT, long sC, int C, int HW, int Cs, void* stream);
/* F4 input pipeline, device half (replaces the CPU work of main_pororo.py:71-84 per frame: transforms.ToTensor +
 * transforms.Normalize, and `video_transform`'s stack + permute): src = pre-decoded uint8 frames [frames][HW][C] (HWC, the
 * layout PIL / numpy hold them in, datasets/pororo.py:118,139), already at the training resolution. Writes
 *   planar (may be NULL): fp32 ((x/255) - mean[c]) / std[c] at  b*sB + t*sT + c*sC + pixel  (frame f = (b,t) = (f/T, f%T);
 *                         same addressing as cpcsv_planar_to_nhwc: the batch tensor the reference's loaders yield), bit-equal
 *                         to torch's fp32 result;
 *   nhwc   (may be NULL): the same values as NHWC [frames][HW][Cs] in nhwc_dtype with zero channel pads.
 * mean / std: DEVICE arrays of C floats. */
int cpcsv_ingest_u8(const void* src, float* planar, void* nhwc, int nhwc_dtype, int frames, int T, long sB, long sT,
                    long sC, int C, int HW, int Cs, const float* mean, const float* stdv, void* stream);
/* generic strided 2-D copy with cast: dst[r][dcol0 + c] = src[r][scol0 + c], c < cols.
 * sdtype/ddtype: 0 fp32, 1 bf16. Used for concat / split / padding of small matrices.
 * accumulate: 0 overwrite, 1 add to dst, 2 overwrite AND zero every other column of the dst rows [0, ldd). */
int cpcsv_copy2d(const void* src, int sdtype, long lds, int scol0, void* dst, int ddtype, long ldd,
                 int dcol0, long rows, int cols, int accumulate, void* stream);
/* torch.cat of up to 4 contiguous fp32 [rows][w_k] matrices + zero pad to ldd + cast (model.py:316,371,378) */
int cpcsv_concat_pad(const float* s0, int w0, const float* s1, int w1, const float* s2, int w2, const float* s3,
                     int w3, int nsrc, void* dst, int ddtype, long rows, int ldd, void* stream);
/* Patch matrix of a k x k / stride s / pad p convolution over NHWC frames [F][H][W][Cs] (C real channels):
 * out[(f,oy,ox)][c*k*k + ky*k + kx], row stride ld >= C*k*k, other columns zero; adjoint = 1 is its transpose-apply
 * (x = d(patch matrix), out = d(frames), overwritten). The order critic's 7x7 stem conv (VideoEncoder,
 * reference model.py:18-22: 49 taps, 3 channels) runs as this + a dense layer on the master viewed [Cout][C*k*k]. */
int cpcsv_im2col(const void* x, void* out, int dtype, int F, int H, int W, int Cs, int C, int k, int s, int p, int ld,
                 int adjoint, void* stream);
/* The batch preparation of one training step (reference trainer.py:254-264,287-288,303-304) in ONE launch, all fp32:
 *   im_motion [IM][td+L]   = cat(im_desc[:, :td], im_lab)            (:255,287)      im_content [IM][T][td] = im_cont[:, :, :td]   (:256)
 *   st_motion [ST][T][td+L] = cat(st_desc[:, :, :td], st_lab)        (:263,288)      st_text    [ST][T][td] = st_desc[:, :, :td]   (:264)
 *   st_text_mean [ST][td]  = st_text.mean(1)                         (:304)          chars [ST][L] = (st_lab.mean(1) > 0)         (:303)
 * ld_* = row strides (elements) of the description / content inputs, whose rows are wider than td (356 of 365 in the reference). */
int cpcsv_batch_prep(const float* im_desc, long ld_imd, const float* im_lab, const float* im_cont, long ld_imc, const float* st_desc,
                     long ld_std, const float* st_lab, float* im_motion, float* im_content, float* st_motion, float* st_text,
                     float* st_text_mean, float* chars, int IM, int ST, int T, int td, int L, void* stream);
/* D_GET_LOGITS input (model.py:89-92): out[n][p][0:C)=feat[n][p][:], out[n][p][Cs_f:Cs_f+E)=cond[n][:]
 * for p in 0..15; and backward: dfeat = dout[..., :C]; dcond not needed (cond is detached). */
int cpcsv_cond_concat(const void* feat, const float* cond, void* out, int dtype, int N, int P, int C,
                      int Cs_f, int E, int Cs_out, void* stream);
/* the three D_GET_LOGITS inputs of a critic update as ONE tensor (miscc/utils.py:74-84): feat = [real (N) | fake (N)] feature
 * rows; out = [(real_i, cond_i) (N) | (real_i, cond_{i+1}) (N-1, the "wrong" pairs of :78) | (fake_i, cond_i) (N)], cond tiled
 * over the P pixels of the map like cpcsv_cond_concat. _bwd: dfeat[2N] from dout[3N-1] (real rows collect two terms). */
int cpcsv_cond_triplet(const void* feat, const float* cond, void* out, int dtype, int N, int P, int C, int Cs_f, int E,
                       int Cs_out, void* stream);
int cpcsv_cond_triplet_bwd(const void* dout, void* dfeat, int dtype, int N, int P, int C, int Cs_f, int Cs_out, void* stream);
/* story critic (model.py:616-617): out[n][p][c] = mean_t in[(n*T+t)][p][c]; bwd broadcasts /T */
int cpcsv_mean_t(const void* in, void* out, int dtype, int N, int T, long inner, void* stream);
int cpcsv_mean_t_bwd(const void* dout, void* din, int dtype, int N, int T, long inner, void* stream);
int cpcsv_fill_zero(void* p, long bytes, void* stream);
/* up to 8 device-to-device copies in ONE launch (the inputs of a captured graph piece go into its static buffers) */
typedef struct cpcsv_copy_list {
    void* dst[8];
    const void* src[8];
    long bytes[8];
    int n;
} cpcsv_copy_list;
int cpcsv_copy_many(const cpcsv_copy_list* l, void* stream);

/* Operand copies of up to CPCSV_PACK_JOBS small fp32 dense weights w [cout][cin] (reference model.py: CA_NET.fc :50, m_net / c_net /
 * image_net / filter_net :165-193, the two GRU cells :206-214 - the layers the multi-tensor Adam launch has just rewritten) in ONE launch:
 * fwd [cout][cin_s] (zero channel pads) and / or lin [cin_s][cout_s] (the transpose, zero pads) - bit for bit what cpcsv_pack_weight
 * writes for taps = S = 1 in fp32, nine + nine launches per step before. A NULL fwd / lin skips that copy. blk0 is filled in by the
 * library (first block of the job). */
#define CPCSV_PACK_JOBS 16
typedef struct cpcsv_pack_job {
    const float* w;
    float* fwd;
    float* lin;
    int cout, cin, cin_s, cout_s;
    int blk0, _pad;
} cpcsv_pack_job;
typedef struct cpcsv_pack_list {
    int n, _pad;
    cpcsv_pack_job j[CPCSV_PACK_JOBS];
} cpcsv_pack_list;
int cpcsv_pack_dense_many(cpcsv_pack_list* l, void* stream);

/* ---- D_GET_LOGITS' 3x3 conv in factored form (csrc/condhead.hip) ------------------------------------------------------------
 * reference model.py:75-80,89-92: the head tiles the condition vector over the 4x4 map (`c_code.repeat(1, 1, 4, 4)`), concatenates it
 * to the 8 ndf feature channels and runs SN-conv3x3 (8 ndf + nef -> 8 ndf) + BatchNorm + LeakyReLU. The nef condition channels are
 * spatially CONSTANT, so their share of the convolution at pixel p is  sum over the taps that are inside the map at p  of
 * pt[sample][tap][o] = <cond[sample], W[o][tap][8 ndf:]>  - one small dense product per sample and tap (cpcsv_gemm_nt with
 * bcol_rows) instead of 9 x nef of the K extent of every pixel row. And the three calls of a critic update (miscc/utils.py:70-84:
 * real / wrong / fake) pair only 2 N distinct feature maps and N distinct condition rows: "wrong" = real features [0, N-1) with the
 * conditions [1, N). The feature part of the conv is therefore computed ONCE per distinct feature map (cpcsv_gemm_nt over [real |
 * fake], wstride = the layer's full K slice, slabs_only), and this entry point assembles the three calls from it:
 *   z[g][n][p][o] = (sum_k ws[k][(feat0[g] + n) * P + p][o]  +  sum_{tap live at p} pt[cond0[g] + n][tap][o]) * (1 / sigma_g)
 * followed by train-mode BatchNorm over the rows of group g (its own batch statistics, running statistics updated group after group)
 * and the activation. One launch: a block owns 4 output channels for ALL rows, so the statistics never leave the block.
 * K goes 9 x (8 ndf + nef) -> 9 x 8 ndf and the rows 3 N - 1 -> 2 N: 45 % of the MACs of the literal form.
 * Outputs: z (the pre-BatchNorm conv output, what cpcsv_bn_bwd_* read) and y, both [R][Cs] dtype with R = sum_g count[g] * P rows in
 * group order; stat_out as cpcsv_bn_apply_partials writes it (per group g at + g * pstride: mean, invstd, scale, shift rows of Cs
 * floats, then the zeroed backward accumulators [2 * CPCSV_BN_SUM_COPIES][Cs] if bwd_sums != 0). */
typedef struct cpcsv_cond_head {
    const float* ws;       /* [nslabs][ws_rows][ldws] fp32 K-slice slabs of the feature conv (nslabs = 1: a plain fp32 product)     */
    int nslabs, ldws;
    long ws_rows;
    const float* pt;       /* [cond rows][ntaps][ldp] fp32 per-tap condition products                                           */
    int ldp;
    int MH, MW;            /* the map (4 x 4); P = MH * MW pixels per sample, 3x3 taps with padding 1                            */
    int ngroups;           /* 1..4 reference calls                                                                               */
    int count[4];          /* samples of call g                                                                                 */
    int feat0[4];          /* its first feature sample (row feat0 * P of a slab)                                                */
    int cond0[4];          /* its first condition row of pt                                                                     */
    const float* galpha[4];/* its {1 / sigma} (device scalar) or NULL                                                            */
    void* z;
    void* y;
    int dtype;
    int C, Cs;             /* output channels, stored channels (multiple of 8; pads are written as zeros)                        */
    const float* gamma;
    const float* beta;
    float* running_mean;   /* NULL: not updated                                                                                  */
    float* running_var;
    float* stat_out;
    long pstride;
    int bwd_sums;          /* 1: zero the backward accumulators behind the four statistics rows                                  */
    int act;
    float eps, momentum;
} cpcsv_cond_head;
int cpcsv_cond_head_fwd(const cpcsv_cond_head* d, void* stream);
/* samples (sum of count[]) one cpcsv_cond_head_fwd launch can take; callers fall back to the literal form beyond it */
int cpcsv_cond_head_max_samples(void);
/* ... and the backward glue: from dz [R][Cs] dtype (dL/dz of the three calls, already divided by their sigma: cpcsv_bn_bwd_apply)
 *   dF[s][p][:]   = sum over the calls g that use feature sample s  of dz[g][s - feat0[g]][p][:]        [feature samples][P][Cs] dtype
 *   dZt[c][tap][:] = sum over the calls g that use condition row c, over the pixels p where tap is live, of dz[g][c - cond0[g]][p][:]
 *                                                                                                     [cond rows][ntaps][Cs] dtype
 * dF is dY of the feature conv's data- and weight-gradient GEMMs, dZt (may be NULL) that of the condition columns' weight gradient
 * (cpcsv_wgrad_tn with dy_tapstride = Cs over the condition rows). Fixed summation order. */
typedef struct cpcsv_cond_head_grad {
    const void* dz;
    void* dF;
    void* dZt;
    int dtype;
    int MH, MW, Cs;
    int nfeat, ncond;      /* feature samples, condition rows                                                                    */
    int ngroups;
    int count[4], feat0[4], cond0[4];
} cpcsv_cond_head_grad;
int cpcsv_cond_head_bwd(const cpcsv_cond_head_grad* d, void* stream);

/* ---- the critics' logit layer (csrc/head.hip) ---------------------------------------------------------------------
 * D_GET_LOGITS' last layer, Conv2d(8*ndf, 1, 4, 4) + Sigmoid over the 4x4 map (reference model.py:79-80): one output per sample,
 * spectral-normed, biased. Three small launches instead of ~25 through the general layer path.
 *   x [R][K] dtype: the flattened NHWC map (K = taps * Cin_s, pads zero);  w [K] dtype: cpcsv_pack_weight forward layout of the
 *   single output channel (k = tap * Cin_s + c);  p / dy / dz [R] fp32.
 * groups: rows [row[g], row[g+1]) are reference call g (real / wrong / fake of miscc/utils.py:74-84, or one call), each with its own
 * spectral-norm state: sigma[g] = {sigma, 1/sigma} (NULL: not normed), u[g] (1 value), v[g] (Cin * taps values in the master's
 * (c, tap) order; u / v NULL: no rank-1 term, e.g. frozen weights).
 *   fwd:   p[r] = sigmoid(<x[r], w> / sigma_g + bias[0])
 *   bwd:   dz[r] = dy[r] p (1 - p);  dx[r][:] = dz[r] / sigma_g * w   (dx NULL: skipped)
 *   wgrad: dW[c*taps + t] += sum_g ( G_g[k] / sigma_g - <G_g, w> / sigma_g^2 * u_g v_g[c*taps + t] ),  G_g[k] = sum_{r in g} dz[r] x[r][k];
 *          db[0] += sum_r dz[r].  dW is the MASTER-layout gradient [Cin][taps] (added to); scratch: cpcsv_logit_head_scratch(K, n) floats.
 *          Fixed summation order (deterministic). */
typedef struct cpcsv_logit_groups {
    int n;
    int row[5];
    const float* sigma[4];
    const float* u[4];
    const float* v[4];
} cpcsv_logit_groups;
int cpcsv_logit_head_fwd(const void* x, const void* w, const float* bias, float* p, int dtype, int R, int K,
                         const cpcsv_logit_groups* g, void* stream);
int cpcsv_logit_head_bwd(const float* dy, const float* p, const void* w, void* dx, float* dz, int dtype, int R, int K,
                         const cpcsv_logit_groups* g, void* stream);
long cpcsv_logit_head_scratch(int K, int ngroups);
int cpcsv_logit_head_wgrad(const float* dz, const void* x, const void* w, float* scratch, float* dW, float* db, int dtype, int R,
                           int K, int Cin, int Cin_s, int taps, const cpcsv_logit_groups* g, void* stream);

/* ---- recurrent text encoders / dynamic filter ---------------------------------------------- */
/* Dense layer over at most 64 rows in exact fp32 (the text / motion encoders and GRU recurrences, model.py:223-224,252-262,
 * 313-346: nn.Linear / nn.GRUCell products over 12-60 rows): y[m][n] = act(alpha * sum_k x[m][k] w[n][k] + bias[n]) for n < N,
 * zeros for N <= n < ldy; w row-major [N][ldw]. ONE launch (cpcsv_gemm_nt needs a split-K pass for these shapes). stats: NULL or
 * the BatchNorm partials [ceil(M/16)][2][ldstat] of the pre-activation values (cpcsv_bn_finalize with 16 rows per partial).
 * K, ldx, ldw, ldy multiples of 4. */
int cpcsv_dense_rows(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int M, int N, int K,
                     const float* alpha, const float* bias, int act, float* stats, int ldstat, const float* init, int ldi,
                     int accumulate, void* stream);
/* init: NULL, or [M][ldi >= ldy] values ADDED to the result (the direct gradient path of a recurrence next to the product);
 * accumulate != 0: the result is added to what y holds (the gradient a recurrence state already has from its other consumer). */
/* One nn.GRUCell step (model.py:223-224, 331, 342) from precomputed input gates gi = W_ih x + b_ih [B][ldg]:
 * hnew = GRU(gi, h; W_hh, b_hh), state rows [B][ldh] (ldh >= H, pads zero), w_hh row-major [3H][ldw]; gates [B][4H] = r, z, n,
 * W_hn h + b_hn for cpcsv_gru_gates_bwd. One launch instead of the W_hh product + cpcsv_gru_gates_fwd. */
int cpcsv_gru_step_fwd(const float* gi, int ldg, const float* h, int ldh, const float* w_hh, int ldw, const float* b_hh,
                       float* hnew, float* gates, int B, int H, void* stream);
/* ... and its weight gradient, added straight into the master-layout gradient: dW[n][k] += sum_m dz[m][n] x[m][k], dW [N][Kr]
 * row-major (Kr = the real input width), M <= 64; db (NULL or [N]): db[n] += sum_m dz[m][n], the bias gradient, from the same launch.
 * Calls that add to one dW / db must be ordered by their stream. */
int cpcsv_dense_rows_wgrad(const float* dz, int ldz, const float* x, int ldx, float* dW, float* db, int M, int N, int Kr, void* stream);
/* ... and for MANY small layers in ONE launch: `t` = the weights (dW [N][Kr] master layout, db [N] or NULL, both added to), each with
 * `npieces` consecutive entries of `p` (one per backward call that contributes: dz [M][ldz], x [M][ldx], M <= 64), summed in list order -
 * no atomics, deterministic. block0 / bx are filled in by the call. The generator's backward parks its ~18 small weight gradients
 * (they feed only the optimiser) and issues this once at its end. */
#define CPCSV_SMALL_WG_TARGETS 16
#define CPCSV_SMALL_WG_PIECES 48
typedef struct cpcsv_wgrad_piece {
    const float* dz;
    const float* x;
    int ldz, ldx, M, _pad;
} cpcsv_wgrad_piece;
typedef struct cpcsv_wgrad_target {
    float* dW;
    float* db;
    int N, Kr, piece0, npieces, block0, bx;
} cpcsv_wgrad_target;
typedef struct cpcsv_small_wgrad_list {
    int ntargets, npieces;
    cpcsv_wgrad_target t[CPCSV_SMALL_WG_TARGETS];
    cpcsv_wgrad_piece p[CPCSV_SMALL_WG_PIECES];
} cpcsv_small_wgrad_list;
int cpcsv_dense_rows_wgrad_multi(cpcsv_small_wgrad_list* l, void* stream);
/* GRUCell pointwise part (nn.GRUCell, model.py:223-224): gi,gh [B][ldg] fp32 (3H gate pre-activations with biases),
 * h [B][ldh] -> hnew [B][ldh]; saves r,z,n,(hn = W_hn h + b_hn) in gates [B][4H] for backward. ldh >= H is the padded
 * width the next step's W_hh GEMM reads (pad columns of hnew are written as zeros). */
int cpcsv_gru_gates_fwd(const float* gi, const float* gh, const float* h, float* hnew, float* gates,
                        int B, int H, int ldg, int ldh, void* stream);
/* dgi, dgh [B][ldg], dh_prev [B][ldh] from dhnew [B][ldh] */
int cpcsv_gru_gates_bwd(const float* dhnew, const float* gates, const float* h, float* dgi, float* dgh,
                        float* dh, int B, int H, int ldg, int ldh, void* stream);
/* DynamicFilterLayer1D (layers.py:69-80): sig [N][C][L], taps [N][C][K] -> out [N][L] */
int cpcsv_dfl1d_fwd(const float* sig, const float* taps, float* out, int N, int C, int L, int K, int pad, void* stream);
int cpcsv_dfl1d_bwd(const float* dout, const float* sig, const float* taps, float* dsig, float* dtaps,
                    int N, int C, int L, int K, int pad, void* stream);
/* CA_NET.reparametrize (model.py:53-60): c = eps*exp(0.5*logvar)+mu; bwd */
int cpcsv_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* out, long n, void* stream);
int cpcsv_reparam_bwd(const float* dout, const float* logvar, const float* eps, float* dmu, float* dlogvar,
                      long n, int accumulate, void* stream);

/* ---- the generator's text / motion encoders as a handful of STAGE launches (csrc/text.hip) -----------------------------------
 * CA_NET, m_net, c_net, the two GRU recurrences, image_net, filter_net and DynamicFilterLayer1D (reference model.py:37-65,
 * 302-346,371-378; layers.py:69-80) are ~2 M weights and <= 64 rows per call: pure launch latency (40-odd dependent launches of
 * 8-20 us at the head of every generator pass, ~90 at the tail of its backward). A STAGE is one launch that runs up to
 * CPCSV_TXT_MAX_JOBS independent jobs side by side; a pass is ~10 stages forward, ~11 backward. All fp32 (exact FMA chains),
 * no atomics: one fixed summation order.
 *
 * Product jobs share one tile engine: a block owns `nc` output columns  col_c = j*js + c*cs  (c < nc <= 4; tile j = its block
 * index within the job) for ALL rows of the call (M <= 256), val[m][c] = sum_k x[m][k] * w[col_c][k], K a multiple of 4
 * (K = 0: no product), x rows and w rows 16-byte aligned (ldx, ldw multiples of 4). Owning every row of its columns is what lets
 * a block finish BatchNorm1d (batch statistics + running-statistics update), the GRU gate math or the BatchNorm backward in its
 * epilogue. A job has `npass` (1 or 2) calls of the SAME layer (the story call and the image call of a generator pass), run one
 * after the other inside each block so that running statistics / accumulated parameter gradients see them in call order. */
#define CPCSV_TXT_MAX_JOBS 8
#define CPCSV_TXT_MAX_ROWS 256
/* job types; P[p][i] = per-pass pointers, Q[i] = shared pointers, A[i] = ints, T[p] = time steps of pass p (unused: 0 / NULL).
 * Product jobs tile the output columns in fours (tile j = columns 4j .. 4j+3) unless stated otherwise. */
#define CPCSV_TXT_DENSE 1    /* y_p = act(BN(x_p w^T + bias)).  Q0 gamma (NULL: no BatchNorm) Q1 beta Q2 running_mean Q3 running_var
                                (Q2 NULL: not updated); P0 lin_p [M][ldy]: the pre-BatchNorm values (saved for backward; NULL: not
                                stored), P1 save_p [2][ldy]: batch mean, invstd (NULL: not stored). Columns N <= col < ldy of y (and
                                lin) are written as zeros. nn.Linear + nn.BatchNorm1d + activation, reference model.py:250-257,302-308.
                                Products and batch statistics are summed in the order of cpcsv_dense_rows + cpcsv_bn_apply_partials
                                (16-row partials of sum / sum of squares, combined in double): bit-identical to that pair */
#define CPCSV_TXT_CA 2       /* CA_NET (model.py:44-65): tile j = columns (j, C+j), C = A0. y_p [M][2C] = relu(x w^T + b) (the ReLU
                                precedes the split); P0 mu_p [M][C], P1 logvar_p [M][C] (contiguous copies), P2 eps_p [M][C] or NULL,
                                P3 code_p [M][A1]: eps*exp(logvar/2)+mu (eps NULL: mu - sample_images feeds the MEAN, model.py:433),
                                columns C <= col < A1 zero */
#define CPCSV_TXT_GRU_FWD 3  /* one nn.GRUCell step (model.py:223-224,331,342): x_p = h_prev [M][ldx], w = W_hh [3H][ldw], K = ldx,
                                tile j = rows (j, H+j, 2H+j) of W_hh, H = A0; bias = b_hh; P0 gi_p [M][A1] (= W_ih x + b_ih),
                                P1 gates_p [M][4H] (r, z, n, W_hn h + b_hn); y_p = h_new [M][ldy] (pads zero); P2 (or NULL): a second,
                                STORY-MAJOR copy of the new state: row (m*T[p] + A2) of [M*T[p]][ldy] (A2 = this step: the rows filter_net
                                reads, model.py:343-346) */
#define CPCSV_TXT_PREP 4     /* padded operand matrices of a call: P0 motion [B][T][md] (or [B][md], T = 1), P1 step noise [T][B][nz],
                                P2 initial-state noise [B][md] -> P3 mpad [T*B][A2] time-major rows (t*B + b), P4 e [T*B][A3] =
                                [noise_t | motion_t | 0], P5 n0pad [B][A2], y_p tpad [B*T][A2]: the motion rows story-major (what
                                image_net reads, model.py:365-373). M[p] = B, A0 = md, A1 = nz (model.py:313-324) */
#define CPCSV_TXT_JOINT 5    /* zmc rows of a call (model.py:371-378): row r = b*T + t (story-major) of y_p [B*T][ldy] (dtype A6:
                                0 fp32 / 1 bf16) = [h_m[t+1][b][0:md) | mu[r % B][0:C) | DFL1D(m_image[r], c_filter[r]) | 0]
                                (the story call's c_mu = r_mu.repeat(T, 1) is TILED, model.py:361; layers.py:69-80).
                                P0 hall_m [T+1][B][A5], P1 mu [B][C], P2 m_image [B*T][ldx] (nch x L), P3 c_filter [B*T][ldw]
                                (nch x KF), both story-major. M[p] = B, A0 = md, A1 = C, A2 = L, A3 = KF, A4 = nch */
#define CPCSV_TXT_DFL_BWD 6  /* backward of JOINT for one call: x_p = dzmc rows of the call [B*T][ldx] (dtype A6): P0 m_image (tanh
                                output) [B*T][ldw], P1 c_filter [B*T][ldy], P2 d_pre_image [B*T][ldw] = dsig * (1 - y^2) (pads zero),
                                P3 d_filter [B*T][ldy] (pads zero; these four story-major), P4 dh_ext_m [T][B][A5] = dzmc[.., 0:md)
                                time-major (pads zero),
                                P5 dmu_tot [B][C] = y_p (dmu_ext [B][C] or NULL) + the rows of dzmc[.., md:md+C) that read mu[b].
                                ints as JOINT */
#define CPCSV_TXT_BN_BWD 7   /* BatchNorm1d backward on a column tile; dy_p = x_p [M][ldx] (K = 0) or the product
                                x_p w^T + P2 init_p [M][ldy] (K > 0: dh_0 = dgh_0 W_hh + dh_0 * z of a recurrence).
                                P0 lin_p, P1 save_p (mean, invstd); Q0 gamma, Q1 dgamma (+=), Q2 dbeta (+=); y_p = dlin_p [M][ldy]
                                (pads zero) */
#define CPCSV_TXT_GRU_BWD 8  /* one step of a GRUCell's backward: dh[m][j] = P0 dh_ext_p[m * A2][j] (rows ldy apart, every A2-th row:
                                A2 = 1, or T for a story-major matrix whose row of this step P0 points at; NULL: 0) + (K > 0:
                                x_p (= dgh of step t+1 [M][ldx]) w^T (w = W_hh^T [ldy][ldw]) + P1 dhz_next_p [M][ldy]);
                                P2 gates_p [M][4H] of this step, P3 h_prev_p [M][ldy]; writes P4 dgi_p, P5 dgh_p [M][A1] (columns
                                3H..A1 zero) and y_p = dhz_p [M][ldy] = dh * z (what step t-1 adds). H = A0 */
#define CPCSV_TXT_CA_BWD 9   /* d code -> d (pre-split CA_NET output): x_p = dlin of c_net [M][ldx], w = W_cnet^T, K; tile over
                                j < C = A0: d = product; dmu = d + P0 dmu_tot_p[m][j]; dlv = P1 dlv_ext_p[m][j] (NULL: 0) +
                                (P2 eps_p ? d * eps * 0.5 * exp(logvar / 2) : 0); P3 xca_p [M][2C] (the ReLU outputs: mask and
                                logvar); y_p = d_xca [M][2C] */
typedef struct cpcsv_txt_job {
    int type, npass;
    int M[2];
    int T[2];
    int N, K, ldx, ldw, ldy;
    int act;
    int A[8];
    float eps, momentum;
    int blk0, nblk;          /* filled in by cpcsv_text_stage: first block / number of blocks of this job */
    const float* x[2];
    const float* w;
    const float* bias;
    void* y[2];
    void* P[2][6];
    void* Q[4];
} cpcsv_txt_job;
typedef struct cpcsv_txt_stage {
    int njobs, _pad;
    cpcsv_txt_job job[CPCSV_TXT_MAX_JOBS];
} cpcsv_txt_stage;
/* one launch: every job of the stage, blocks = sum of the jobs' tiles */
int cpcsv_text_stage(cpcsv_txt_stage* st, void* stream);

/* ---- losses (miscc/utils.py:51-52,184-188; nn.MSELoss trainer.py:222) ----------------------- */
/* each writes loss[0] (fp32 mean) and grad = d loss / d input (already divided by the mean size) */
int cpcsv_bce_fwd(const float* p, const float* target, float* loss, float* grad, long n, void* stream);
/* the three nn.BCELoss terms of a critic update in one launch (miscc/utils.py:76-101): p / target = [real | wrong | fake]
 * (n0, n1, n2 entries); out[g] = mean BCE of group g, out[3] = sum_g w_g out[g], grad = d out[3] / d p. */
int cpcsv_bce_groups(const float* p, const float* target, float* out, float* grad, int n0, int n1, int n2, float w0, float w1,
                     float w2, void* stream);
/* acc: NULL, or receives get_multi_acc of the same logits (miscc/utils.py:108,153,313-321: labels that are 1 AND predicted
 * sigmoid(x) >= .5, over the number of labels that are 1) - the accuracy the loop logs beside this loss, from the same launch */
int cpcsv_mlsm_fwd(const float* logits, const float* target, float* loss, float* grad, float* acc, int N, int C, int ld, void* stream);
/* Weighted sum of up to 8 device scalars and its backward (errG_total = im_errG + KL terms + ratio * (...), reference
 * trainer.py:409-413: a dozen scalar launches forward and as many backward otherwise): out[0] = sum_i w[i] * x[i][0];
 * dx[i] = g[0] * w[i]. */
typedef struct cpcsv_scalar_list {
    const float* x[8];
    float w[8];
    int n;
} cpcsv_scalar_list;
int cpcsv_lincomb_fwd(const cpcsv_scalar_list* l, float* out, void* stream);
int cpcsv_lincomb_bwd(const float* g, const cpcsv_scalar_list* l, float* dx, void* stream);
int cpcsv_kl_fwd(const float* mu, const float* logvar, float* loss, float* dmu, float* dlogvar, long n, void* stream);
/* count = number of logical elements the mean divides by (n may include zero channel pads) */
int cpcsv_mse_fwd(const void* a, const void* b, int dtype, float* loss, void* da, void* db, long n, long count, void* stream);
/* y = alpha * x, with alpha read from a device scalar (chain rule through a scalar loss) */
int cpcsv_scale_by(const void* x, void* y, int dtype, const float* alpha, float mult, long n, int accumulate, void* stream);

/* ---- optimiser (torch.optim.Adam, trainer.py:212-220) --------------------------------------- */
/* multi-tensor Adam: table[i] = {p, g, m, v} device pointers (fp32), sizes[i] element counts, one block per
 * (tensor, 4096-element chunk). hyper is a DEVICE array {step, lr}: the call first advances step by one, then
 * applies the update with bias corrections computed from it on the device - nothing step-dependent is baked into
 * the launch, so the call can live inside a captured HIP graph. One launch for the whole optimiser. */
int cpcsv_adam_step(void* const* table, const long* sizes, int ntensors, long total_chunks,
                    const int* chunk_tensor, const long* chunk_offset, float* hyper, float beta1, float beta2,
                    float eps, void* stream);
int cpcsv_adam_chunk(void);  /* elements handled per Adam block (chunk table granularity) */
/* ---- fused per-layer optimiser step -------------------------------------------------------------------------------
 * One launch per weight tensor and step replaces {cpcsv_unpack_wgrad per backward call, cpcsv_adam_step, cpcsv_pack_weight}:
 * every master element (o, i, t) is touched once.
 *   g(o,i,t) = sum of the accumulator slices feeding tap t (tapmap / masks as in cpcsv_pack_weight[_sum])
 *              - sum_k (gw_k / sigma_k^2) * u_k[o] * v_k[i*taps + t]        (spectral-norm terms of the step's calls, k < nterms;
 *                the 1/sigma_k of each call is already in the accumulator: cpcsv_wgrad_desc.alpha)
 *   Adam(p, m, v; g) with hyper = {step count (already advanced by cpcsv_adam_step of the same optimiser), lr}  (torch.optim.Adam,
 *   reference trainer.py:212-220)
 *   fwd / bwd / lin operand copies rewritten from the NEW p in `dtype` (layouts of cpcsv_pack_weight; pad rows/columns are
 *   not touched: they were zeroed by the first cpcsv_pack_weight and never change). NULL = not needed.
 * The accumulator G is left as is (the caller zeroes its gradient bucket at the start of a step). Deterministic. */
typedef struct cpcsv_update_desc {
    const float* G;    /* accumulator [Cout][S*Cin_s] fp32 (the layout cpcsv_wgrad_tn fills)  */
    float* p;          /* master weight [Cout][Cin][taps] fp32                               */
    float* m;          /* Adam first / second moments, same layout                           */
    float* v;
    void* fwd;         /* [Cout][S*Cin_s]   */
    void* bwd;         /* [Cin][S*Cout_s]   (conv)  */
    void* lin;         /* [S*Cin_s][Cout_s] (dense) */
    const float* hyper;
    float beta1, beta2, eps;
    int dtype, Cout, Cin, taps, S, Cin_s, Cout_s;
    int sum;           /* 0: slice sl = tap tapmap[sl]; 1: slice sl = sum of the taps in masks[sl] */
    int8_t tapmap[CPCSV_MAX_TAPS];
    uint16_t masks[CPCSV_MAX_TAPS];
    int nterms;
    const float* gw[4];
    const float* sigma[4];
    const float* u[4];
    const float* v_sn[4];
    float gscale;      /* 0 or 1: G as is; otherwise every accumulator value is multiplied by gscale first (1/world when the
                          data-parallel exchange SUMS the ranks' accumulators: the mean costs no extra pass over G). The
                          spectral-norm rank-1 terms are NOT scaled: in data-parallel runs their gw scalars are mean-reduced with
                          the net's flat gradient buffer before this launch (sigma, u, v are the same on every rank) */
    float step_add;    /* added to hyper[0] before the bias corrections: 1 for launches that go out from INSIDE the backward pass,
                          i.e. before cpcsv_adam_step of the same optimiser step has advanced the counter; 0 behind it */
    int g_bf16;        /* 1: G points at bf16 values in the accumulator's layout - the wire buffer of a data-parallel exchange with the
                          bf16 payload (the fp32 accumulator is cast once before the all-reduce, nothing is cast back) */
} cpcsv_update_desc;
int cpcsv_layer_update(const cpcsv_update_desc* d, void* stream);

/* ---- streaming convolutions with a degenerate GEMM dimension (bf16 only; csrc/thin.hip) ---------------------------
 * The generator's output convs and the critics' first conv are HBM-bound (AI 9-44 FLOP/B): the wide
 * tensor is staged ONCE per tile in LDS (or read straight as 16-byte pixels), all taps are served from there.
 * cpcsv_thin_supported(kind, Cs, Cout, H, W): kind 0 = 3x3 s1 p1 with Cout <= 4 (StoryGAN.img / img_seg,
 * reference model.py:272-274,298-300), kind 1 = 4x4 s2 p1 from an 8-stored-channel image to <= 128 stored channels
 * (encode_img.0, model.py:499,541,583). Unsupported shapes go through cpcsv_gemm_nt / cpcsv_wgrad_tn.
 *   x      NHWC [N][H][W][Cs] bf16          w_fwd  cpcsv_pack_weight forward layout [Cout][taps*Cs]
 *   y, dz  NHWC [..][8] bf16 (3x3) / [N][H/2][W/2][128] (4x4)      w_bwd  cpcsv_pack_weight backward layout [Cin][9*8]
 *   G      fp32 accumulator [Cout][9*Cs] (same layout cpcsv_wgrad_tn fills); `slabs` = caller workspace of
 *          cpcsv_thin3x3_wgrad_slabs(...) * Cout*9*Cs floats (per-block partials, summed in a fixed order: deterministic) */
int cpcsv_thin_supported(int kind, int Cs, int Cout, int H, int W);
int cpcsv_thin3x3_fwd(const void* x, const void* w_fwd, void* y, int N, int H, int W, int Cs, int Cout, int act, void* stream);
int cpcsv_thin3x3_dgrad(const void* dz, const void* w_bwd, void* dx, int N, int H, int W, int Cs, int Cout, void* stream);
int cpcsv_thin3x3_wgrad_slabs(int N, int H, int W, int Cs);
int cpcsv_thin3x3_wgrad(const void* dz, const void* x, float* G, float* slabs, int N, int H, int W, int Cs, int Cout,
                        void* stream);
int cpcsv_thin4x4s2_fwd(const void* x, const void* w_fwd, void* y, const float* alpha, int N, int H, int W, int Cout,
                        int act, void* stream);
/* weight gradient of the same layer (64-wide images): G [Cout][16*8] += dz^T x with all 16 taps x 8 stored channels as
 * ONE matrix dimension, dz read once. `slabs` = cpcsv_thin4x4s2_wgrad_slabs(N, H, W) * Cout*128 floats of caller workspace
 * (0 slabs: shape not served, use cpcsv_wgrad_tn); partials are summed in a fixed order. */
/* data gradient of the same layer (W == 64): dx NHWC [N][H][W][8] = alpha * transposed-conv(dz, w_bwd); w_bwd is the
 * cpcsv_pack_weight backward layout [8][16*128]; alpha (device scalar, 1/sigma of a spectral-normed layer) may be NULL. */
int cpcsv_thin4x4s2_dgrad(const void* dz, const void* w_bwd, void* dx, const float* alpha, int N, int H, int W, void* stream);
int cpcsv_thin4x4s2_wgrad_slabs(int N, int H, int W);
int cpcsv_thin4x4s2_wgrad(const void* dz, const void* x, float* G, float* slabs, int N, int H, int W, int Cout, void* stream);

/* Reproducible mode (tests, debugging): 1 = every cross-block floating-point reduction runs in ONE fixed order (weight
 * gradients without pixel splits, BatchNorm/spectral-norm/bias sums without contended atomics), so two runs of the same
 * step - eager or replayed from a HIP graph - give bit-identical results. Process-wide; returns the previous setting.
 * The reference has no counterpart (cuDNN/ATen reductions are whatever the library picks). */
int cpcsv_set_deterministic(int on);
/* Layout self-description: fills out[0] = sizeof(struct), out[1] = number of fields, then (offset, size) per field in declaration
 * order, for the struct selected by `which`; returns the number of ints written (out == NULL: the number needed), negative on
 * error. A binding compares this with its own mirror of the struct before the first launch: a field present on one side only
 * would otherwise shift everything behind it silently. */
#define CPCSV_ABI_TAP 0
#define CPCSV_ABI_GEMM_DESC 1
#define CPCSV_ABI_WGRAD_DESC 2
#define CPCSV_ABI_SN_JOB 3
#define CPCSV_ABI_BN_GROUPS 4
#define CPCSV_ABI_UPDATE_DESC 5
#define CPCSV_ABI_SCALAR_LIST 6
#define CPCSV_ABI_COPY_LIST 7
#define CPCSV_ABI_LOGIT_GROUPS 8
#define CPCSV_ABI_WGRAD_PIECE 9
#define CPCSV_ABI_WGRAD_TARGET 10
#define CPCSV_ABI_SMALL_WGRAD_LIST 11
#define CPCSV_ABI_PACK_JOB 12
#define CPCSV_ABI_PACK_LIST 13
#define CPCSV_ABI_TXT_JOB 14
#define CPCSV_ABI_TXT_STAGE 15
#define CPCSV_ABI_COND_HEAD 16
#define CPCSV_ABI_COND_HEAD_GRAD 17
int cpcsv_abi_layout(int which, int* out, int cap);
int cpcsv_version(void);
const char* cpcsv_arch(void);

#ifdef __cplusplus
}
#endif
#endif
