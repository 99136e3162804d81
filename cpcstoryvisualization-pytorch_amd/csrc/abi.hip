// abi.hip — self-description of the C ABI's structs (host code only, no kernels).
//
// The descriptors of include/cpcsv_hip.h travel by value from a foreign-language binding (ctypes in cpcsv/_lib.py, or the
// stub INTEGRATION.md shows) into this library. A field added on one side only would silently shift everything behind it, so
// the library reports the layout IT was compiled with and the binding checks its own against that before the first launch
// (tests/test_host_cpu.py::test_abi_layout_matches_ctypes; cpcsv/_lib.py verify_layout()).
#include <cstddef>
#include "../../include/cpcsv_hip.h"

namespace {
struct Field { int off, size; };
#define F(S, f) Field{(int)offsetof(S, f), (int)sizeof(((S*)0)->f)}

const Field k_tap[] = {F(cpcsv_tap, oy), F(cpcsv_tap, ox), F(cpcsv_tap, wtap), F(cpcsv_tap, _pad)};
const Field k_gemm[] = {
    F(cpcsv_gemm_desc, A), F(cpcsv_gemm_desc, B), F(cpcsv_gemm_desc, C), F(cpcsv_gemm_desc, dtype), F(cpcsv_gemm_desc, M),
    F(cpcsv_gemm_desc, N), F(cpcsv_gemm_desc, Cs), F(cpcsv_gemm_desc, ldb), F(cpcsv_gemm_desc, ldc), F(cpcsv_gemm_desc, ntaps),
    F(cpcsv_gemm_desc, taps), F(cpcsv_gemm_desc, MH), F(cpcsv_gemm_desc, MW), F(cpcsv_gemm_desc, IH), F(cpcsv_gemm_desc, IW),
    F(cpcsv_gemm_desc, sy), F(cpcsv_gemm_desc, sx), F(cpcsv_gemm_desc, up_shift), F(cpcsv_gemm_desc, pool_rows),
    F(cpcsv_gemm_desc, scatter), F(cpcsv_gemm_desc, OH), F(cpcsv_gemm_desc, OW), F(cpcsv_gemm_desc, osy), F(cpcsv_gemm_desc, osx),
    F(cpcsv_gemm_desc, ooy), F(cpcsv_gemm_desc, oox), F(cpcsv_gemm_desc, alpha), F(cpcsv_gemm_desc, bias), F(cpcsv_gemm_desc, act),
    F(cpcsv_gemm_desc, stats), F(cpcsv_gemm_desc, ldstat), F(cpcsv_gemm_desc, out_f32), F(cpcsv_gemm_desc, splitk),
    F(cpcsv_gemm_desc, ws), F(cpcsv_gemm_desc, ldws), F(cpcsv_gemm_desc, ws_rows), F(cpcsv_gemm_desc, nphases),
    F(cpcsv_gemm_desc, ph_tap0), F(cpcsv_gemm_desc, ph_ntaps), F(cpcsv_gemm_desc, ph_ooy), F(cpcsv_gemm_desc, ph_oox),
    F(cpcsv_gemm_desc, order_m_fast), F(cpcsv_gemm_desc, ngroups), F(cpcsv_gemm_desc, grow), F(cpcsv_gemm_desc, galpha),
    F(cpcsv_gemm_desc, addend), F(cpcsv_gemm_desc, ldadd), F(cpcsv_gemm_desc, korder), F(cpcsv_gemm_desc, wstride), F(cpcsv_gemm_desc, patch),
    F(cpcsv_gemm_desc, slabs_only), F(cpcsv_gemm_desc, bcol_rows), F(cpcsv_gemm_desc, bcol_koff)};
const Field k_wgrad[] = {
    F(cpcsv_wgrad_desc, dY), F(cpcsv_wgrad_desc, X), F(cpcsv_wgrad_desc, dW), F(cpcsv_wgrad_desc, dtype), F(cpcsv_wgrad_desc, M),
    F(cpcsv_wgrad_desc, N), F(cpcsv_wgrad_desc, Cs), F(cpcsv_wgrad_desc, ldy), F(cpcsv_wgrad_desc, lddw), F(cpcsv_wgrad_desc, ntaps),
    F(cpcsv_wgrad_desc, taps), F(cpcsv_wgrad_desc, MH), F(cpcsv_wgrad_desc, MW), F(cpcsv_wgrad_desc, IH), F(cpcsv_wgrad_desc, IW),
    F(cpcsv_wgrad_desc, sy), F(cpcsv_wgrad_desc, sx), F(cpcsv_wgrad_desc, up_shift), F(cpcsv_wgrad_desc, splits),
    F(cpcsv_wgrad_desc, dy_gather), F(cpcsv_wgrad_desc, DYH), F(cpcsv_wgrad_desc, DYW), F(cpcsv_wgrad_desc, dy_sy),
    F(cpcsv_wgrad_desc, dy_sx), F(cpcsv_wgrad_desc, legacy), F(cpcsv_wgrad_desc, accumulate), F(cpcsv_wgrad_desc, alpha),
    F(cpcsv_wgrad_desc, dY2), F(cpcsv_wgrad_desc, X2), F(cpcsv_wgrad_desc, M1), F(cpcsv_wgrad_desc, creal),
    F(cpcsv_wgrad_desc, wstride), F(cpcsv_wgrad_desc, dy_tapstride)};
const Field k_snjob[] = {F(cpcsv_sn_job, w), F(cpcsv_sn_job, u), F(cpcsv_sn_job, v), F(cpcsv_sn_job, work), F(cpcsv_sn_job, out),
                         F(cpcsv_sn_job, rows), F(cpcsv_sn_job, cols)};
const Field k_bng[] = {F(cpcsv_bn_groups, n), F(cpcsv_bn_groups, row), F(cpcsv_bn_groups, pstride), F(cpcsv_bn_groups, tile),
                       F(cpcsv_bn_groups, nph), F(cpcsv_bn_groups, TM), F(cpcsv_bn_groups, sigma)};
const Field k_upd[] = {
    F(cpcsv_update_desc, G), F(cpcsv_update_desc, p), F(cpcsv_update_desc, m), F(cpcsv_update_desc, v), F(cpcsv_update_desc, fwd),
    F(cpcsv_update_desc, bwd), F(cpcsv_update_desc, lin), F(cpcsv_update_desc, hyper), F(cpcsv_update_desc, beta1),
    F(cpcsv_update_desc, beta2), F(cpcsv_update_desc, eps), F(cpcsv_update_desc, dtype), F(cpcsv_update_desc, Cout),
    F(cpcsv_update_desc, Cin), F(cpcsv_update_desc, taps), F(cpcsv_update_desc, S), F(cpcsv_update_desc, Cin_s),
    F(cpcsv_update_desc, Cout_s), F(cpcsv_update_desc, sum), F(cpcsv_update_desc, tapmap), F(cpcsv_update_desc, masks),
    F(cpcsv_update_desc, nterms), F(cpcsv_update_desc, gw), F(cpcsv_update_desc, sigma), F(cpcsv_update_desc, u),
    F(cpcsv_update_desc, v_sn), F(cpcsv_update_desc, gscale), F(cpcsv_update_desc, step_add), F(cpcsv_update_desc, g_bf16)};
const Field k_scal[] = {F(cpcsv_scalar_list, x), F(cpcsv_scalar_list, w), F(cpcsv_scalar_list, n)};
const Field k_logit[] = {F(cpcsv_logit_groups, n), F(cpcsv_logit_groups, row), F(cpcsv_logit_groups, sigma), F(cpcsv_logit_groups, u), F(cpcsv_logit_groups, v)};
const Field k_wgp[] = {F(cpcsv_wgrad_piece, dz), F(cpcsv_wgrad_piece, x), F(cpcsv_wgrad_piece, ldz), F(cpcsv_wgrad_piece, ldx), F(cpcsv_wgrad_piece, M), F(cpcsv_wgrad_piece, _pad)};
const Field k_wgt[] = {F(cpcsv_wgrad_target, dW), F(cpcsv_wgrad_target, db), F(cpcsv_wgrad_target, N), F(cpcsv_wgrad_target, Kr), F(cpcsv_wgrad_target, piece0), F(cpcsv_wgrad_target, npieces), F(cpcsv_wgrad_target, block0), F(cpcsv_wgrad_target, bx)};
const Field k_wgl[] = {F(cpcsv_small_wgrad_list, ntargets), F(cpcsv_small_wgrad_list, npieces), F(cpcsv_small_wgrad_list, t), F(cpcsv_small_wgrad_list, p)};
const Field k_copy[] = {F(cpcsv_copy_list, dst), F(cpcsv_copy_list, src), F(cpcsv_copy_list, bytes), F(cpcsv_copy_list, n)};
const Field k_pkj[] = {F(cpcsv_pack_job, w), F(cpcsv_pack_job, fwd), F(cpcsv_pack_job, lin), F(cpcsv_pack_job, cout), F(cpcsv_pack_job, cin),
                       F(cpcsv_pack_job, cin_s), F(cpcsv_pack_job, cout_s), F(cpcsv_pack_job, blk0), F(cpcsv_pack_job, _pad)};
const Field k_pkl[] = {F(cpcsv_pack_list, n), F(cpcsv_pack_list, _pad), F(cpcsv_pack_list, j)};
const Field k_txj[] = {F(cpcsv_txt_job, type), F(cpcsv_txt_job, npass), F(cpcsv_txt_job, M), F(cpcsv_txt_job, T), F(cpcsv_txt_job, N), F(cpcsv_txt_job, K),
                       F(cpcsv_txt_job, ldx), F(cpcsv_txt_job, ldw), F(cpcsv_txt_job, ldy), F(cpcsv_txt_job, act), F(cpcsv_txt_job, A),
                       F(cpcsv_txt_job, eps), F(cpcsv_txt_job, momentum), F(cpcsv_txt_job, blk0), F(cpcsv_txt_job, nblk), F(cpcsv_txt_job, x),
                       F(cpcsv_txt_job, w), F(cpcsv_txt_job, bias), F(cpcsv_txt_job, y), F(cpcsv_txt_job, P), F(cpcsv_txt_job, Q)};
const Field k_txs[] = {F(cpcsv_txt_stage, njobs), F(cpcsv_txt_stage, _pad), F(cpcsv_txt_stage, job)};
const Field k_ch[] = {F(cpcsv_cond_head, ws), F(cpcsv_cond_head, nslabs), F(cpcsv_cond_head, ldws), F(cpcsv_cond_head, ws_rows), F(cpcsv_cond_head, pt),
                      F(cpcsv_cond_head, ldp), F(cpcsv_cond_head, MH), F(cpcsv_cond_head, MW), F(cpcsv_cond_head, ngroups), F(cpcsv_cond_head, count),
                      F(cpcsv_cond_head, feat0), F(cpcsv_cond_head, cond0), F(cpcsv_cond_head, galpha), F(cpcsv_cond_head, z), F(cpcsv_cond_head, y),
                      F(cpcsv_cond_head, dtype), F(cpcsv_cond_head, C), F(cpcsv_cond_head, Cs), F(cpcsv_cond_head, gamma), F(cpcsv_cond_head, beta),
                      F(cpcsv_cond_head, running_mean), F(cpcsv_cond_head, running_var), F(cpcsv_cond_head, stat_out), F(cpcsv_cond_head, pstride),
                      F(cpcsv_cond_head, bwd_sums), F(cpcsv_cond_head, act), F(cpcsv_cond_head, eps), F(cpcsv_cond_head, momentum)};
const Field k_chg[] = {F(cpcsv_cond_head_grad, dz), F(cpcsv_cond_head_grad, dF), F(cpcsv_cond_head_grad, dZt), F(cpcsv_cond_head_grad, dtype),
                       F(cpcsv_cond_head_grad, MH), F(cpcsv_cond_head_grad, MW), F(cpcsv_cond_head_grad, Cs), F(cpcsv_cond_head_grad, nfeat),
                       F(cpcsv_cond_head_grad, ncond), F(cpcsv_cond_head_grad, ngroups), F(cpcsv_cond_head_grad, count), F(cpcsv_cond_head_grad, feat0),
                       F(cpcsv_cond_head_grad, cond0)};
#undef F

template <int N>
int emit(const Field (&f)[N], int total, int* out, int cap) {
    const int need = 2 + 2 * N;
    if (!out) return need;
    if (cap < need) return -1002;
    out[0] = total;
    out[1] = N;
    for (int i = 0; i < N; ++i) { out[2 + 2 * i] = f[i].off; out[3 + 2 * i] = f[i].size; }
    return need;
}
}  // namespace

extern "C" int cpcsv_abi_layout(int which, int* out, int cap) {
    switch (which) {
        case CPCSV_ABI_TAP: return emit(k_tap, (int)sizeof(cpcsv_tap), out, cap);
        case CPCSV_ABI_GEMM_DESC: return emit(k_gemm, (int)sizeof(cpcsv_gemm_desc), out, cap);
        case CPCSV_ABI_WGRAD_DESC: return emit(k_wgrad, (int)sizeof(cpcsv_wgrad_desc), out, cap);
        case CPCSV_ABI_SN_JOB: return emit(k_snjob, (int)sizeof(cpcsv_sn_job), out, cap);
        case CPCSV_ABI_BN_GROUPS: return emit(k_bng, (int)sizeof(cpcsv_bn_groups), out, cap);
        case CPCSV_ABI_UPDATE_DESC: return emit(k_upd, (int)sizeof(cpcsv_update_desc), out, cap);
        case CPCSV_ABI_SCALAR_LIST: return emit(k_scal, (int)sizeof(cpcsv_scalar_list), out, cap);
        case CPCSV_ABI_COPY_LIST: return emit(k_copy, (int)sizeof(cpcsv_copy_list), out, cap);
        case CPCSV_ABI_LOGIT_GROUPS: return emit(k_logit, (int)sizeof(cpcsv_logit_groups), out, cap);
        case CPCSV_ABI_WGRAD_PIECE: return emit(k_wgp, (int)sizeof(cpcsv_wgrad_piece), out, cap);
        case CPCSV_ABI_WGRAD_TARGET: return emit(k_wgt, (int)sizeof(cpcsv_wgrad_target), out, cap);
        case CPCSV_ABI_SMALL_WGRAD_LIST: return emit(k_wgl, (int)sizeof(cpcsv_small_wgrad_list), out, cap);
        case CPCSV_ABI_PACK_JOB: return emit(k_pkj, (int)sizeof(cpcsv_pack_job), out, cap);
        case CPCSV_ABI_PACK_LIST: return emit(k_pkl, (int)sizeof(cpcsv_pack_list), out, cap);
        case CPCSV_ABI_TXT_JOB: return emit(k_txj, (int)sizeof(cpcsv_txt_job), out, cap);
        case CPCSV_ABI_TXT_STAGE: return emit(k_txs, (int)sizeof(cpcsv_txt_stage), out, cap);
        case CPCSV_ABI_COND_HEAD: return emit(k_ch, (int)sizeof(cpcsv_cond_head), out, cap);
        case CPCSV_ABI_COND_HEAD_GRAD: return emit(k_chg, (int)sizeof(cpcsv_cond_head_grad), out, cap);
        default: return -1001;
    }
}
