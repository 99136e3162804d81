"""ctypes binding of libcpcsv_hip.so (C ABI declared in include/cpcsv_hip.h).

The library is built in-tree by `make -C csrc` (see __graft_entry__.build). There is NO
fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPCSV_LIB_PATH") or os.path.join(_HERE, "libcpcsv_hip.so")      # (CPCSV_LIB_PATH: A/B builds of the same sources)
BN_SUM_COPIES = 8          # CPCSV_BN_SUM_COPIES in include/cpcsv_hip.h
MAX_TAPS = 16

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


class Tap(C.Structure):
    _fields_ = [("oy", C.c_int8), ("ox", C.c_int8), ("wtap", C.c_uint8), ("_pad", C.c_uint8)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("Cs", C.c_int),
        ("ldb", C.c_int), ("ldc", C.c_int), ("ntaps", C.c_int),
        ("taps", Tap * MAX_TAPS),
        ("MH", C.c_int), ("MW", C.c_int), ("IH", C.c_int), ("IW", C.c_int),
        ("sy", C.c_int), ("sx", C.c_int), ("up_shift", C.c_int), ("pool_rows", C.c_int),
        ("scatter", C.c_int), ("OH", C.c_int), ("OW", C.c_int), ("osy", C.c_int), ("osx", C.c_int),
        ("ooy", C.c_int), ("oox", C.c_int),
        ("alpha", C.c_void_p), ("bias", C.c_void_p), ("act", C.c_int),
        ("stats", C.c_void_p), ("ldstat", C.c_int), ("out_f32", C.c_int),
        ("splitk", C.c_int), ("ws", C.c_void_p), ("ldws", C.c_int), ("ws_rows", C.c_long), ("nphases", C.c_int),
        ("ph_tap0", C.c_int * 4), ("ph_ntaps", C.c_int * 4), ("ph_ooy", C.c_int * 4), ("ph_oox", C.c_int * 4),
        ("order_m_fast", C.c_int),
        ("ngroups", C.c_int), ("grow", C.c_int * 5), ("galpha", C.c_void_p * 4),
        ("addend", C.c_void_p), ("ldadd", C.c_int), ("korder", C.c_int), ("wstride", C.c_int), ("patch", C.c_int),
        ("slabs_only", C.c_int), ("bcol_rows", C.c_int), ("bcol_koff", C.c_int),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dY", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p),
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("Cs", C.c_int), ("ldy", C.c_int), ("lddw", C.c_int),
        ("ntaps", C.c_int), ("taps", Tap * MAX_TAPS),
        ("MH", C.c_int), ("MW", C.c_int), ("IH", C.c_int), ("IW", C.c_int),
        ("sy", C.c_int), ("sx", C.c_int), ("up_shift", C.c_int), ("splits", C.c_int),
        ("dy_gather", C.c_int), ("DYH", C.c_int), ("DYW", C.c_int), ("dy_sy", C.c_int), ("dy_sx", C.c_int),
        ("legacy", C.c_int), ("accumulate", C.c_int), ("alpha", C.c_void_p),
        ("dY2", C.c_void_p), ("X2", C.c_void_p), ("M1", C.c_int), ("creal", C.c_int),
        ("wstride", C.c_int), ("dy_tapstride", C.c_int),
    ]


class BnGroups(C.Structure):
    _fields_ = [("n", C.c_int), ("row", C.c_long * 5), ("pstride", C.c_long), ("tile", C.c_int * 5), ("nph", C.c_int), ("TM", C.c_int),
                ("sigma", C.c_void_p * 4)]


class SnJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("work", C.c_void_p), ("out", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int)]


class UpdateDesc(C.Structure):
    _fields_ = [
        ("G", C.c_void_p), ("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
        ("fwd", C.c_void_p), ("bwd", C.c_void_p), ("lin", C.c_void_p), ("hyper", C.c_void_p),
        ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("dtype", C.c_int), ("Cout", C.c_int), ("Cin", C.c_int), ("taps", C.c_int), ("S", C.c_int),
        ("Cin_s", C.c_int), ("Cout_s", C.c_int), ("sum", C.c_int),
        ("tapmap", C.c_int8 * MAX_TAPS), ("masks", C.c_uint16 * MAX_TAPS),
        ("nterms", C.c_int),
        ("gw", C.c_void_p * 4), ("sigma", C.c_void_p * 4), ("u", C.c_void_p * 4), ("v_sn", C.c_void_p * 4),
        ("gscale", C.c_float), ("step_add", C.c_float), ("g_bf16", C.c_int),
    ]


class ScalarList(C.Structure):
    _fields_ = [("x", C.c_void_p * 8), ("w", C.c_float * 8), ("n", C.c_int)]


class LogitGroups(C.Structure):
    _fields_ = [("n", C.c_int), ("row", C.c_int * 5), ("sigma", C.c_void_p * 4), ("u", C.c_void_p * 4), ("v", C.c_void_p * 4)]


SMALL_WG_TARGETS, SMALL_WG_PIECES = 16, 48


class WgradPiece(C.Structure):
    _fields_ = [("dz", C.c_void_p), ("x", C.c_void_p), ("ldz", C.c_int), ("ldx", C.c_int), ("M", C.c_int), ("_pad", C.c_int)]


class WgradTarget(C.Structure):
    _fields_ = [("dW", C.c_void_p), ("db", C.c_void_p), ("N", C.c_int), ("Kr", C.c_int), ("piece0", C.c_int), ("npieces", C.c_int),
                ("block0", C.c_int), ("bx", C.c_int)]


class SmallWgradList(C.Structure):
    _fields_ = [("ntargets", C.c_int), ("npieces", C.c_int), ("t", WgradTarget * SMALL_WG_TARGETS), ("p", WgradPiece * SMALL_WG_PIECES)]


PACK_JOBS = 16


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("fwd", C.c_void_p), ("lin", C.c_void_p), ("cout", C.c_int), ("cin", C.c_int), ("cin_s", C.c_int),
                ("cout_s", C.c_int), ("blk0", C.c_int), ("_pad", C.c_int)]


class PackList(C.Structure):
    _fields_ = [("n", C.c_int), ("_pad", C.c_int), ("j", PackJob * PACK_JOBS)]


class CopyList(C.Structure):
    _fields_ = [("dst", C.c_void_p * 8), ("src", C.c_void_p * 8), ("bytes", C.c_long * 8), ("n", C.c_int)]


TXT_MAX_JOBS, TXT_MAX_ROWS = 8, 256
TXT_DENSE, TXT_CA, TXT_GRU_FWD, TXT_PREP, TXT_JOINT, TXT_DFL_BWD, TXT_BN_BWD, TXT_GRU_BWD, TXT_CA_BWD = 1, 2, 3, 4, 5, 6, 7, 8, 9


class TxtJob(C.Structure):
    _fields_ = [("type", C.c_int), ("npass", C.c_int), ("M", C.c_int * 2), ("T", C.c_int * 2), ("N", C.c_int), ("K", C.c_int),
                ("ldx", C.c_int), ("ldw", C.c_int), ("ldy", C.c_int), ("act", C.c_int), ("A", C.c_int * 8),
                ("eps", C.c_float), ("momentum", C.c_float), ("blk0", C.c_int), ("nblk", C.c_int),
                ("x", C.c_void_p * 2), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p * 2),
                ("P", (C.c_void_p * 6) * 2), ("Q", C.c_void_p * 4)]


class TxtStage(C.Structure):
    _fields_ = [("njobs", C.c_int), ("_pad", C.c_int), ("job", TxtJob * TXT_MAX_JOBS)]


class CondHead(C.Structure):
    _fields_ = [("ws", C.c_void_p), ("nslabs", C.c_int), ("ldws", C.c_int), ("ws_rows", C.c_long), ("pt", C.c_void_p), ("ldp", C.c_int),
                ("MH", C.c_int), ("MW", C.c_int), ("ngroups", C.c_int), ("count", C.c_int * 4), ("feat0", C.c_int * 4), ("cond0", C.c_int * 4),
                ("galpha", C.c_void_p * 4), ("z", C.c_void_p), ("y", C.c_void_p), ("dtype", C.c_int), ("C", C.c_int), ("Cs", C.c_int),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("stat_out", C.c_void_p),
                ("pstride", C.c_long), ("bwd_sums", C.c_int), ("act", C.c_int), ("eps", C.c_float), ("momentum", C.c_float)]


class CondHeadGrad(C.Structure):
    _fields_ = [("dz", C.c_void_p), ("dF", C.c_void_p), ("dZt", C.c_void_p), ("dtype", C.c_int), ("MH", C.c_int), ("MW", C.c_int), ("Cs", C.c_int),
                ("nfeat", C.c_int), ("ncond", C.c_int), ("ngroups", C.c_int), ("count", C.c_int * 4), ("feat0", C.c_int * 4), ("cond0", C.c_int * 4)]


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float

# name -> argtypes (all return int unless noted); must list EVERY symbol of include/cpcsv_hip.h
SIGNATURES = {
    "cpcsv_gemm_mtile": [C.POINTER(GemmDesc)],
    "cpcsv_gemm_ntile": [C.POINTER(GemmDesc)],
    "cpcsv_gemm_nt": [C.POINTER(GemmDesc), _P],
    "cpcsv_wgrad_tn": [C.POINTER(WgradDesc), _P],
    "cpcsv_pack_weight": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "cpcsv_unpack_wgrad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "cpcsv_rank1_sub": [_P, _P, _P, _P, _P, _L, _L, _P],
    "cpcsv_wgrad_dot": [_P, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    "cpcsv_pack_weight_sum": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "cpcsv_unpack_wgrad_sum": [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    "cpcsv_spectral_sigma": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cpcsv_sn_multi_blocks": [_I, _I, _I],
    "cpcsv_spectral_sigma_multi": [_P, _I, _P, _I, _P, _I, _I, _P],
    "cpcsv_spectral_sigma_multi1": [_P, _I, _P, _I, _P, _P, _I, _P],
    "cpcsv_bn_finalize": [_P, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _I, _P, _P, _P],
    "cpcsv_bn_apply": [_P, _P, _I, _P, _P, _L, _I, _I, _I, _P, _P],
    "cpcsv_bn_apply_partials": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _P, _P],
    "cpcsv_bn_bwd_reduce": [_P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P],
    "cpcsv_colsum": [_P, _I, _P, _L, _I, _I, _P],
    "cpcsv_concat_pad": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _L, _I, _P],
    "cpcsv_bn_bwd_apply": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P, _F, _P, _P],
    "cpcsv_act_bwd": [_P, _P, _P, _I, _L, _I, _P],
    "cpcsv_gate_fwd": [_P, _P, _P, _I, _L, _P],
    "cpcsv_gate_bwd": [_P, _P, _P, _P, _P, _I, _L, _P],
    "cpcsv_planar_to_nhwc": [_P, _I, _P, _I, _I, _I, _L, _L, _L, _I, _I, _I, _P],
    "cpcsv_nhwc_to_planar": [_P, _I, _P, _I, _I, _I, _L, _L, _L, _I, _I, _I, _P],
    "cpcsv_ingest_u8": [_P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _I, _I, _P, _P, _P],
    "cpcsv_copy2d": [_P, _I, _L, _I, _P, _I, _L, _I, _L, _I, _I, _P],
    "cpcsv_im2col": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "cpcsv_batch_prep": [_P, _L, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpcsv_cond_concat": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "cpcsv_cond_triplet": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "cpcsv_cond_triplet_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "cpcsv_bce_groups": [_P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P],
    "cpcsv_mean_t": [_P, _P, _I, _I, _I, _L, _P],
    "cpcsv_mean_t_bwd": [_P, _P, _I, _I, _I, _L, _P],
    "cpcsv_fill_zero": [_P, _L, _P],
    "cpcsv_dense_rows": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P],
    "cpcsv_gru_step_fwd": [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _P],
    "cpcsv_dense_rows_wgrad": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P],
    "cpcsv_dense_rows_wgrad_multi": [C.POINTER(SmallWgradList), _P],
    "cpcsv_gru_gates_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cpcsv_gru_gates_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cpcsv_dfl1d_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpcsv_dfl1d_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpcsv_reparam_fwd": [_P, _P, _P, _P, _L, _P],
    "cpcsv_reparam_bwd": [_P, _P, _P, _P, _P, _L, _I, _P],
    "cpcsv_bce_fwd": [_P, _P, _P, _P, _L, _P],
    "cpcsv_mlsm_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "cpcsv_lincomb_fwd": [C.POINTER(ScalarList), _P, _P],
    "cpcsv_lincomb_bwd": [_P, C.POINTER(ScalarList), _P, _P],
    "cpcsv_copy_many": [C.POINTER(CopyList), _P],
    "cpcsv_pack_dense_many": [C.POINTER(PackList), _P],
    "cpcsv_logit_head_fwd": [_P, _P, _P, _P, _I, _I, _I, C.POINTER(LogitGroups), _P],
    "cpcsv_logit_head_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(LogitGroups), _P],
    "cpcsv_logit_head_scratch": [_I, _I],
    "cpcsv_logit_head_wgrad": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(LogitGroups), _P],
    "cpcsv_text_stage": [C.POINTER(TxtStage), _P],
    "cpcsv_cond_head_fwd": [C.POINTER(CondHead), _P],
    "cpcsv_cond_head_max_samples": [],
    "cpcsv_cond_head_bwd": [C.POINTER(CondHeadGrad), _P],
    "cpcsv_kl_fwd": [_P, _P, _P, _P, _P, _L, _P],
    "cpcsv_mse_fwd": [_P, _P, _I, _P, _P, _P, _L, _L, _P],
    "cpcsv_scale_by": [_P, _P, _I, _P, _F, _L, _I, _P],
    "cpcsv_adam_step": [_P, _P, _I, _L, _P, _P, _P, _F, _F, _F, _P],
    "cpcsv_thin_supported": [_I, _I, _I, _I, _I],
    "cpcsv_thin3x3_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "cpcsv_thin3x3_dgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpcsv_thin3x3_wgrad_slabs": [_I, _I, _I, _I],
    "cpcsv_thin3x3_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpcsv_thin4x4s2_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "cpcsv_thin4x4s2_dgrad": [_P, _P, _P, _P, _I, _I, _I, _P],
    "cpcsv_thin4x4s2_wgrad_slabs": [_I, _I, _I],
    "cpcsv_thin4x4s2_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "cpcsv_layer_update": [_P, _P],
    "cpcsv_adam_chunk": [],
    "cpcsv_set_deterministic": [_I],
    "cpcsv_abi_layout": [_I, C.POINTER(C.c_int), _I],
    "cpcsv_set_wgrad_linear": [_I],
    "cpcsv_version": [],
    "cpcsv_arch": [],
}

_lib = None

# which-code of cpcsv_abi_layout -> the ctypes mirror of that struct (CPCSV_ABI_* in include/cpcsv_hip.h)
ABI_STRUCTS = {0: Tap, 1: GemmDesc, 2: WgradDesc, 3: SnJob, 4: BnGroups, 5: UpdateDesc, 6: ScalarList, 7: CopyList, 8: LogitGroups, 9: WgradPiece, 10: WgradTarget, 11: SmallWgradList,
               12: PackJob, 13: PackList, 14: TxtJob, 15: TxtStage, 16: CondHead, 17: CondHeadGrad}


def layout_of(struct):
    """[sizeof, nfields, (offset, size) per field ...] of a ctypes Structure - the format cpcsv_abi_layout reports."""
    out = [C.sizeof(struct), len(struct._fields_)]
    for name, _ in struct._fields_:
        f = getattr(struct, name)
        out += [f.offset, f.size]
    return out


def verify_layout(lib):
    """Compare every descriptor struct of this binding with the layout the LIBRARY was compiled with (cpcsv_abi_layout). A
    field added to include/cpcsv_hip.h but not here (or the reverse) would shift everything behind it without any error."""
    buf = (C.c_int * 256)()
    for which, struct in ABI_STRUCTS.items():
        n = lib.cpcsv_abi_layout(which, buf, len(buf))
        if n < 2:
            raise RuntimeError("cpcsv_abi_layout(%d) failed with code %d" % (which, n))
        theirs, mine = list(buf[:n]), layout_of(struct)
        if theirs != mine:
            names = [f[0] for f in struct._fields_]
            bad = next((names[i] for i in range(min(theirs[1], mine[1]))
                        if theirs[2 + 2 * i:4 + 2 * i] != mine[2 + 2 * i:4 + 2 * i]), "(field count / struct size)")
            raise RuntimeError("ABI mismatch between %s and libcpcsv_hip.so at %s.%s: the library was built from another "
                               "include/cpcsv_hip.h (rebuild: make -C csrc)" % (os.path.basename(__file__), struct.__name__, bad))


def load():
    """Load the shared object (once). Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libcpcsv_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cpcstoryvisualization-pytorch_amd/csrc`. There is no CPU/PyTorch fallback." % LIB_PATH)
    # torch ships its own libamdhip64.so: it has to be in the process BEFORE this library is mapped, otherwise the
    # loader resolves our HIP symbols to /opt/rocm's copy and the kernels launch on a second runtime that knows none of
    # torch's streams or allocations (first launch fails with hipErrorNoDevice)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "cpcsv_arch" else (C.c_long if name == "cpcsv_logit_head_scratch" else C.c_int)
    verify_layout(lib)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))
