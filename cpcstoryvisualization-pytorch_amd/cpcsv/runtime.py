"""Process-wide runtime state of the HIP path: compute dtype, stream handle, small helpers."""
import os

import torch

from . import _lib

def _initial_env_has(name, value):
    """Was NAME=value in the environment this process was STARTED with? (/proc/self/environ is the initial block;
    later os.environ / putenv changes do not show up there.)"""
    try:
        with open("/proc/self/environ", "rb") as fh:
            return ("%s=%s" % (name, value)).encode() in fh.read().split(b"\0")
    except OSError:
        return False


def _hip_runtime_started():
    """The HIP/HSA runtime opens /dev/kfd when it initialises - also when torch only asked torch.cuda.is_available(),
    which does NOT flip torch.cuda.is_initialized(). The CLR flags are read at that moment."""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink("/proc/self/fd/" + fd) == "/dev/kfd":
                    return True
            except OSError:
                pass
    except OSError:
        pass
    return torch.cuda.is_initialized()


# see graphs.many_graphs_safe(): DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment BEFORE the HIP runtime reads
# its flags. Trusted only if the process was started with it, or if it is set now and the runtime provably has not
# started yet; otherwise the tested <= 2400-node scheme (no-grad pass + critic graphs) is used and a warning printed.
_PC = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
if _initial_env_has(_PC, "0") or (os.environ.get("CPCSV_PACKET_CAPTURE_EARLY") == "0" and os.environ.get(_PC) == "0"):
    # in the start-up environment, or set by an entry point (bench.py, tests/conftest.py, __graft_entry__.py) that
    # checked torch had not been imported yet
    PACKET_CAPTURE_OFF = True
elif not _hip_runtime_started():
    os.environ.setdefault(_PC, "0")
    PACKET_CAPTURE_OFF = os.environ[_PC] == "0"
else:
    PACKET_CAPTURE_OFF = False
    print("[cpcsv] the HIP runtime was initialised before cpcsv was imported and %s=0 was not in the start-up "
          "environment: capturing only the no-grad and critic graphs (set %s=0 before starting python, or import the "
          "package before touching torch.cuda, to capture every piece)" % (_PC, _PC))

# kernel arguments in device memory (the ROCm 7.2 default on gfx950; with host-memory kernargs the step is 1.2 ms slower:
# profiles/r03_experiments.txt) - stated explicitly where the runtime has not started yet, for images whose default differs
if not _hip_runtime_started():
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

_STATE = {"dtype": os.environ.get("CPCSV_DTYPE", "bf16"), "subpixel": os.environ.get("CPCSV_SUBPIXEL", "1") != "0"}


def set_compute_dtype(name):
    """'bf16' (MFMA bf16 operands, fp32 accumulate — the performance mode BASELINE.json names)
    or 'fp32' (exact f32 MFMA — the parity mode; the reference is fp32 only)."""
    if name not in ("bf16", "fp32"):
        raise ValueError("compute dtype must be 'bf16' or 'fp32'")
    _STATE["dtype"] = name


def set_deterministic(on):
    """Reproducible reductions in every kernel (include/cpcsv_hip.h: cpcsv_set_deterministic): parity tests and
    eager-vs-graph comparisons then agree bit for bit run to run. Slower on the big maps; off by default
    (CPCSV_DETERMINISTIC=1 turns it on at import)."""
    _STATE["det"] = bool(on)
    return bool(_lib.load().cpcsv_set_deterministic(int(bool(on))))


def deterministic():
    return _STATE.get("det", False)


def set_subpixel(on):
    """Sub-pixel form of upsample+conv3x3 (2.25x fewer FLOPs; default on). Read when a layer is first planned."""
    _STATE["subpixel"] = bool(on)


def subpixel():
    return _STATE["subpixel"]


def compute_dtype_name():
    return _STATE["dtype"]


def tdtype():
    return torch.bfloat16 if _STATE["dtype"] == "bf16" else torch.float32


def dcode(t=None):
    if t is None:
        return _lib.BF16 if _STATE["dtype"] == "bf16" else _lib.F32
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    raise TypeError("unsupported dtype %s" % t.dtype)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_current_device = torch._C._cuda_getDevice if hasattr(torch._C, "_cuda_getDevice") else torch.cuda.current_device


_FORCED = [None]          # raw stream handle the kernel wrappers launch on instead of torch's current stream
_WGRAD = [None]           # torch.cuda.Stream that takes the weight-gradient side of LayerFn.backward (None = inline)
_KEEP = []


def stream():
    """hipStream_t the next kernel goes to: torch's current stream (raw handle; ~0.2 us instead of ~9 us through
    torch.cuda.current_stream()), or the stream forced by `forced_stream`."""
    f = _FORCED[0]
    if f is not None:
        return f
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


class forced_stream:
    """Launch the kernels of the body on `s` without switching torch's current stream (no allocator pool switch).
    The caller orders `s` against the current stream and keeps every tensor the body touches alive until the join."""

    def __init__(self, s):
        self.raw = s.cuda_stream

    def __enter__(self):
        self.prev, _FORCED[0] = _FORCED[0], self.raw

    def __exit__(self, *a):
        _FORCED[0] = self.prev


def set_wgrad_stream(s):
    """Weight-gradient GEMMs, their unpack and the bias column sums of every LayerFn.backward run on stream `s` from
    now on (None: inline). They feed only the optimizer, so taking them off the stream that carries the data-gradient
    chain shortens the critical path of the backward pass. Used while the generator's backward is CAPTURED
    (cpcsv/graphs.py): there the fork costs nothing on the host and becomes a parallel branch of the graph. The
    caller joins `s` and calls release_kept() afterwards."""
    _WGRAD[0] = s
    _SIDE_EPOCH[0] += 1


def wgrad_stream():
    return _WGRAD[0]


_SIDE_EPOCH = [0]


def note_side_write(side, *params):
    """The weight-gradient branch `side` has just been handed a launch that read-modify-writes the master-layout .grad of
    `params` (small dense layers: cpcsv_wgrad_tn with accumulate, cpcsv_colsum). A later launch on ANOTHER stream that adds to
    the same gradient - the <= 64-row pass of the same layer runs cpcsv_dense_rows_wgrad inline, non-atomically - has to
    wait for it (wait_side_writes); the other order is covered by the fork (the branch starts behind the current stream)."""
    ev = torch.cuda.Event()
    ev.record(side)
    for p in params:
        if p is not None:
            p._cpcsv_side_ev = (_SIDE_EPOCH[0], ev)


def wait_side_writes(*params):
    """Order the current stream behind the weight-gradient branch's pending writes to these parameters' gradients (same
    backward pass only: an older pass was joined long ago)."""
    seen = set()
    for p in params:
        tag = getattr(p, "_cpcsv_side_ev", None) if p is not None else None
        if tag is not None and tag[0] == _SIDE_EPOCH[0] and id(tag[1]) not in seen:
            seen.add(id(tag[1]))
            if _FORCED[0] is None:
                torch.cuda.current_stream().wait_event(tag[1])
            else:
                torch.cuda.ExternalStream(_FORCED[0]).wait_event(tag[1])


_LATE = [None, []]        # stream + parked launches of the fused per-layer updates that wait for the tail of the backward


def set_late_stream(s):
    """Generator backward, captured form: the fused optimiser launches of the decoder's layers (HBM-bound, ~1 ms per step)
    are parked while the data-gradient chain walks the decoder - where they would compete with its GEMMs - and go out on
    stream `s` once the chain reaches the text / motion encoders, whose long chain of tiny launches leaves the GPU idle."""
    _LATE[0], _LATE[1] = s, []


def late_stream():
    return _LATE[0]


def defer_late(fn):
    _LATE[1].append(fn)


def flush_late(wside):
    """Enqueue the parked launches on the late stream, ordered after everything the current stream and the
    weight-gradient stream hold so far."""
    s, fns = _LATE
    if s is None or not fns:
        return
    fork_to(s)
    if wside is not None:
        s.wait_stream(wside)
    with forced_stream(s):
        for fn in fns:
            fn()
    del fns[:]


_SMALL_WG = [False, []]   # (parking on?, parked pieces) of the small dense layers' weight gradients


def defer_small_wgrads(on):
    """While on, the <= 64-row fp32 dense layers (text / motion encoders, GRU cells) PARK their weight-gradient launches
    (park_small_wgrad) instead of issuing one ~12-25 us launch each on the backward's critical chain; flush_small_wgrads() issues
    them as ONE launch (cpcsv_dense_rows_wgrad_multi). Their results feed only the optimiser."""
    if on:
        del _SMALL_WG[1][:]          # (whatever an aborted earlier pass left parked is stale: it must not reach THIS pass's launch)
    _SMALL_WG[0] = bool(on) and os.environ.get("CPCSV_SMALL_WG_BATCH", "1") != "0"


def small_wgrads_deferred():
    return _SMALL_WG[0]


def park_small_wgrad(dW, db, dz, x, M, N, Kr):
    _SMALL_WG[1].append((dW, db, dz, x, int(M), int(N), int(Kr)))


def discard_small_wgrads():
    """Drop whatever is parked (pieces of a backward pass that never reached its flush)."""
    del _SMALL_WG[1][:]


def flush_small_wgrads():
    """One launch (per 16 weights / 48 pieces) for everything parked, on the current stream; pieces of one weight keep their order.
    A weight's pieces must agree on (db, N, Kr) (one bias gradient per weight: a piece without one next to a piece with one would
    have the launch add the bias sums of BOTH); a weight with more pieces than one launch holds is continued in the next launch
    (same stream: its read-modify-write stays ordered). The parked list is cleared once the batch has been validated."""
    jobs = list(_SMALL_WG[1])
    if not jobs:
        return
    import ctypes as C
    from . import kernels as K
    order, by_w = [], {}
    for j in jobs:
        key = j[0].data_ptr()
        if key not in by_w:
            by_w[key] = []
            order.append(key)
        by_w[key].append(j)
    for key in order:                            # validated BEFORE the list is cleared: a refused batch stays parked for the caller to inspect
        pieces = by_w[key]
        dbs = {(q[1].data_ptr() if q[1] is not None else None) for q in pieces}
        if len(dbs) > 1 or len({(q[5], q[6]) for q in pieces}) > 1:
            raise RuntimeError("parked weight-gradient pieces of one weight disagree on their bias gradient / shape")
    del _SMALL_WG[1][:]
    state = {"lst": _lib.SmallWgradList(), "nt": 0, "npc": 0}

    def launch():
        lst = state["lst"]
        lst.ntargets, lst.npieces = state["nt"], state["npc"]
        K._call("cpcsv_dense_rows_wgrad_multi", C.byref(lst), stream())
        state["lst"], state["nt"], state["npc"] = _lib.SmallWgradList(), 0, 0
    for key in order:
        pieces = by_w[key]
        while pieces:
            if state["nt"] == _lib.SMALL_WG_TARGETS or state["npc"] == _lib.SMALL_WG_PIECES:
                launch()
            room = _lib.SMALL_WG_PIECES - state["npc"]
            if len(pieces) > room and state["npc"] > 0 and len(pieces) <= _lib.SMALL_WG_PIECES:
                launch()                                        # the whole weight fits a fresh launch: keep it together
                room = _lib.SMALL_WG_PIECES
            part, pieces = pieces[:room], pieces[room:]
            lst, nt, npc = state["lst"], state["nt"], state["npc"]
            t = lst.t[nt]
            t.dW, t.db, t.N, t.Kr, t.piece0, t.npieces = part[0][0].data_ptr(), ptr(part[0][1]), part[0][5], part[0][6], npc, len(part)
            for (_, _, dz, x, m, n, kr) in part:
                pc = lst.p[npc]
                pc.dz, pc.x, pc.ldz, pc.ldx, pc.M = dz.data_ptr(), x.data_ptr(), dz.shape[1], x.shape[1], m
                npc += 1
            state["nt"], state["npc"] = nt + 1, npc
    if state["nt"]:
        launch()


_BRANCH = [0, None]       # (id, role) of the generator pass being enqueued when its two halves run on two streams


class branch:
    """The story half and the image half of a generator pass are independent (different inputs, same weights); run
    on two streams they fill the CUs that the many small layers of one half leave idle. Inside `with branch(i, role)`
    the layers use per-branch descriptors/workspaces, and BatchNorm layers order their running-statistics updates:
    the 'first' half records an event after its update, the 'second' half waits for it (the reference updates them in
    that order, and r <- (1-m) r + m b does not commute)."""

    def __init__(self, i, role):
        self.new = [i, role]

    def __enter__(self):
        self.old = list(_BRANCH)
        _BRANCH[:] = self.new

    def __exit__(self, *a):
        _BRANCH[:] = self.old


def branch_id():
    return _BRANCH[0]


def branch_role():
    return _BRANCH[1]


_GROUPS = [None]


class row_groups:
    """Inside `with row_groups((n0, n1, ...))` every fused layer treats its input as several PASSES back to back along
    the leading dimension (n0 images / rows of the first pass, then n1, ...): one set of launches, but BatchNorm
    statistics, running-statistics updates and spectral-norm iterations per pass and in pass order - what the reference
    does with separate calls (critic real / fake batches, miscc/utils.py:70-84; the story and image halves of a
    generator pass, model.py:348,426)."""

    def __init__(self, counts):
        self.new = tuple(int(c) for c in counts) if counts is not None else None

    def __enter__(self):
        self.old, _GROUPS[0] = _GROUPS[0], self.new

    def __exit__(self, *a):
        _GROUPS[0] = self.old


def current_groups():
    return _GROUPS[0]


def fork_to(side):
    """Order `side` after everything enqueued so far on torch's current stream."""
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)


def keep_alive(*tensors):
    """Hold references to tensors a side stream still reads until release_kept(): their memory must not be handed to
    later allocations of the main stream before the side stream is joined."""
    _KEEP.append(tensors)


def release_kept():
    _KEEP.clear()


def pad8(n):
    return (int(n) + 7) // 8 * 8


def ptr(t):
    return None if t is None else t.data_ptr()


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("cpcsv ops run on the GPU only (HIP kernels); got a %s tensor. "
                           "There is no CPU fallback in the product path." % t.device)


if os.environ.get("CPCSV_DETERMINISTIC", "0") == "1":
    set_deterministic(True)
