#!/usr/bin/env python3
"""Regenerate the ctypes struct stubs of INTEGRATION.md §2 from cpcsv/_lib.py (the binding the tests exercise), so the
documented binding can never lag behind include/cpcsv_hip.h again.    python tools/gen_integration_stub.py [--check]
The block between the BEGIN/END GENERATED STRUCTS markers is replaced; --check only compares (exit 1 on drift)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cpcstoryvisualization-pytorch_amd"))
BEGIN, END = "# --- BEGIN GENERATED STRUCTS (tools/gen_integration_stub.py) ---", "# --- END GENERATED STRUCTS ---"

_SIMPLE = {C.c_void_p: "C.c_void_p", C.c_int: "C.c_int", C.c_long: "C.c_long", C.c_float: "C.c_float", C.c_int8: "C.c_int8",
           C.c_uint8: "C.c_uint8", C.c_uint16: "C.c_uint16"}


def tname(t):
    if t in _SIMPLE:
        return _SIMPLE[t]
    if hasattr(t, "_length_"):
        return "%s * %d" % (tname(t._type_), t._length_)
    return t.__name__


def render():
    from cpcsv import _lib
    cnames = {0: "cpcsv_tap", 1: "cpcsv_gemm_desc", 2: "cpcsv_wgrad_desc", 3: "cpcsv_sn_job", 4: "cpcsv_bn_groups", 5: "cpcsv_update_desc", 6: "cpcsv_scalar_list", 7: "cpcsv_copy_list", 8: "cpcsv_logit_groups", 9: "cpcsv_wgrad_piece", 10: "cpcsv_wgrad_target", 11: "cpcsv_small_wgrad_list", 12: "cpcsv_pack_job", 13: "cpcsv_pack_list", 14: "cpcsv_txt_job", 15: "cpcsv_txt_stage", 16: "cpcsv_cond_head", 17: "cpcsv_cond_head_grad"}
    out = [BEGIN]
    for which, st in _lib.ABI_STRUCTS.items():
        out.append("class %s(C.Structure):          # mirrors %s in include/cpcsv_hip.h, field for field" % (st.__name__, cnames[which]))
        line = "    _fields_ = ["
        for i, (name, t) in enumerate(st._fields_):
            item = '("%s", %s)' % (name, tname(t)) + ("," if i + 1 < len(st._fields_) else "]")
            if len(line) + len(item) + 1 > 118:
                out.append(line.rstrip())
                line = "                "
            line += item + " "
        out.append(line.rstrip())
        out.append("")
    out += ["def check_layout(lib):              # refuse to run against a library built from another header",
            "    lib.cpcsv_abi_layout.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]",
            "    buf = (C.c_int * 256)()",
            "    for which, st in enumerate((%s)):" % ", ".join(s.__name__ for s in _lib.ABI_STRUCTS.values()),
            "        n = lib.cpcsv_abi_layout(which, buf, 256)",
            "        mine = [C.sizeof(st), len(st._fields_)]",
            "        for name, _ in st._fields_:",
            "            mine += [getattr(st, name).offset, getattr(st, name).size]",
            "        assert n > 0 and list(buf[:n]) == mine, st.__name__",
            END]
    return "\n".join(out)


def main():
    path = os.path.join(REPO, "INTEGRATION.md")
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END) + len(END)
    new = text[:a] + render() + text[b:]
    if "--check" in sys.argv:
        if new != text:
            print("INTEGRATION.md struct stubs are stale: run python tools/gen_integration_stub.py")
            return 1
        return 0
    open(path, "w").write(new)
    return 0


if __name__ == "__main__":
    sys.exit(main())
