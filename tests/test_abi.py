"""The C-ABI library loads and exports every symbol include/madiff_hip.h declares, and the ctypes
signatures in madiff_amd/lib.py agree with the header (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    src = open(os.path.join(ROOT, "include", "madiff_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(madiff_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a.strip() for a in m.group(3).split(",")]
        if args == ["void"]:
            args = []
        decls[m.group(2)] = args
    return decls


def ctype_of(arg):
    if "*" in arg or "madiff_stream_t" in arg:
        return C.c_void_p
    if re.match(r"(const\s+)?long\b", arg):
        return C.c_long
    if re.match(r"(const\s+)?float\b", arg):
        return C.c_float
    if re.match(r"(const\s+)?int\b", arg):
        return C.c_int
    raise AssertionError(f"unparsed argument {arg!r}")


def test_library_exports_and_signatures():
    from madiff_amd import lib
    decls = header_decls()
    assert len(decls) >= 24
    for name, args in decls.items():
        assert lib.exported(name), f"{name} declared in the header but not exported by the .so"
        if name in ("madiff_version", "madiff_arch"):
            continue
        assert name in lib._SIGS, f"{name} has no ctypes signature"
        want = [ctype_of(a) for a in args]
        assert lib._SIGS[name] == want, f"{name}: ctypes signature differs from the header"
    for name in lib._SIGS:
        assert name in decls, f"{name} bound in lib.py but not declared in the header"
    assert lib.version() == 1 and lib.arch() == "gfx950"
