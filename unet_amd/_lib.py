"""ctypes binding of libunet_hip.so, generated from include/unet_hip.h.

The prototypes are parsed from the header at import, so the header stays the single source of
truth for the C ABI (tests check that every declared symbol is exported).  There is no fallback:
if the shared library is missing, every op raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UNET_HIP_LIB") or os.path.join(_HERE, "libunet_hip.so")   # override: diagnostic builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "unet_hip.h")

_lib = None
_protos = None


class UNetHipError(RuntimeError):
    pass


def _ctype(arg: str):
    if "*" in arg:
        return ctypes.c_void_p
    if "long long" in arg:
        return ctypes.c_longlong
    if "double" in arg:
        return ctypes.c_double
    if "float" in arg:
        return ctypes.c_float
    return ctypes.c_int


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(unet_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    return protos


def protos():
    global _protos
    if _protos is None:
        _protos = parse_header()
    return _protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UNetHipError(
                f"{LIB_PATH} is missing: build it with `make -C unet_amd/csrc` (hipcc, gfx950). "
                "unet_amd has no CPU or eager fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in protos().items():
            fn = getattr(L, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().unet_last_error().decode(errors="replace")
        raise UNetHipError(f"{what} failed (rc={rc}): {msg}")
