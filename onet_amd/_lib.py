"""ctypes binding of libonet_hip.so (C ABI: include/onet_hip.h).

The prototypes are PARSED from the header, so the Python side cannot drift from
the ABI.  There is no CPU fallback: if the library cannot be loaded (or built),
importing the compute path raises."""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "onet_hip.h")
# ONET_HIP_LIB: another build of the same library (onet_amd.build --variant ...) for same-box A/B runs
LIBPATH = os.environ.get("ONET_HIP_LIB") or os.path.join(HERE, "libonet_hip.so")

_CT = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    protos = {}
    for m in re.finditer(r"((?:const\s+)?\w+\s*\*?)\s*(onet_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.endswith("*"):
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CT[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(?:const\s+)?(\w+)\s*(\*?)\s*(\w+)$", a)
                if not mm:
                    raise ValueError(f"cannot parse argument {a!r} of {name}")
                base, ptr, an = mm.groups()
                argtypes.append(ctypes.c_void_p if ptr else _CT[base])
                argnames.append(an)
        protos[name] = (restype, argtypes, argnames)
    return protos


class OnetHipError(RuntimeError):
    pass


_lib = None
_protos = None


def load(build_if_missing: bool = True):
    """Load (building in-tree first if needed) libonet_hip.so.  Raises if impossible."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        if not build_if_missing:
            raise OnetHipError(f"{LIBPATH} not built; run `python -m onet_amd.build`")
        from . import build as _build
        _build.build()
    lib = ctypes.CDLL(LIBPATH)
    _protos = parse_header()
    for name, (restype, argtypes, _) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # header/ABI drift must be loud
            raise OnetHipError(f"libonet_hip.so does not export {name} declared in onet_hip.h") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().onet_last_error().decode()


# Optional per-launch timing hook (bench.py): an object with begin() -> token and end(name, nbytes, token), installed by
# onet_amd.ops while a profile is being taken.  None (default): no overhead.
PROFILE_HOOK = None


def call(name: str, *args, nbytes=None):
    """Call an int-returning entry point; raise OnetHipError on a negative code.  `nbytes`: the algorithmic HBM bytes of
    this launch (streaming kernels), recorded with its duration when a profile hook is installed."""
    lib = load()
    hook = PROFILE_HOOK
    tok = hook.begin() if hook is not None else None
    rc = getattr(lib, name)(*args)
    if tok is not None:
        hook.end(name, nbytes, tok)
    if rc != 0:
        raise OnetHipError(f"{name} failed ({rc}): {lib.onet_last_error().decode()}")
    return rc
