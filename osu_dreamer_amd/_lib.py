"""ctypes binding of libosudreamer_hip.so — the C ABI declared in include/osu_dreamer_hip.h.

The binding is generated from the header itself, so Python argtypes cannot drift from the
C declarations.  There is NO fallback: if the HIP library is missing, `lib()` raises.
(`use_library()` exists so the test-suite can point the same host code at the SIMT
emulator build of the same kernel sources; nothing in the package calls it.)
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "osu_dreamer_hip.h")
DEFAULT_SO = os.environ.get("OSU_DREAMER_HIP_LIB", os.path.join(_HERE, "libosudreamer_hip.so"))

OD_F32, OD_BF16, OD_F32X3, OD_F32X3W, OD_F16 = 0, 1, 2, 3, 4
OD_EPI_NONE, OD_EPI_SILU = 0, 1
OD_ACT_NONE, OD_ACT_SILU = 0, 1

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p, "const float*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p, "void**": ctypes.POINTER(ctypes.c_void_p),
    "const char*": ctypes.c_char_p,
    "long*": ctypes.POINTER(ctypes.c_long), "int*": ctypes.POINTER(ctypes.c_int), "double*": ctypes.POINTER(ctypes.c_double),
}


class HipKernelError(RuntimeError):
    pass


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """{function name: (return type, [(arg type, arg name), ...])} for every declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?:^|\n)\s*(int|const char\*)\s+(od_\w+)\s*\(([^;]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                typ = mm.group(1).strip().replace(" *", "*")
                parsed.append((typ, mm.group(2)))
        out[name] = (ret, parsed)
    return out


class _Lib:
    def __init__(self, path: str):
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.decls = parse_header()
        for name, (ret, args) in self.decls.items():
            fn = getattr(self.cdll, name)       # AttributeError if the symbol is not exported
            fn.restype = ctypes.c_char_p if ret != "int" else ctypes.c_int
            fn.argtypes = [_CTYPES[t] for t, _ in args]
        self._err = self.cdll.od_error_string

    def call(self, name: str, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self._err(rc)
            raise HipKernelError(f"{name} failed with code {rc}: {msg.decode() if msg else '?'}")

    def __getattr__(self, name):
        if name.startswith("od_"):
            return lambda *a: self.call(name, *a)
        raise AttributeError(name)


_lib: Optional[_Lib] = None


def lib() -> _Lib:
    """The loaded HIP library.  Raises if it has not been built — there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(DEFAULT_SO):
            raise RuntimeError(
                f"{DEFAULT_SO} not found: build it with osu_dreamer_amd/csrc/build.sh "
                "(python -c 'import __graft_entry__ as g; g.build()').  osu_dreamer_amd has no CPU fallback.")
        _lib = _Lib(DEFAULT_SO)
    return _lib


def use_library(path: str) -> _Lib:
    """Bind a specific shared object (tests: the emulator build).  Not used by the package."""
    global _lib
    _lib = _Lib(path)
    return _lib


def loaded_path() -> Optional[str]:
    return _lib.path if _lib is not None else None


def source_sha() -> str:
    """Hash of the kernel sources the loaded library was built from (od_build_source_sha)."""
    return lib().cdll.od_build_source_sha().decode()
