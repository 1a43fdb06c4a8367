"""ctypes binding of the C-ABI kernel library (include/svpc_hip.h ↔ svpc_amd/csrc/libsvpc_hip.so).

The header is the single source of truth: argument types are derived from its declarations.  There is no CPU
fallback — if the library (or a symbol) is missing the first kernel call raises, loudly.
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVPC_LIB", os.path.join(_HERE, "csrc", "libsvpc_hip.so"))   # override: kernel experiments only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "svpc_hip.h")

_lib = None
_decls = None


class SvpcKernelError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """→ {name: (restype, [ctypes arg types])} for every ``svpc_*`` function declared in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(svpc_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "svpc_stream_t" in a:
                    types.append(ctypes.c_void_p)
                elif a.startswith("size_t"):
                    types.append(ctypes.c_size_t)
                elif a.startswith("float"):
                    types.append(ctypes.c_float)
                elif a.startswith("double"):
                    types.append(ctypes.c_double)
                elif a.startswith("unsigned"):
                    types.append(ctypes.c_uint)
                elif a.startswith("long long"):
                    types.append(ctypes.c_longlong)
                elif a.startswith("int"):
                    types.append(ctypes.c_int)
                else:
                    raise ValueError("unparsed argument %r in %s" % (a, name))
        decls[name] = (ctypes.c_char_p if "char" in ret else ctypes.c_int, types)
    return decls


def declarations():
    global _decls
    if _decls is None:
        _decls = parse_header()
    return _decls


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvpcKernelError(
            "svpc_amd: HIP kernel library not built (%s missing). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C svpc_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, args) in declarations().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise SvpcKernelError("svpc_amd: %s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = ret
        fn.argtypes = args
    _lib = lib
    return lib


_FN = {}


def call(name, *args):
    """Invoke ``svpc_<name>``; raise with the library's error string on a non-zero status."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), "svpc_" + name)
    rc = fn(*args)
    if rc != 0:
        lib = load()
        msg = lib.svpc_last_error()
        raise SvpcKernelError("svpc_%s failed (%d): %s" % (name, rc, msg.decode() if msg else "?"))
    return rc
