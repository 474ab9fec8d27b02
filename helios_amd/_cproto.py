"""Minimal C-prototype reader used to bind plain C-ABI functions with ctypes.

The C-ABI of this project uses only `double`, `int`, `int32_t`, `int64_t`, `uint64_t`, `size_t`,
`void*` / `const void*`, `double*` / `const double*`, `int*` / `const int*`, `char*` -- no structs
cross the boundary -- so a small regular expression is enough to turn the declarations of a header
into ctypes signatures.  Keeping the header as the single source of truth means a test can check
that the shared library exports every symbol the header declares (tests/test_abi.py).
"""
import ctypes
import re

_SCALARS = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "size_t": ctypes.c_size_t,
    "double": ctypes.c_double,
    "float": ctypes.c_float,
    "void": None,
}

_PROTO = re.compile(
    r"(?:^|[;{}\n])\s*(?:extern\s+)?(?:HX_API\s+)?"
    r"(void\s*\*|const\s+char\s*\*|void|int|int32_t|int64_t|uint64_t|double)\s*"
    r"([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{}()]*)\)\s*[;{]",
    re.S,
)


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)
    return text


def _ctype_of(decl):
    """ctypes type of one parameter declaration such as `const double* temp`."""
    d = decl.strip()
    if "*" in d:
        base = d.replace("const", " ").split("*")[0].strip()
        if d.count("*") > 1:
            return ctypes.c_void_p  # pointer-to-pointer: out-parameters for handles
        if base == "double":
            return ctypes.POINTER(ctypes.c_double)
        if base in ("int", "int32_t"):
            return ctypes.POINTER(ctypes.c_int32)
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    toks = d.replace("const", " ").replace("unsigned", " ").split()
    if not toks:
        raise ValueError("cannot parse parameter %r" % decl)
    return _SCALARS[toks[0]]


def parse_prototypes(text, prefix):
    """Return {name: (restype, [argtypes], [argnames])} for every function whose name starts with
    `prefix` declared (or defined) in the C text."""
    out = {}
    for m in _PROTO.finditer(_strip_comments(text)):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        if not name.startswith(prefix):
            continue
        params = params.strip()
        argtypes, argnames = [], []
        if params and params != "void":
            for p in params.split(","):
                argtypes.append(_ctype_of(p))
                argnames.append(re.split(r"[\s*]+", p.strip())[-1])
        ret = ret.replace(" ", "")
        if ret.startswith("constchar"):
            restype = ctypes.c_char_p
        elif ret == "void*":
            restype = ctypes.c_void_p
        else:
            restype = _SCALARS[ret]
        out[name] = (restype, argtypes, argnames)
    return out


def bind(lib, protos):
    """Apply parsed prototypes to a loaded ctypes library; raises AttributeError on a missing symbol."""
    for name, (restype, argtypes, _names) in protos.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
