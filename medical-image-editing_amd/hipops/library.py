"""PyTorch operator registration of the C ABI: every kernel entry point of include/vqwnet_hip.h is a dispatcher
operator `torch.ops.vqw.<name>` with a schema string derived from its C prototype.

    const T* p      ->  Tensor? p            (read)
    T* p            ->  Tensor(a!)? p        (written: outputs, in-place buffer updates, workspaces)
    int / long / size_t / int64_t  -> int,   float / double -> float
    void* stream    ->  dropped: the kernel is enqueued on torch's CURRENT stream
    int status      ->  () ; a non-zero status raises RuntimeError with vqw_last_error()

so the mutation annotations the reference boundary asks for (`vq_ema_update_(Tensor(a!) embed, ...)`) come
from the `const` qualifiers of the header and cannot drift from it.  The operators are out-variant (the caller
allocates results through the caching allocator), have a CUDA (= ROCm) kernel only - a CPU tensor reaches no kernel
and raises, there is no fallback - and a no-op fake kernel, so FakeTensor / torch.compile tracing sees shapes and side
effects without touching the GPU.  Host-side queries (`*_ws_bytes`, `*_supported`, `*_parts`, ...) take no tensors
and stay plain C calls.

The differentiable operators of hipops.ops (torch.autograd.Function over these kernels) are what the nn.Module
classes call; `functional.py` additionally exposes the main ones as functional dispatcher operators with
torch.library.register_autograd formulas.
"""
import ctypes
import os
import re

import torch

from . import _lib

NAMESPACE = "vqw"
_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "include", "vqwnet_hip.h")

_INT = {"int", "long", "size_t", "int64_t", "int32_t", "unsigned"}
_FLT = {"float", "double"}


def parse_header(path=_HDR):
    """-> {name: (return type, [(kind, name)])} with kind in {'in', 'out', 'int', 'float', 'stream', 'host'}."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(vqw_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                pname = re.findall(r"\w+", p)[-1]
                if "*" in p:
                    if pname == "stream":
                        kind = "stream"
                    elif name.endswith("_host"):
                        kind = "host"                 # host arrays passed by value (vqw_weighted_sum_host)
                    else:
                        kind = "in" if p.startswith("const ") else "out"
                else:
                    base = p.replace("const ", "").split()[0]
                    kind = "int" if base in _INT else "float" if base in _FLT else None
                    if kind is None:
                        raise RuntimeError("unhandled parameter %r of %s" % (p, name))
                args.append((kind, pname))
        protos[name] = (ret, args)
    return protos


def schema_of(name, args):
    """Schema string of a kernel entry point (None when it is not expressible as a tensor operator)."""
    if not args or args[-1][0] != "stream" or any(k == "host" for k, _ in args):
        return None
    parts, letter = [], 0
    for kind, pname in args[:-1]:
        if kind == "in":
            parts.append("Tensor? %s" % pname)
        elif kind == "out":
            parts.append("Tensor(%s!)? %s" % (chr(ord("a") + letter), pname))
            letter += 1
        elif kind == "int":
            parts.append("int %s" % pname)
        else:
            parts.append("float %s" % pname)
    return "%s(%s) -> ()" % (name[len("vqw_"):], ", ".join(parts))


_lib_def = None
_ops = {}
SCHEMAS = {}


def _make_kernel(name, args):
    cfn = getattr(_lib.load(), name)
    kinds = [k for k, _ in args[:-1]]
    c_void_p = ctypes.c_void_p

    def kernel(*a):
        ca = [(c_void_p(v.data_ptr()) if v is not None else None) if k in ("in", "out") else v for k, v in zip(kinds, a)]
        ca.append(c_void_p(torch.cuda.current_stream().cuda_stream))
        rc = cfn(*ca)
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (name, rc, (_lib.load().vqw_last_error() or b"").decode()))
    return kernel


def register():
    """Define the namespace once per process (idempotent).  Raises if libvqwnet_hip.so is not built."""
    global _lib_def
    if _lib_def is not None:
        return _ops
    _lib.load()
    lib = torch.library.Library(NAMESPACE, "DEF")
    for name, (ret, args) in parse_header().items():
        sch = schema_of(name, args)
        if sch is None:
            continue
        short = name[len("vqw_"):]
        lib.define(sch)
        lib.impl(short, _make_kernel(name, args), "CUDA")
        torch.library.register_fake("%s::%s" % (NAMESPACE, short), lambda *a: None, lib=lib)
        SCHEMAS[short] = sch
        _ops[name] = getattr(getattr(torch.ops, NAMESPACE), short).default
    _lib_def = lib
    return _ops


class Dispatch:
    """`L.vqw_xxx(tensor or None, ..., scalars ..., STREAM)` with the argument list of the C function: kernels go through
    torch.ops.vqw.*, tensor-free host queries straight to the C library.  Returns the C convention's 0 (errors raise)."""
    STREAM = object()

    def __init__(self):
        self._ops = register()
        self._c = _lib.load()
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            op = self._ops.get(name)
            if op is None:
                fn = getattr(self._c, name)
            else:
                def fn(*a, _op=op):
                    _op(*a[:-1])          # the trailing stream placeholder is implied by the current stream
                    return 0
            self._cache[name] = fn
        return fn
