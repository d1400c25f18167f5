"""ctypes loader for libeventad_hip.so.  Prototypes are read from include/eventad_hip.h so the
Python side can never drift from the C ABI.  There is NO fallback: if the library is missing
or a symbol is absent, importing/using the native path raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "eventad_hip.h")
SO_PATH = os.path.join(_HERE, "lib", "libeventad_hip.so")

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "double": ctypes.c_double}


def parse_header(path=HEADER):
    """-> {name: [ctypes arg types]} for every `int eventad_*(...)` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(eventad_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if "*" in a:
                args.append(ctypes.c_void_p)
            else:
                args.append(_CTYPES[a.replace("const ", "").rsplit(" ", 1)[0].strip()])
        protos[m.group(1)] = args
    return protos


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "eventad_amd: native library %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C eventad_amd/csrc`.  There is no CPU/PyTorch fallback for the hot path." % SO_PATH)
        lib = ctypes.CDLL(SO_PATH)
        for name, argtypes in parse_header().items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


class NativeError(RuntimeError):
    pass


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise NativeError("%s failed with code %d%s" % (name, rc, " (bad argument)" if rc == -1 else " (hipError_t)"))


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "native kernels need contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_handle():
    """Raw handle of torch's current stream on the current device -- the value of torch.cuda.current_stream().cuda_stream
    without building the Stream object (about 7 us per call: a tenth of the host time of an eager forward pass of a few
    hundred launches)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())

