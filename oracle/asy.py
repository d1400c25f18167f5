"""oracle/asy.py -- TEST INFRASTRUCTURE ONLY.  ctypes front end of oracle/asy_ref.c, with the tensor signatures of the
reference's `asy_tools` extension (src/dagr/asynchronous/asy_tools/main.cu:69-236).  PARITY UNPINNED w.r.t. reference
outputs (see asy_ref.c).  Only tests/ may import this."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libasy_ref.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "asy_ref.c")
        if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def masked_isdiff(indices, x_old, x_new, atol, rtol):
    lib().ora_masked_isdiff(_p(indices), _p(x_old), _p(x_new), indices.shape[0], x_old.shape[1], ctypes.c_float(atol), ctypes.c_float(rtol))
    return indices[indices > -1]


def masked_inplace_BN(indices, x, x_out, running_mean, running_var, weight, bias, eps):
    lib().ora_masked_inplace_bn(_p(indices), _p(x), _p(x_out), _p(running_mean), _p(running_var), _p(weight), _p(bias),
                                indices.shape[0], x.shape[1], ctypes.c_float(eps))


def masked_lin(indices, x_in, x_out, weight, bias, add):
    lib().ora_masked_lin(_p(indices), _p(x_in), _p(x_out), _p(weight), _p(bias), indices.shape[0], weight.shape[1], weight.shape[0], int(bool(add)))


def masked_lin_no_bias(indices, x_in, x_out, weight, add):
    masked_lin(indices, x_in, x_out, weight, None, add)
