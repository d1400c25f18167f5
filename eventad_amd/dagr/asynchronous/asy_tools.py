"""`asy_tools` with the reference's four pybind signatures (src/dagr/asynchronous/asy_tools/main.cu:239-244) over the
C ABI (csrc/asy.hip).  Tensors are CUDA, contiguous, fp32 / int64 as in the reference; launches go on the current
stream.  `eventad_amd.compat.install()` registers this module as `asy_tools`."""
import torch

from ..._lib import call, ptr


def _check(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("asy_tools: tensors must be CUDA tensors")
        if not t.is_contiguous():
            raise RuntimeError("asy_tools: tensors must be contiguous")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def masked_isdiff(indices, x_old, x_new, atol, rtol):
    """-> the entries of `indices` whose rows differ (main.cu:96-125); `indices` itself is overwritten with -1 marks."""
    _check(indices, x_old, x_new)
    call("eventad_masked_isdiff", ptr(indices), ptr(x_old), ptr(x_new), indices.shape[0], x_old.shape[1], float(atol), float(rtol),
         _stream())
    return indices[indices > -1]


def masked_inplace_BN(indices, x, x_out, running_mean, running_var, weight, bias, eps):
    _check(indices, x, x_out, running_mean, running_var, weight, bias)
    call("eventad_masked_inplace_bn", ptr(indices), ptr(x), ptr(x_out), ptr(running_mean), ptr(running_var), ptr(weight),
         ptr(bias), indices.shape[0], x.shape[1], float(eps), _stream())


def masked_lin(indices, x_in, x_out, weight, bias, add):
    _check(indices, x_in, x_out, weight, bias)
    call("eventad_masked_lin", ptr(indices), ptr(x_in), ptr(x_out), ptr(weight), ptr(bias), indices.shape[0], weight.shape[1],
         weight.shape[0], int(bool(add)), _stream())


def masked_lin_no_bias(indices, x_in, x_out, weight, add):
    _check(indices, x_in, x_out, weight)
    call("eventad_masked_lin", ptr(indices), ptr(x_in), ptr(x_out), ptr(weight), None, indices.shape[0], weight.shape[1],
         weight.shape[0], int(bool(add)), _stream())
