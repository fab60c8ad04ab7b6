"""Device-to-device copy out of engine-owned memory into a torch tensor (hipMemcpyAsync on
torch's current stream), used by the env state views."""
import ctypes as C

import torch

_hip = None


def _rt():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipMemcpyAsync.restype = C.c_int
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    return _hip


def copy_d2d(dst_ptr, src_ptr, nbytes, device):
    stream = torch.cuda.current_stream(device).cuda_stream
    rc = _rt().hipMemcpyAsync(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes, 3, C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"hipMemcpyAsync failed with {rc}")
