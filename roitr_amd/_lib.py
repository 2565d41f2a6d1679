"""ctypes binding of libroitr_hip.so (the C-ABI boundary, include/*.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing or fails to load,
importing an operator raises.  (The CPU restatement lives under oracle/ and is test-only.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libroitr_hip.so")
_lib = None


class RoitrError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RoitrError(
                f"{LIB_PATH} is missing: build it with `python -m roitr_amd.build` "
                "(hipcc --offload-arch=gfx950).  roitr_amd has no CPU fallback.")
        # torch first: the library must bind to the HIP runtime torch brings along.  Loaded before torch, it pulls in the system
        # libamdhip64 and the process ends up with two runtimes -- the engine then sees "no ROCm-capable device" (build() followed by
        # smoke() in one process did exactly that).  torch is the package's device-memory / stream plumbing anyway.
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.roitr_last_error.restype = ctypes.c_char_p
        _lib.roitr_knn_workspace_bytes.restype = ctypes.c_size_t
        _lib.roitr_geo_table_floats.restype = ctypes.c_size_t
        for name in ("roitr_engine_create",):
            if hasattr(_lib, name):
                getattr(_lib, name).restype = ctypes.c_void_p
    return _lib


def check(status, what=""):
    if status != 0:
        raise RoitrError(f"{what} failed with status {status}: {lib().roitr_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL).  A host tensor is an error, not a fallback: every entry point of the
    library dereferences its pointers on the device."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RoitrError("roitr_amd needs ROCm device tensors (no CPU fallback): got a tensor on " + str(t.device))
    return ctypes.c_void_p(t.data_ptr())


def host_ptr(t):
    """Pointer of a HOST tensor, for the few host-side entry points (roitr_geo_table_build)."""
    if t.is_cuda:
        raise RoitrError("host_ptr: expected a CPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def c_float(x):
    """A by-value float argument (ctypes would otherwise pass a Python float as a double)."""
    return ctypes.c_float(float(x))
