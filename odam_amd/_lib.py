"""ctypes loader for the in-tree HIP library (odam_amd/libodam_amd.so).

The product path has no CPU fallback: if the library is missing or an entry point returns a
non-zero code, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODAM_AMD_LIB") or os.path.join(_HERE, "libodam_amd.so")   # the override is for A/B builds of the library
_lib = None

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_void_p = ctypes.c_void_p


class OdamError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OdamError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch ships its own HIP runtime; it must be the first one mapped into the process.  Loading this library
        # before torch binds it to the system libamdhip64 instead, and the two runtimes then disagree about the
        # devices ("no ROCm-capable device is detected" at the first hipMalloc).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.odam_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().odam_last_error().decode("utf-8", "replace")
        raise OdamError(f"{what} failed with code {code}: {msg}")


def ptr(t):
    """device/host pointer of a torch tensor (or None) as a c_void_p"""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())
