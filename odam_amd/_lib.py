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


# Switches of the HOST program (experiments, the nested test harness): "ODAM_CONFIG=cg.ring=2,cg.f32=0", or the variable
# names of rounds 1-2, are translated here into odam_config_set calls when the library is loaded.  The library itself never
# reads the environment; config() returns what is in force so that a program can log it (bench.py does).
_LEGACY_ENV = {"ODAM_CG_BIG": ("cg.ring", False), "ODAM_CG_BIG_F32": ("cg.f32", False), "ODAM_CG_FUSE": ("cg.fuse", False),
               "ODAM_CG_FUSE_BF16": ("cg.fuse_bf16", False), "ODAM_CG_S1": ("cg.s1", False), "ODAM_CG_NO_UT": ("cg.ut", True),
               "ODAM_CG_VARIANT": ("cg.tiles", False), "ODAM_CG_FORCE": ("cg.force", False), "ODAM_CG_NO_PRESPLIT": ("cg.presplit", True),
               "ODAM_ATT_X3": ("att.x3", False), "ODAM_ATT_FP32": ("att.bf16_mfma", True), "ODAM_STEM_ROWS": ("stem.rows", False),
               "ODAM_ASSOC_PERSIST": ("assoc.persist", False), "ODAM_SQ_SPLIT": ("sq.split", False),
               "ODAM_SQ_SPLIT_WAIT": ("sq.split_wait", False), "ODAM_SQ_S": ("sq.lanes", False)}


def _apply_env(L):
    pairs = []
    for name, (key, presence_means_off) in _LEGACY_ENV.items():
        if name in os.environ:
            pairs.append((key, 0 if presence_means_off else int(os.environ[name])))
    for item in filter(None, os.environ.get("ODAM_CONFIG", "").split(",")):
        k, v = item.split("=")
        pairs.append((k.strip(), int(v)))
    for k, v in pairs:
        if L.odam_config_set(k.encode(), ctypes.c_int(v)) != 0:
            raise OdamError(L.odam_last_error().decode("utf-8", "replace"))


def config():
    """the library's switch table as {key: value} (odam_config.h)"""
    buf = ctypes.create_string_buffer(1024)
    check(lib().odam_config_describe(buf, 1024), "odam_config_describe")
    return {k: int(v.rstrip("*")) for k, v in (kv.split("=") for kv in buf.value.decode().split())}


def get_config(key):
    """one switch of the table (a single cheap call: used per frame by the associator)"""
    v = ctypes.c_int(0)
    check(lib().odam_config_get(key.encode(), ctypes.byref(v)), f"odam_config_get({key})")
    return v.value


def set_config(key, value):
    check(lib().odam_config_set(key.encode(), ctypes.c_int(int(value))), f"odam_config_set({key})")


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
        _apply_env(L)
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
