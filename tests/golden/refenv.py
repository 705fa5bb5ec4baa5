"""Import harness for the upstream reference (likojack/ODAM) -- golden-vector generation ONLY.

This file is used only inside the build container, where /root/reference exists.
Nothing under tests/ that runs on the GPU box imports it.  It
  * builds the reference's Cython sampler binding from a scratch copy of
    /root/reference/src/super_quadric (the reference tree itself is read-only),
  * registers empty stand-in modules for third-party packages the reference imports
    at module load but never calls on the detect / SQ-fit path (easydict, cv2, open3d,
    trimesh, quaternion, plyfile, torchvision),
  * chdirs to /root/reference because sq_libs.py:388 opens the scale prior by relative path.
"""
import os
import subprocess
import sys
import types
import shutil

REF = "/root/reference"
SCRATCH = os.environ.get("ODAM_REF_SCRATCH", "/tmp/odam_ref_scratch")


def _build_sampler():
    dst = os.path.join(SCRATCH, "sq")
    so_dir = os.path.join(dst, "learnable_primitives", "fast_sampler")
    if os.path.isdir(so_dir) and any(f.endswith(".so") for f in os.listdir(so_dir)):
        return dst
    os.makedirs(SCRATCH, exist_ok=True)
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    shutil.copytree(os.path.join(REF, "src", "super_quadric"), dst)
    gen = os.path.join(so_dir, "_sampler.c")
    if os.path.exists(gen):
        os.remove(gen)  # re-cythonise with the container's Cython
    subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=dst,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return dst


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setattr__(k, v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    if "easydict" not in sys.modules:
        _stub("easydict", EasyDict=_EasyDict)
    for n in ("cv2", "open3d", "trimesh", "quaternion"):
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                _stub(n)
    if "plyfile" not in sys.modules:
        _stub("plyfile", PlyData=object, PlyElement=object)
    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except Exception:
            import torch
            tv = _stub("torchvision", __version__="0.11.2")
            ops = _stub("torchvision.ops")
            boxes = _stub("torchvision.ops.boxes",
                          box_area=lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
            misc = _stub("torchvision.ops.misc", interpolate=torch.nn.functional.interpolate)
            ops.boxes, ops.misc = boxes, misc
            tr = _stub("torchvision.transforms")
            trf = _stub("torchvision.transforms.functional")
            tr.functional = trf
            tr.RandomCrop = object
            models = _stub("torchvision.models")
            mu = _stub("torchvision.models._utils")

            class IntermediateLayerGetter(torch.nn.ModuleDict):
                def __init__(self, model, return_layers):
                    layers = {}
                    rl = dict(return_layers)
                    for name, module in model.named_children():
                        layers[name] = module
                        if name in rl:
                            del rl[name]
                        if not rl:
                            break
                    super().__init__(layers)
                    self.return_layers = dict(return_layers)

                def forward(self, x):
                    out = {}
                    for name, module in self.items():
                        x = module(x)
                        if name in self.return_layers:
                            out[self.return_layers[name]] = x
                    return out

            mu.IntermediateLayerGetter = IntermediateLayerGetter
            models._utils = mu
            ds = _stub("torchvision.datasets", CocoDetection=object)
            tv.ops, tv.transforms, tv.models, tv.datasets = ops, tr, models, ds


def setup():
    """Make `import src.processor` etc. work; returns the scratch super_quadric path."""
    sq = _build_sampler()
    _install_stubs()
    for p in (sq, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.chdir(REF)
    return sq
