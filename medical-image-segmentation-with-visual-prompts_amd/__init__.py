"""mivp_amd -- MI355X-native (gfx950) Swin-UNETR hot path.

Host side (this package): the reference's ``SwinUnetR(conf)`` nn.Module surface
and the data-parallel training-step harness, in Python on PyTorch-ROCm (device
memory, streams, torch.distributed only).  Device side: hand-written HIP kernels
behind the C ABI of ``include/mivp.h`` (``libmivp_hip.so``, built in-tree by
``build.py``).  No CPU fallback: ops raise if the library is missing or the
tensors are not on the GPU.
"""
from . import _lib, geometry  # noqa: F401

__all__ = ["_lib", "geometry"]


def __getattr__(name):
    # heavy modules are imported lazily so that `import mivp_amd` stays cheap
    if name in ("swin_ops", "ops", "swin_unetr", "train"):
        import importlib
        return importlib.import_module(f"mivp_amd.{name}")
    if name == "SwinUnetR":
        from .swin_unetr import SwinUnetR
        return SwinUnetR
    raise AttributeError(name)
