"""ctypes binding of libmivp_hip.so (the C ABI declared in include/mivp.h).

The product path has NO fallback: if the library is missing or a call fails,
a RuntimeError is raised.  Tensors are passed as raw device pointers
(``tensor.data_ptr()``) plus the current torch HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmivp_hip.so")
ABI_VERSION = 12

i32, f32, vp, i64 = C.c_int32, C.c_float, C.c_void_p, C.c_int64


class SwinDesc(C.Structure):
    _fields_ = [(n, i32) for n in ("B", "C", "heads", "vol_in", "vol_out", "P", "Nq", "Nqp", "Np", "Npp", "Nkp",
                                   "aug", "augp", "has_mask")] + [("win", i32 * 3), ("q_scale", f32), ("ln_eps", f32),
                                                                  ("attn_drop_thr", C.c_uint32), ("attn_drop_scale", f32),
                                                                  ("attn_seed", C.c_uint32), ("proj_drop_thr", C.c_uint32),
                                                                  ("proj_drop_scale", f32), ("proj_seed", C.c_uint32),
                                                                  ("seed_epoch", C.c_void_p)]


class MergeDesc(C.Structure):
    _fields_ = [("B", i32), ("C", i32), ("dims", i32 * 3), ("odims", i32 * 3), ("merge_last", i32), ("Cout", i32),
                ("ln_eps", f32)]


class ConvDesc(C.Structure):
    _fields_ = [("B", i32), ("dims", i32 * 3), ("Cin", i32), ("Cout", i32), ("Kp", i32), ("pro_affine", i32),
                ("pro_lrelu", i32), ("add_residual", i32), ("out_f32", i32)]


class EmbedDesc(C.Structure):
    _fields_ = [("B", i32), ("Cin", i32), ("dims", i32 * 3), ("C", i32), ("nblk", i32)]


class UpcatDesc(C.Structure):
    _fields_ = [("B", i32), ("idims", i32 * 3), ("odims", i32 * 3), ("scale", i32 * 3), ("Cx", i32), ("Cs", i32),
                ("align_corners", i32)]


class OperandDesc(C.Structure):
    _fields_ = [("mode", i32), ("ld", i32), ("rows", i32), ("hd", i32), ("dims", i32 * 3), ("cin", i32)]


class GemmTnDesc(C.Structure):
    _fields_ = [("T", i64), ("M", i32), ("N", i32), ("a", OperandDesc), ("b", OperandDesc), ("alpha", f32),
                ("accumulate", i32), ("perm_cin", i32)]


_lib = None


def lib():
    """Load the shared library once; fail loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"mivp_amd: {LIB_PATH} is missing. Build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the hot path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.mivp_last_error.restype = C.c_char_p
        _lib.mivp_conv3d_wgrad_small_ws.restype = C.c_size_t
        _lib.mivp_conv3d_wgrad_rows_ws.restype = C.c_size_t
        _lib.mivp_conv3d_fwd_ws.restype = C.c_size_t
        _lib.mivp_dice_focal_ws.restype = C.c_size_t
        _lib.mivp_head_conv_ws.restype = C.c_size_t
        _lib.mivp_gemm_tn_ws.restype = C.c_size_t
        _lib.mivp_uphead_fwd_ws.restype = C.c_size_t
        ver = _lib.mivp_abi_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"mivp_amd: ABI version mismatch: library {ver}, binding {ABI_VERSION}")
    return _lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as a C pointer (every kernel is launched on it).  The raw getter is
    ~10x cheaper than building a torch.cuda.Stream object per launch (80+ launches per step)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous."""
    if t is None:
        return C.c_void_p(0)
    if not t.is_contiguous():
        raise RuntimeError("mivp_amd: non-contiguous tensor passed to the C ABI")
    if not t.is_cuda:
        raise RuntimeError("mivp_amd: the HIP kernels need device tensors (no CPU fallback; the CPU oracle "
                           "lives in oracle/ and is test infrastructure only)")
    return C.c_void_p(t.data_ptr())


def call(name, *args):
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"mivp_amd: {name} failed with code {rc}: {lib().mivp_last_error().decode()}")


# ---------------------------------------------------------------------------------------------
# optional per-kernel timing (bench.py): HIP events on the stream the kernel is launched on
# ---------------------------------------------------------------------------------------------
_prof = {"on": False, "sel": {}}          # key -> {"names": (...), "pred": callable | None, "events": [], "desc": ..., "entry": ...}


def profile_select(name, pred=None, key="roofline"):
    """Time every call of C-ABI entry ``name`` (a name or a tuple of names) whose ctypes args satisfy ``pred`` while
    profiling is on; several selections can be active under different ``key``s."""
    _prof["sel"][key] = {"names": (name,) if isinstance(name, str) else tuple(name), "pred": pred, "events": [],
                         "desc": None, "entry": None}


def profile_reset(on: bool):
    if on:
        for s in _prof["sel"].values():
            s["events"] = []
    _prof["on"] = on


def profile_result(key="roofline"):
    """(mean launch duration in ms, launches, descriptor of the last timed launch); ``profile_entry(key)`` names it."""
    s = _prof["sel"].get(key)
    if not s or not s["events"]:
        return 0.0, 0, None
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in s["events"]]
    return sum(ms) / len(ms), len(ms), s["desc"]


def profile_entry(key="roofline"):
    s = _prof["sel"].get(key)
    return s["entry"] if s else None


_plain_call = call


def call(name, *args):  # noqa: F811  (wraps the plain call with the optional event pair)
    if _prof["on"]:
        for s in _prof["sel"].values():
            if name in s["names"] and (s["pred"] is None or s["pred"](args)):
                a = torch.cuda.Event(enable_timing=True)
                b = torch.cuda.Event(enable_timing=True)
                a.record()
                _plain_call(name, *args)
                b.record()
                s["events"].append((a, b))
                d = args[0]._obj
                s["desc"] = type(d).from_buffer_copy(d)
                s["entry"] = name
                return
    _plain_call(name, *args)
