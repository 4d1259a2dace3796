"""ctypes binding of libsmx.so (include/smx.h).  No CPU fallback: if the library is missing the
import of the product path fails with a build hint."""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMX_LIB selects an experimental build of the same ABI (tuning only)
LIB_PATH = os.environ.get("SMX_LIB") or os.path.join(_HERE, "csrc", "libsmx.so")

SMX_PATH_DECIMATED = 1
SMX_PATH_DIRECT = 2
SMX_PATH_DECIM16 = 3


class SmxError(RuntimeError):
    pass


class smx_plan(ctypes.Structure):
    _fields_ = [("path", ctypes.c_int), ("k", ctypes.c_int), ("L", ctypes.c_int),
                ("bands", ctypes.c_int), ("nsplit", ctypes.c_int), ("workgroups", ctypes.c_int),
                ("groups", ctypes.c_int)]


class smx_options(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("nsplit", "placement", "round", "force_direct", "full8", "fourstep",
                                            "fs_bgroups", "fold_gradw", "decim16", "conv1", "st_plain")]


class smx_shape(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("rows", ctypes.c_int), ("D", ctypes.c_int), ("F", ctypes.c_int),
                ("n_fft", ctypes.c_int), ("k", ctypes.c_int)]


_lock = threading.Lock()
_lib = None

_P = ctypes.c_void_p
_I = ctypes.c_int
_LL = ctypes.c_longlong
_SZ = ctypes.c_size_t

_SIGS = {
    "smx_version": (ctypes.c_int, []),
    "smx_last_error": (ctypes.c_char_p, []),
    "smx_set_option": (_I, [ctypes.c_char_p, _I]),
    "smx_options_default": (_I, [_P]),
    "smx_options_push": (_I, [_P]),
    "smx_options_pop": (_I, []),
    "smx_options_epoch": (ctypes.c_ulonglong, []),
    "smx_tables_epoch": (ctypes.c_ulonglong, []),
    "smx_build_flags": (ctypes.c_char_p, []),
    "smx_diag_clock": (_I, [_P, _I, _P]),
    "smx_plan_query": (_I, [_I, _I, _I, _I, ctypes.POINTER(smx_plan)]),
    "smx_workspace_bytes": (_I, [_I, _I, _I, _I, ctypes.POINTER(_SZ)]),
    "smx_prepare": (_I, [_I]),
    "smx_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _P]),
    "smx_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _P]),
    "smx_spectrum": (_I, [_P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "smx_grad_w": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "smx_wfilter_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "smx_wfilter_grad_w": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "smx_cmul": (_I, [_P, _P, _P, _LL, _LL, _I, _P]),
    "smx_cmul_grad_w": (_I, [_P, _P, _P, _LL, _LL, _P]),
    "smx_rng_next": (_I, [_P, _P, _P]),
    "smx_forward_dropout": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, ctypes.c_float,
                                 _P, _P, _P]),
    "smx_backward_dropout": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I,
                                  ctypes.c_float, _P, _P, _P]),
    "smx_block_forward_dropout": (_I, [_P, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P, _SZ,
                                       _I, _I, _I, _I, ctypes.c_float, _P, _P, _P]),
    "smx_block_backward_dropout": (_I, [_P] * 14 + [_SZ, _I, _I, _I, _I, _I, ctypes.c_float, _P, _P, _P]),
    "smx_block_supported": (_I, [_I]),
    "smx_plan_query_ex": (_I, [ctypes.POINTER(smx_shape), ctypes.POINTER(smx_plan)]),
    "smx_workspace_bytes_ex": (_I, [ctypes.POINTER(smx_shape), ctypes.POINTER(_SZ)]),
    "smx_forward_ex": (_I, [ctypes.POINTER(smx_shape), _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _P, _P, _P]),
    "smx_backward_ex": (_I, [ctypes.POINTER(smx_shape), _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _P,
                             _P, _P, _P]),
    "smx_row_scale_supported": (_I, [ctypes.POINTER(smx_shape)]),
    "smx_cfft_ex": (_I, [ctypes.POINTER(smx_shape), _P, _P, _P, _SZ, _P]),
    "smx_rfft_ex": (_I, [ctypes.POINTER(smx_shape), _P, _P, ctypes.c_float, _I, _P, _SZ, _P]),
    "smx_irfft_ex": (_I, [ctypes.POINTER(smx_shape), _P, _P, ctypes.c_float, _I, _P, _SZ, _P]),
    "smx_cfft_workspace_bytes": (_I, [ctypes.POINTER(smx_shape), ctypes.POINTER(_SZ)]),
    "smx_conv_supported": (_I, [ctypes.POINTER(smx_shape)]),
    "smx_conv_workspace_bytes": (_I, [ctypes.POINTER(smx_shape), ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)]),
    "smx_conv_forward": (_I, [ctypes.POINTER(smx_shape), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "smx_conv_backward": (_I, [ctypes.POINTER(smx_shape), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "smx_phase_filter": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "smx_phase_filter_backward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "smx_conv_response": (_I, [_I, _I, _P, _P, _P, _P, _P, _P]),
    "smx_conv_response_backward": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "smx_spectrum_ex": (_I, [ctypes.POINTER(smx_shape), _P, _P, _P, _SZ, _P]),
    "smx_block_forward": (_I, [_P, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P, _SZ,
                               _I, _I, _I, _I, _P]),
    "smx_block_backward": (_I, [_P] * 14 + [_SZ, _I, _I, _I, _I, _I, _P]),
    "smx_dwconv3_workspace_bytes": (_I, [_I, _I, _I, ctypes.POINTER(_SZ)]),
    "smx_dwconv3_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "smx_dwconv3_backward": (_I, [_P] * 10 + [_SZ, _I, _I, _I, _P]),
    "smx_spectral_ln_supported": (_I, [_I]),
    "smx_spectral_ln_forward": (_I, [_P, _P, _P, ctypes.c_float, _P, _I, _I, _I, _I, _P]),
    "smx_spectral_ln_backward": (_I, [_P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _I, _I, _I, _I, _P]),
    "smx_planar_cmul_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "smx_planar_cmul_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "smx_planar_add": (_I, [_P, _P, _P, _LL, _P]),
    "smx_planar_split": (_I, [_P, _P, _LL, _P]),
    "smx_spectral_gate_workspace_bytes": (_I, [_I, _I, _I, ctypes.POINTER(_SZ)]),
    "smx_spectral_gate_forward": (_I, [_P] * 7 + [_I, _I, _I, _P]),
    "smx_spectral_gate_backward": (_I, [_P] * 12 + [_SZ, _I, _I, _I, _P]),
    "smx_mix_workspace_bytes": (_I, [ctypes.POINTER(_SZ)]),
    "smx_mix_forward": (_I, [_P] * 5 + [ctypes.c_float, _P, _LL, _P]),
    "smx_mix_backward": (_I, [_P] * 4 + [ctypes.c_float] + [_P] * 5 + [_SZ, _LL, _P]),
}


_SINCE = {"smx_diag_clock": 302, "smx_dwconv3_workspace_bytes": 302, "smx_dwconv3_forward": 302,
          "smx_dwconv3_backward": 302, "smx_spectral_ln_supported": 302, "smx_spectral_ln_forward": 302,
          "smx_spectral_ln_backward": 302, "smx_planar_cmul_forward": 302, "smx_planar_cmul_backward": 302,
          "smx_planar_add": 302, "smx_planar_split": 302, "smx_spectral_gate_workspace_bytes": 303,
          "smx_spectral_gate_forward": 303, "smx_spectral_gate_backward": 303, "smx_mix_workspace_bytes": 303,
          "smx_mix_forward": 303, "smx_mix_backward": 303}        # entry points younger than the oldest library the A/B tools still load


def load(path: str):
    """dlopen one build of the library and declare its signatures (tools/ab_inproc.py loads several)."""
    import torch  # noqa: F401  (maps torch's libamdhip64 before ours resolves it)
    h = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        if name in _SINCE and h.smx_version() < _SINCE[name]:
            continue                       # an older build loaded beside the current one (A/B tools only)
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    check_build_flags((h.smx_build_flags() or b"").decode(), path)
    return h


def check_build_flags(flags: str, path: str) -> None:
    """Refuse a library built with -DSMX_AB_* (csrc/build.sh with a mis-set SMX_EXTRA): same ABI, same
    smx_version(), garbage results."""
    if "SMX_AB_" in flags and os.environ.get("SMX_ALLOW_ABLATION") != "1":
        raise SmxError(f"{path} is a timing-ablation build ({flags.strip()}): it returns wrong results by design. "
                       f"Rebuild with csrc/build.sh (no SMX_EXTRA), or set SMX_ALLOW_ABLATION=1 for tools/ab.sh.")


def lib():
    """Load libsmx.so once.  torch must be imported first so the HIP runtime torch uses is the
    one this library binds to (same SONAME, already mapped)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SmxError(
                f"{LIB_PATH} not found: the HIP library is not built. Run "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or "
                f"`{os.path.join(_HERE, 'csrc', 'build.sh')}`. There is no CPU fallback.")
        _lib = load(LIB_PATH)
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().smx_last_error()
        raise SmxError(f"libsmx error {rc}: {msg.decode() if msg else '?'}")


def plan(B: int, N: int, D: int, F: int) -> smx_plan:
    p = smx_plan()
    check(lib().smx_plan_query(B, N, D, F, ctypes.byref(p)))
    return p


def workspace_bytes(B: int, N: int, D: int, F: int) -> int:
    s = _SZ()
    check(lib().smx_workspace_bytes(B, N, D, F, ctypes.byref(s)))
    return int(s.value)


def plan_ex(sh: smx_shape) -> smx_plan:
    p = smx_plan()
    check(lib().smx_plan_query_ex(ctypes.byref(sh), ctypes.byref(p)))
    return p


def workspace_bytes_ex(sh: smx_shape) -> int:
    s = _SZ()
    check(lib().smx_workspace_bytes_ex(ctypes.byref(sh), ctypes.byref(s)))
    return int(s.value)


def set_option(name: str, value: int) -> None:
    """Process-wide default of a plan knob (include/smx.h).  Nothing needs clearing on the Python side: every
    per-shape memo in functional.py is keyed by opts_key(), which changes with the library's options epoch."""
    check(lib().smx_set_option(name.encode(), int(value)))


_tls = threading.local()


def opts_key() -> tuple:
    """What the plan of a shape depends on besides the shape: the library's process-wide options epoch and this
    thread's scoped overrides (options())."""
    return (int(lib().smx_options_epoch()), getattr(_tls, "stack", ()))


def current_options():
    """The innermost options this thread has pushed, as a dict (None: the process-wide defaults apply)."""
    st = getattr(_tls, "stack", ())
    return dict(zip((n for n, _ in smx_options._fields_), st[-1])) if st else None


def effective_options() -> dict:
    """The options in force for this thread right now, ALWAYS a full dict: the innermost scoped push, or a snapshot of
    the process-wide defaults.  What an autograd node notes in forward and re-applies around its backward, so that
    neither the autograd thread's empty scope nor a set_option() between the two halves can make them plan
    differently (another workspace layout, another save layout of the one-launch convolution)."""
    cur = current_options()
    if cur is not None:
        return cur
    o = smx_options()
    check(lib().smx_options_default(ctypes.byref(o)))
    return {n: int(getattr(o, n)) for n, _ in smx_options._fields_}


class options:
    """`with _lib.options(nsplit=2, fourstep=0): ...` -- the calls made by THIS thread inside the block plan with
    these knobs instead of the process-wide defaults (smx_options_push / smx_options_pop): plan choice as an
    argument of the calling context, so two users of one process can run different plans."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        o = smx_options()
        check(lib().smx_options_default(ctypes.byref(o)))
        for k, v in self.kw.items():
            if not hasattr(o, k):
                raise ValueError(f"unknown plan option {k!r}")
            setattr(o, k, int(v))
        check(lib().smx_options_push(ctypes.byref(o)))
        vals = tuple(getattr(o, n) for n, _ in smx_options._fields_)
        _tls.stack = getattr(_tls, "stack", ()) + (vals,)
        return self

    def __exit__(self, *a):
        _tls.stack = _tls.stack[:-1]
        check(lib().smx_options_pop())
