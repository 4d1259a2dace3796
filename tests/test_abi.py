"""C ABI surface (include/smx.h) checked without a GPU: exports, plan logic, argument validation."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "include", "smx.h")


@pytest.fixture(scope="module")
def L():
    """The ctypes binding, with libsmx.so built first if this is a fresh checkout (hipcc cross-compiles
    gfx950 without a GPU; the product itself never builds implicitly -- it fails loudly instead)."""
    import subprocess
    from tensor_cuda_fft_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["bash", os.path.join(ROOT, "tensor-cuda-fft-_amd", "csrc", "build.sh")], check=True,
                       capture_output=True)
    return _lib


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smx_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_documented_entry_points():
    names = declared_functions()
    for must in ("smx_forward", "smx_backward", "smx_spectrum", "smx_grad_w", "smx_workspace_bytes",
                 "smx_plan_query", "smx_last_error", "smx_version", "smx_wfilter_forward",
                 "smx_wfilter_grad_w", "smx_cmul", "smx_cmul_grad_w", "smx_prepare", "smx_set_option",
                 "smx_block_supported", "smx_block_forward", "smx_block_backward"):
        assert must in names


def test_library_exports_every_declared_symbol(L):
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in smx.h but not exported by libsmx.so"
    assert L.lib().smx_version() == 303
    assert set(L._SIGS) == set(declared_functions())


def test_header_cites_the_reference_lines_it_replaces():
    src = open(HDR).read()
    for cite in ("spectral_layers.py:88-116", "wirtinger_ops.py:170-203", "wirtinger_ops.py:45-50"):
        assert cite in src


def test_plan_selection(L):
    p = L.plan(64, 4096, 256, 128)                 # C2: one fused launch per direction
    assert (p.path, p.k, p.L, p.bands, p.nsplit, p.workgroups) == (1, 128, 16, 1, 1, 512)
    p = L.plan(8, 65536, 256, 128)                 # C3: too few (b, d-tile) pairs -> residue split
    assert p.path == 1 and p.L == 256 and p.nsplit == 8 and p.workgroups == 512
    p = L.plan(64, 4096, 512, 256)                 # C5: k = 256 -> two bands
    assert p.path == 1 and p.bands == 2 and p.k == 256
    assert L.plan(8, 512, 256, 128).path == 1      # C1
    assert L.plan(2, 100, 8, 4).path == 2          # N % 256 != 0
    assert L.plan(2, 512, 7, 4).path == 2          # odd D
    p = L.plan(2, 4096, 8, 300)                    # 256 < k <= 512 -> four bands
    assert p.path == 1 and p.bands == 4 and p.k == 300
    p = L.plan(2, 8704, 8, 600)                    # k > 512, L = 34 -> band groups of the four-band kernels
    assert (p.path, p.bands, p.groups, p.nsplit) == (1, 4, 2, 1)
    assert L.plan(2, 8704, 8, 1536).groups == 3 and L.plan(64, 4096, 256, 128).groups == 1
    p = L.plan(2, 4352, 8, 600)                    # odd tile counts 17 ... 31: four-step path (round 3)
    assert (p.path, p.bands, p.groups) == (1, 0, 1)
    p = L.plan(2, 12288, 8, 700)                   # L = 48 = 12 x 4: two-level columns (round 3)
    assert (p.path, p.bands, p.groups) == (1, 0, 1)
    p = L.plan(2, 6144, 8, 600)                    # even L up to 32: four-step path
    assert (p.path, p.bands, p.groups) == (1, 0, 1)
    p = L.plan(2, 4096, 8, 600)                    # k > 512 at 5 <= L <= 16 or L in {32, 64, 128, 256} -> four-step path
    assert (p.path, p.bands, p.groups) == (1, 0, 1)
    p = L.plan(2, 16384, 8, 5000)                  # L = 64: two-level column transform, same plan fields
    assert (p.path, p.bands, p.groups, p.L) == (1, 0, 1, 64)
    p = L.plan(2, 9728, 8, 4000)                   # L = 38 = 2 x 19: band groups
    assert (p.path, p.bands, p.groups) == (1, 4, 8)
    assert L.plan(1, 1, 4, 2).k == 0               # N = 1 -> no bins
    assert L.plan(2, 20, 16, 8).k == 8 and L.plan(2, 21, 8, 100).k == 10   # floor(N/2)


def test_workspace_sizes(L):
    assert L.workspace_bytes(64, 4096, 256, 128) >= 64 * 128 * 256 * 8     # grad slab at least
    al = lambda v: (v + 255) // 256 * 256
    # direct plan: the 64 KiB sync area every layout starts with, three (B,k,D) complex spectra, the
    # LayerNorm-gradient partials of the block API
    assert L.workspace_bytes(2, 100, 8, 4) == 65536 + 3 * al(2 * 4 * 8 * 8) + al(50 * 2 * 8 * 4)
    L.set_option("nsplit", 4)
    try:
        assert L.plan(64, 4096, 256, 128).nsplit == 4
        assert L.workspace_bytes(64, 4096, 256, 128) > 64 * 8 * 4 * 32768
    finally:
        L.set_option("nsplit", 0)


def test_argument_validation_without_touching_the_gpu(L):
    lib = L.lib()
    err = lambda: lib.smx_last_error().decode()
    assert lib.smx_forward(None, None, None, None, None, None, None, 0, 0, 256, 2, 1, 0, None) == -1
    assert "positive" in err()
    assert lib.smx_forward(None, None, None, None, None, None, None, 0, 1, 256, 2, 1, 0, None) == -1
    assert "non-NULL" in err()
    assert lib.smx_backward(None, None, None, None, None, None, None, None, None, 0, 1, 256, 2, 1, 3,
                            None) == -1
    assert lib.smx_set_option(b"no_such_option", 1) == -1 and "unknown option" in err()
    assert lib.smx_set_option(None, 1) == -1
    with pytest.raises(L.SmxError, match="positive"):
        L.plan(1, 0, 1, 1)
    assert lib.smx_cmul(None, None, None, 0, 5, 0, None) == 0      # empty problem is a no-op
    assert lib.smx_cmul(None, None, None, 1, 5, 0, None) == -1


def test_round3_entry_points_validate_without_touching_the_gpu(L):
    """smx_conv_response / smx_phase_filter (and their backward calls): argument checks come before any HIP call;
    the conv1 plan knob is a field of smx_options and an option name."""
    lib = L.lib()
    err = lambda: lib.smx_last_error().decode()
    one = ctypes.c_float(0.0)
    p = ctypes.addressof(one)
    assert lib.smx_conv_response(2048, 4096, p, None, None, p, p, None) == -1 and "taps" in err()
    assert lib.smx_conv_response(2048, 128, None, None, None, p, p, None) == -1 and "non-NULL" in err()
    assert lib.smx_conv_response_backward(2048, 128, 1025, p, None, None, None, p, p, p, None) == -1 and "non-NULL" in err()
    assert lib.smx_conv_response_backward(2048, 128, 100, p, p, None, p, p, p, p, None) == -1 and "n_fft / 2 + 1" in err()
    assert lib.smx_phase_filter(p, p, 4, 600, 1024, p, p, None) == -1 and "k <=" in err()
    assert lib.smx_phase_filter(None, p, 4, 100, 1024, p, p, None) == -1 and "non-NULL" in err()
    assert lib.smx_phase_filter_backward(p, p, p, p, 4, 100, 1024, 50, p, p, None) == -1 and "row_pitch" in err()
    assert "conv1" in [n for n, _ in L.smx_options._fields_]
    with L.options(conv1=0):
        assert L.current_options()["conv1"] == 0
    assert lib.smx_set_option(b"conv1", 1) == 0


def test_block_entry_points_validate_without_touching_the_gpu(L):
    lib = L.lib()
    err = lambda: lib.smx_last_error().decode()
    assert [lib.smx_block_supported(d) for d in (1, 256, 4096, 4100, 8192, 1023, 1025)] == \
        [1, 1, 1, 0, 0, 1, 0]
    n = lambda k: [None] * k
    assert lib.smx_block_forward(*n(3), 1e-5, *n(6), None, 0, 1, 256, 8192, 4, None) == -2
    assert "LayerNorm width" in err()
    assert lib.smx_block_forward(*n(3), 1e-5, *n(6), None, 0, 1, 256, 8, 4, None) == -1
    assert "non-NULL" in err()
    assert lib.smx_block_backward(*n(13), None, 0, 1, 256, 8, 4, 0, None) == -1 and "phases" in err()
    assert lib.smx_block_backward(*n(13), None, 0, 1, 256, 8, 4, 3, None) == -1 and "non-NULL" in err()
    assert "spectral_layers.py:185" in open(HDR).read()


def test_general_shapes_plan_and_validation(L):
    """smx_*_ex: zero-padded rows, explicit bin count, Nyquist bin (no GPU needed for the plan)."""
    S = L.smx_shape
    p = L.plan_ex(S(2, 192, 32, 129, 256, 129))
    assert (p.path, p.bands, p.groups, p.k) == (L.SMX_PATH_DECIMATED, 1, 1, 129)      # self-paired Nyquist slot
    p = L.plan_ex(S(16, 1024, 64, 513, 1024, 513))
    assert (p.path, p.bands, p.groups) == (L.SMX_PATH_DECIMATED, 4, 1)
    p = L.plan_ex(S(1, 1024, 8, 1025, 2048, 1025))
    assert (p.path, p.bands, p.groups) == (L.SMX_PATH_DECIMATED, 0, 1)                # four-step path (L = 8)
    p = L.plan_ex(S(1, 4096, 8, 2049, 4096, 2049))
    assert (p.path, p.bands, p.groups) == (L.SMX_PATH_DECIMATED, 0, 1)                # four-step path (L = 16)
    p = L.plan_ex(S(1, 8704, 8, 4353, 8704, 4353))
    assert (p.path, p.bands, p.groups) == (L.SMX_PATH_DECIMATED, 4, 9)                # L = 34: band groups
    assert L.plan_ex(S(2, 100, 16, 65, 128, 65)).path == L.SMX_PATH_DIRECT
    # the layer's own entry points are the special case rows = n_fft, k = min(F, n_fft / 2)
    a, b = L.plan(64, 4096, 256, 128), L.plan_ex(S(64, 4096, 256, 128, 4096, 128))
    assert [getattr(a, f) for f, _ in a._fields_] == [getattr(b, f) for f, _ in b._fields_]
    assert L.workspace_bytes(64, 4096, 256, 128) == L.workspace_bytes_ex(S(64, 4096, 256, 128, 4096, 128))
    lib = L.lib()
    p = L.smx_plan()
    for bad in (S(2, 300, 8, 4, 256, 4),          # rows > n_fft
                S(2, 256, 8, 200, 256, 130),      # k > n_fft/2 + 1
                S(2, 256, 8, 100, 256, 101)):     # k > F
        assert lib.smx_plan_query_ex(ctypes.byref(bad), ctypes.byref(p)) == -1


# ---- round 3: plan knobs as an argument of the calling context, bounded memo tables, build identity ----------
def test_scoped_options_override_the_defaults_for_this_thread_only(L):
    import threading
    lib = L.lib()
    base = L.plan(64, 4096, 256, 128)
    key0 = L.opts_key()
    with L.options(nsplit=4):
        assert L.plan(64, 4096, 256, 128).nsplit == 4
        assert L.workspace_bytes(64, 4096, 256, 128) > 64 * 8 * 4 * 32768
        assert L.opts_key() != key0
        seen = []
        t = threading.Thread(target=lambda: seen.append(L.plan(64, 4096, 256, 128).nsplit))
        t.start(); t.join()
        assert seen == [base.nsplit]                      # another thread still plans with the defaults
        with L.options(force_direct=1):                   # nested: the innermost wins, the outer one comes back
            assert L.plan(64, 4096, 256, 128).path == 2
        assert L.plan(64, 4096, 256, 128).nsplit == 4
    assert L.plan(64, 4096, 256, 128).nsplit == base.nsplit and L.opts_key() == key0
    assert lib.smx_options_pop() == -1 and "without a matching push" in lib.smx_last_error().decode()
    with pytest.raises(ValueError, match="unknown plan option"):
        with L.options(no_such_knob=1):
            pass


def test_option_epoch_invalidates_python_memos_even_for_a_c_caller(L):
    """functional.py memoises per (shape, opts_key()); a smx_set_option made behind Python's back (a C caller in the
    same process) bumps the epoch, so no stale plan-dependent size can be served."""
    from tensor_cuda_fft_amd import functional as fn
    lib = L.lib()
    fn._ws_bytes_cache.clear()
    a = fn._ws_bytes(64, 4096, 256, 128)
    e0 = lib.smx_options_epoch()
    assert lib.smx_set_option(b"nsplit", 4) == 0              # raw C-ABI call, not _lib.set_option
    try:
        assert lib.smx_options_epoch() != e0
        assert fn._ws_bytes(64, 4096, 256, 128) > a           # recomputed under the new options
    finally:
        lib.smx_set_option(b"nsplit", 0)
    assert fn._ws_bytes(64, 4096, 256, 128) == a


def test_memo_tables_are_bounded():
    from tensor_cuda_fft_amd import functional as fn
    m = fn._Memo(cap=8)
    for i in range(100):
        assert m.get(("shape", i), lambda: i) == i
    assert len(m.d) == 8 and ("shape", 99) in m and ("shape", 0) not in m
    assert fn._WS_CACHE_MAX <= 64
    assert not hasattr(fn, "_ws_retired")                     # nothing is kept alive "forever" any more


def test_build_flags_identify_an_ablation_build(L):
    lib = L.lib()
    assert lib.smx_build_flags() == b""                       # the shipped build has no such switches
    assert lib.smx_version() >= 300
    with pytest.raises(L.SmxError, match="timing-ablation build"):
        L.check_build_flags(" SMX_AB_NO_FWD", "/x/libsmx_ab.so")
    L.check_build_flags(" SMX_NT_STORE=0", "/x/libsmx_plain_stores.so")      # a tuning build is not refused


def test_round4_entry_points_validate_their_arguments(L):
    """smx_dwconv3_*, smx_spectral_ln_*, smx_planar_*, smx_diag_clock: argument checks return codes and messages without
    touching a GPU (NULL pointers, non-positive shapes, the C <= 1024 rule of SpectralLayerNorm)."""
    import ctypes
    lib = L.lib()
    n = ctypes.c_size_t()
    assert lib.smx_dwconv3_workspace_bytes(2, 100, 16, ctypes.byref(n)) == 0
    assert n.value >= (2 * 4 * 5 * 16 + 2 * 4 * 16) * 4 and n.value % 256 == 0      # [B ceil(T/32)][5][C] + [B][4][C]
    assert lib.smx_dwconv3_workspace_bytes(0, 100, 16, ctypes.byref(n)) != 0
    assert lib.smx_dwconv3_forward(None, None, None, None, None, 2, 100, 16, None) != 0
    assert b"non-NULL" in lib.smx_last_error()
    assert lib.smx_dwconv3_backward(None, None, None, None, None, None, None, None, None, None, 0, 2, 100, 16, None) != 0
    assert lib.smx_spectral_ln_supported(1024) == 1 and lib.smx_spectral_ln_supported(1025) == 0
    assert lib.smx_spectral_ln_forward(None, None, None, 1e-5, None, 0, 2, 9, 2048, None) != 0
    assert b"1024" in lib.smx_last_error()
    assert lib.smx_spectral_ln_backward(None, None, None, None, 1e-5, None, None, None, 1, 2, 9, 16, None) != 0
    assert lib.smx_planar_cmul_forward(None, None, None, None, 2, 9, 16, None) != 0
    assert lib.smx_planar_cmul_backward(None, None, None, None, None, None, None, 2, 9, 16, None) != 0
    assert lib.smx_planar_add(None, None, None, 0, None) != 0 and lib.smx_planar_split(None, None, 10, None) != 0
    assert lib.smx_diag_clock(None, 10, None) != 0
    assert "st_plain" in [f for f, _ in L.smx_options._fields_]
    # 0.3.3: the gate chain and the fusion line of the twin blocks
    assert lib.smx_spectral_gate_workspace_bytes(4, 33, 256, ctypes.byref(n)) == 0
    assert n.value >= 4 * 2 * 33 * 8 and n.value % 256 == 0                      # [B ceil(C/128)][F] complex partials
    assert lib.smx_spectral_gate_workspace_bytes(4, 33, 255, ctypes.byref(n)) != 0
    assert b"even channel count" in lib.smx_last_error()
    assert lib.smx_spectral_gate_forward(None, None, None, None, None, None, None, 4, 33, 256, None) != 0
    assert b"non-NULL" in lib.smx_last_error()
    assert lib.smx_spectral_gate_backward(*([None] * 12), 0, 4, 33, 256, None) != 0
    assert lib.smx_mix_workspace_bytes(ctypes.byref(n)) == 0 and n.value % 256 == 0
    assert lib.smx_mix_forward(None, None, None, None, None, 0.1, None, 16, None) != 0
    assert lib.smx_mix_backward(None, None, None, None, 0.1, None, None, None, None, None, 0, 16, None) != 0


def test_batch_rows_of_two_gib_leave_the_streaming_plan(L):
    """The streaming kernels address a batch row as a raw buffer with 32-bit offsets (round 4): a row of 2 GiB or more
    (rows * D * 4 >= 2^31) is planned on the direct path instead -- correct, slow, and never an out-of-range offset."""
    assert L.plan(1, 65536, 8190, 64).path == L.SMX_PATH_DECIMATED          # 2 146 959 360 bytes: still streams
    assert L.plan(1, 65536, 8192, 64).path == L.SMX_PATH_DIRECT             # exactly 2 GiB

