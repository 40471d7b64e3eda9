"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/droid_hip.h declares; the droid_backends module exposes the reference's nine functions and refuses
CPU tensors (there is no CPU fallback).  No compute is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch  # noqa: F401  (must precede loading libdroid_hip: shares torch's HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "droid-slam_amd")
LIBDIR = os.path.join(PKG, "sanitize") if os.environ.get("DROID_HIP_TEST_SANITIZE", "0") == "1" else PKG
LIB = os.path.join(LIBDIR, "libdroid_hip.so")
EXT = os.path.join(LIBDIR, "droid_backends.so")


@pytest.fixture(scope="module")
def built():
    if not (os.path.exists(LIB) and os.path.exists(EXT)):
        subprocess.check_call([sys.executable, os.path.join(PKG, "build.py")])
    return LIB


def declared_functions():
    src = open(os.path.join(ROOT, "include", "droid_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_reference_entry_points():
    names = declared_functions()
    for must in ("dh_ba", "dh_corr_index_fwd", "dh_corr_index_bwd", "dh_altcorr_fwd", "dh_altcorr_bwd",
                 "dh_frame_distance", "dh_projmap", "dh_iproj", "dh_depth_filter"):
        assert must in names


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    lib.dh_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.dh_version()
    lib.dh_status_string.restype = ctypes.c_char_p
    assert lib.dh_status_string(1) == b"invalid argument"


def test_argument_errors_are_reported_before_any_launch(built):
    lib = ctypes.CDLL(built)
    # negative sizes / null pointers -> DH_ERR_ARG without touching the device
    assert lib.dh_corr_index_fwd(None, None, None, 0, 1, 4, 4, 4, 4, 3, None) == 1
    assert lib.dh_corr_index_fwd(None, None, None, 7, 0, 4, 4, 4, 4, 3, None) in (0, 4)
    assert lib.dh_frame_distance(None, None, None, None, None, None, 2, 4, 4, ctypes.c_float(0.3), None) == 1
    lib.dh_ba_workspace_bytes.restype = ctypes.c_size_t
    assert lib.dh_ba_workspace_bytes(8, 32, 48, 64, 1, 8, 0) > 0
    assert lib.dh_ba_workspace_bytes(8, 32, 48, 64, 5, 3, 0) == 0       # t1 < t0
    # entry points added for the MI355X path: same convention (status codes: 1 = argument, 4 = unsupported)
    lib.dh_corr_pyramid_bytes.restype = ctypes.c_size_t
    assert lib.dh_corr_pyramid_bytes(1, 48, 64) == 2 * 64 * 48 * sum((((48 >> l) + 1) * (64 >> l)) * 64 for l in range(4)) // 64
    assert lib.dh_corr_pyramid_bytes(1, 48, 8) == 0                      # w must be 16, 32 or 64
    assert lib.dh_corr_pyramid_lookup_nhwc(None, None, None, 4, 48, 64, None) == 1
    assert lib.dh_altcorr_fwd_nhwc(None, None, None, None, None, None, 2, 2, 64, 48, 64, 48, 64, 3, None) == 4   # C != 128
    assert lib.dh_altcorr_fwd_nhwc(None, None, None, None, None, None, 2, 2, 128, 48, 64, 48, 64, 3, None) == 1  # null pointers
    assert lib.dh_segment_mean_f16(None, None, None, None, 3, ctypes.c_long(12), None) == 1                       # row % 8
    assert lib.dh_conv2d_nhwc_f16(None, None, None, 1, None, None, None, 1, 8, 8, 3, 3, 4, 32, 64, 0,
                                  None, 0, 4, None, None, 0, None, 0, None, None) == 1
    # round 5: the accumulator-tile layout (out_is_f32 == 3 / cinit_stride < 0) takes whole 128-cout tiles and the plain epilogue only
    assert lib.dh_conv2d_nhwc_f16(None, None, None, 1, None, None, None, 1, 8, 64, 3, 3, 64, 64, 1152, 0,
                                  None, 3, 64, None, None, 0, None, 0, None, None) == 1
    assert lib.dh_conv2d_nhwc_f16(None, None, None, 1, None, None, None, 1, 8, 64, 3, 3, 128, 128, 1152, 1,
                                  None, 3, 128, None, None, 0, None, 0, None, None) == 1
    assert lib.dh_conv_set_timestamps(None, 0) == 4                      # measurement hook of the -DDH_ABLATION build only
    # frame-level pyramid build (round 5)
    lib.dh_corr_pyramid_prepared_bytes.restype = ctypes.c_size_t
    assert lib.dh_corr_pyramid_prepared_bytes(3, 48, 64) == 3 * (3072 + 768 + 192 + 48) * 128 * 2 and lib.dh_corr_pyramid_prepared_bytes(3, 48, 40) == 0
    assert lib.dh_corr_pyramid_prepare_frames(None, None, 2, 128, 48, 64, 48, 64, None) == 1          # null pointers
    assert lib.dh_corr_pyramid_prepare_frames(None, None, 0, 128, 48, 64, 48, 64, None) == 0          # no frames
    assert lib.dh_corr_pyramid_prepare_frames(None, None, 2, 64, 48, 64, 48, 64, None) == 4           # C != 128
    assert lib.dh_corr_pyramid_prepare_frames(None, None, 2, 128, 48, 64, 50, 64, None) == 1          # image larger than its canvas
    assert lib.dh_corr_pyramid_build_indexed(None, None, None, None, 2, 0, 48, 64, None) == 0         # no edges
    assert lib.dh_corr_pyramid_build_indexed(None, None, None, None, 2, 3, 48, 64, None) == 1
    rows, cols = ctypes.c_int(), ctypes.c_int()
    assert lib.dh_ba_system_shape(1, 512, ctypes.byref(rows), ctypes.byref(cols)) == 0
    assert cols.value == 3072 and rows.value == 3072 + 64


def test_droid_backends_module_surface(built):
    import droid_backends as db
    for name in ("ba", "frame_distance", "projmap", "depth_filter", "iproj", "altcorr_forward",
                 "altcorr_backward", "corr_index_forward", "corr_index_backward"):     # src/droid.cpp:246-259
        assert callable(getattr(db, name))


def test_droid_backends_has_no_cpu_fallback(built):
    import droid_backends as db
    with pytest.raises(RuntimeError, match="ROCm device tensor"):
        db.iproj(torch.zeros(2, 7), torch.ones(2, 4, 4), torch.ones(4))
    with pytest.raises(RuntimeError, match="ROCm device tensor"):
        db.corr_index_forward(torch.zeros(1, 4, 4, 4, 4), torch.zeros(1, 2, 4, 4), 3)


def test_dropin_packages_import_and_torch_scatter_semantics(built):
    """the reference's import closure (depth_video.py:3-10, factor_graph.py:1-11, droid_net.py:18, geom/ba.py:1-8) needs
    `droid_backends`, `lietorch` and `torch_scatter`: all three resolve inside this repository"""
    import torch
    import lietorch
    import torch_scatter
    assert {"SE3", "Sim3", "SO3", "cat"} <= set(dir(lietorch))
    x = torch.arange(12.0).reshape(6, 2); ix = torch.tensor([0, 1, 0, 2, 2, 2])
    assert torch.equal(torch_scatter.scatter_sum(x, ix, dim=0), torch.stack([x[[0, 2]].sum(0), x[1], x[3:].sum(0)]))
    assert torch.allclose(torch_scatter.scatter_mean(x, ix, dim=0), torch.stack([x[[0, 2]].mean(0), x[1], x[3:].mean(0)]))


def test_sim3_group_algebra(built):
    """Sim3 (training / trajectory alignment only; plain tensor arithmetic): product, inverse and action against 4x4 matrices"""
    import lietorch
    torch.manual_seed(0)

    def rnd(n):
        d = torch.randn(n, 8, dtype=torch.float64)
        d[:, 3:7] /= d[:, 3:7].norm(dim=-1, keepdim=True); d[:, 7] = d[:, 7].abs() + 0.5
        return lietorch.Sim3(d)
    a, b = rnd(5), rnd(5)
    Ma, Mb = a.matrix(), b.matrix()
    assert (Ma @ Mb - (a * b).matrix()).abs().max() < 1e-12
    assert (torch.linalg.inv(Ma) - a.inv().matrix()).abs().max() < 1e-12
    X = torch.randn(5, 7, 4, dtype=torch.float64)
    assert ((Ma[:, None] @ X[..., None])[..., 0] - a.act(X)).abs().max() < 1e-12
    assert torch.equal(lietorch.Sim3.Identity(2, device="cpu").data[0], torch.tensor([0, 0, 0, 0, 0, 0, 1.0, 1.0]))
    with pytest.raises(NotImplementedError):
        a.log()


def test_unmodified_reference_modules_import_against_this_repo(built):
    """factor_graph.py / depth_video.py / droid_net.py of the reference checkout resolve their whole import closure
    (droid_backends, lietorch, torch_scatter) inside droid-slam_amd/ -- in a subprocess, reference untouched.
    Skipped where the reference checkout is absent (the GPU box)."""
    import subprocess
    import sys
    ref = "/root/reference/droid_slam"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "import importlib, warnings; warnings.simplefilter('ignore')\n"
            "for m in ['depth_video', 'factor_graph', 'droid_net', 'geom.projective_ops', 'geom.ba', 'modules.corr']:\n"
            "    importlib.import_module(m)\n"
            "import droid_backends, lietorch\n"
            "assert droid_backends.__file__.startswith(%r) and lietorch.__file__.startswith(%r)\n"
            "print('closure ok')\n") % (PKG, ref, PKG, PKG)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd="/tmp")
    assert r.returncode == 0 and "closure ok" in r.stdout, r.stdout[-2000:]


def test_convolution_weight_packing_layouts(built):
    """host logic of the update operator: the kernel-ordered weight copies (`pack_conv_halo`) follow the layout rules that
    csrc/conv.hip applies on its side (halo2 / 16- and 32-channel halo slabs / opt-in LDS-DMA), checked element by element"""
    import random
    import torch
    from droid_amd import update as U
    rnd = random.Random(0)
    w = torch.randn(256, 448, 3, 3).half()
    # default: conv3x3_halo2_kernel, [T, chunk32, dy, dx, row, slot', 8] with slot' = slot ^ ((row >> 2) & 3)
    import droid_backends as db
    saved = {k: db.get_option(k) for k in ("conv_dma", "conv_halo2")}
    db.set_option("conv_dma", 0); db.set_option("conv_halo2", 1)
    p = U.pack_conv_halo(w)
    assert tuple(p.shape) == (2, 14, 3, 3, 128, 4, 8)
    for _ in range(500):
        T, c, dy, dx, r, sp, e = (rnd.randrange(n) for n in (2, 14, 3, 3, 128, 4, 8))
        assert p[T, c, dy, dx, r, sp, e] == w[T * 128 + r, c * 32 + (sp ^ ((r >> 2) & 3)) * 8 + e, dy, dx]
    # first halo kernel: [T, chunk16, tap, 128, 16]
    db.set_option("conv_halo2", 0)
    p = U.pack_conv_halo(w)
    assert tuple(p.shape) == (2, 28, 9, 128, 16)
    for _ in range(500):
        T, c, t, r, e = (rnd.randrange(n) for n in (2, 28, 9, 128, 16))
        assert p[T, c, t, r, e] == w[T * 128 + r, c * 16 + e, t // 3, t % 3]
    # opt-in LDS-DMA kernel: [T, chunk64, tap, row, slot', 8] with slot' = slot ^ ((row >> 1) & 7) -- only a -DDH_ABLATION
    # build carries that prototype; the shipped library refuses the switch
    if db.get_option("ablation_build"):
        db.set_option("conv_dma", 1)
        p = U.pack_conv_halo(w)
        assert tuple(p.shape) == (2, 7, 9, 128, 8, 8)
        for _ in range(500):
            T, c, t, r, sp, e = (rnd.randrange(n) for n in (2, 7, 9, 128, 8, 8))
            assert p[T, c, t, r, sp, e] == w[T * 128 + r, c * 64 + (sp ^ ((r >> 1) & 7)) * 8 + e, t // 3, t % 3]
    else:
        with pytest.raises(RuntimeError):
            db.set_option("conv_dma", 1)
    db.set_option("conv_dma", 0); db.set_option("conv_halo2", 1)
    # small cout tiles: 32-channel slabs, cout padded to the tile
    p = U.pack_conv_halo(torch.randn(4, 256, 3, 3).half())
    assert tuple(p.shape) == (1, 8, 9, 32, 32) and torch.count_nonzero(p[0, :, :, 4:]) == 0
    # 64-cout layers (flow_encoder.2): conv3x3_halo64_kernel (option conv_halo64, default 1) reads the halo2 layout of the layer PADDED
    # to 128 couts -- rows 64..127 zero; with the option off, the first halo kernel's 64-cout slabs
    saved64 = db.get_option("conv_halo64")
    assert saved64 == 1
    w64 = torch.randn(64, 128, 3, 3).half()
    p = U.pack_conv_halo(w64)
    assert tuple(p.shape) == (1, 4, 3, 3, 128, 4, 8) and torch.count_nonzero(p[:, :, :, :, 64:]) == 0
    for _ in range(500):
        c, dy, dx, r, sp, e = (rnd.randrange(n) for n in (4, 3, 3, 64, 4, 8))
        assert p[0, c, dy, dx, r, sp, e] == w64[r, c * 32 + (sp ^ ((r >> 2) & 3)) * 8 + e, dy, dx]
    p48 = U.pack_conv_halo(torch.randn(48, 128, 3, 3).half())                       # CoutPad = 64 as well
    assert tuple(p48.shape) == (1, 4, 3, 3, 128, 4, 8) and torch.count_nonzero(p48[:, :, :, :, 48:]) == 0
    db.set_option("conv_halo64", 0)
    p = U.pack_conv_halo(w64)
    assert tuple(p.shape) == (1, 4, 9, 64, 32)
    for _ in range(200):
        c, t, r, e = (rnd.randrange(n) for n in (4, 9, 64, 32))
        assert p[0, c, t, r, e] == w64[r, c * 32 + e, t // 3, t % 3]
    db.set_option("conv_halo64", 1); db.set_option("conv_halo2", 0)                 # the padded layout needs the halo2 rule too
    assert tuple(U.pack_conv_halo(w64).shape) == (1, 4, 9, 64, 32)
    db.set_option("conv_halo2", 1); db.set_option("conv_halo64", saved64)
    assert U.pack_conv_halo(torch.randn(128, 8, 7, 7).half()) is None              # not a 3x3 kernel
    # correlation channel map: our level-planar (yoff, xoff) order against the reference's (xoff, yoff) order
    m = U.corr_channel_map()
    assert m.numel() == 224 and (m >= 0).sum() == 196 and sorted(m[m >= 0].tolist()) == list(range(196))
    # first correlation layer packed for the fused lookup kernel: every reference channel exactly once, in the kernel's K order
    w0 = torch.randn(128, 196).half().float()
    p = U.pack_corr0_fused(w0)
    assert tuple(p.shape) == (13, 128, 16) and p.dtype == torch.float16
    seen = set()
    for l in range(4):
        for kk in range(48):
            ch = l * 49 + (kk % 7) * 7 + kk // 7                      # level*49 + xoff*7 + yoff, kk = yoff*7 + xoff
            assert torch.equal(p[l * 3 + kk // 16, :, kk % 16].float(), w0[:, ch]); seen.add(ch)
        assert torch.equal(p[12, :, l].float(), w0[:, l * 49 + 48]); seen.add(l * 49 + 48)
    assert seen == set(range(196)) and torch.count_nonzero(p[12, :, 4:]) == 0
    for k, v in saved.items():
        db.set_option(k, v)


def test_option_store(built):
    """dh_set_option / dh_get_option: known names round-trip, unknown names are an argument error"""
    import droid_backends as db
    v = db.get_option("chol_lookahead")
    db.set_option("chol_lookahead", 0)
    assert db.get_option("chol_lookahead") == 0
    db.set_option("chol_lookahead", v)
    with pytest.raises(RuntimeError):
        db.set_option("no_such_option", 1)


def test_round3_entry_points_and_host_helpers(built):
    """argument checks of the entry points added in round 3 (no launch), the option epoch, and the host-side size / staging
    helpers that mirror them"""
    lib = ctypes.CDLL(built)
    assert lib.dh_corr_volume_build(None, None, None, 0, 2, 128, 30, 40, None) == 1            # null pointers
    assert lib.dh_corr_volume_build(None, None, None, 0, 0, 128, 30, 40, None) == 0            # no edges: nothing to do
    assert lib.dh_corr_volume_pool(None, None, 0, ctypes.c_long(0), 30, 40, None) == 0
    assert lib.dh_corr_volume_pool(None, None, 7, ctypes.c_long(4), 30, 40, None) == 1         # null pointers before the dtype is looked at
    assert lib.dh_altcorr_fwd_nhwc_level(None, None, None, None, None, None, 2, 2, 128, 48, 64, 24, 32, 3, 1, ctypes.c_long(10), None) == 1   # stride < 49 HW
    assert lib.dh_reproject_ex(None, None, None, 1, None, None, None, None, 0, 48, 64, None) == 0
    assert lib.dh_corr_pyramid_lookup_corr0(None, None, None, None, None, 0, 48, 64, None) == 0     # no edges
    assert lib.dh_corr_pyramid_lookup_corr0(None, None, None, None, None, 2, 48, 64, None) == 1     # null pointers
    assert lib.dh_corr_pyramid_lookup_corr0(None, None, None, None, None, 2, 30, 40, None) == 1     # not a pyramid shape
    assert lib.dh_get_option(b"lookup_fused", ctypes.byref(ctypes.c_int())) == 0
    assert lib.dh_corr_index_fwd(None, None, None, 2, 0, 4, 4, 4, 4, 3, None) == 0             # DH_F64 accepted
    e0 = lib.dh_options_epoch()
    v = ctypes.c_int()
    assert lib.dh_get_option(b"lookup_mode", ctypes.byref(v)) == 0
    assert lib.dh_set_option(b"lookup_mode", v.value) == 0 and lib.dh_options_epoch() == e0    # same value: no new epoch
    assert lib.dh_set_option(b"lookup_mode", v.value + 1) == 0 and lib.dh_options_epoch() == e0 + 1
    lib.dh_set_option(b"lookup_mode", v.value)
    assert lib.dh_get_option(b"ba_strict", ctypes.byref(v)) == 0 and v.value == 1               # ba reports bad arguments by default
    from droid_amd.corr import CorrBlock
    lib.dh_corr_pyramid_bytes.restype = ctypes.c_size_t
    for h, w in ((48, 64), (40, 64), (16, 16), (24, 32)):
        assert CorrBlock.bytes_per_edge(h, w) == lib.dh_corr_pyramid_bytes(1, h, w)
    assert CorrBlock.supported(40, 64) and CorrBlock.canvas(40, 64) == (40, 64) and CorrBlock.canvas(48, 64) == (48, 64)
    # sizes outside the layout up to 64 columns sit on a zero-padded canvas; wider images up to 64 rows sit on it TRANSPOSED;
    # more than 64 in both dimensions: 64-column strips, one record per (source strip, target strip) pair (round 5)
    assert CorrBlock.supported(30, 40) and CorrBlock.canvas(30, 40) == (32, 64) and CorrBlock.canvas(12, 16) == (16, 16)
    assert CorrBlock.supported(30, 80) and CorrBlock.is_transposed(30, 80) and CorrBlock.canvas(30, 80) == (80, 32)
    assert CorrBlock.canvas(60, 80) == (80, 64) and CorrBlock.canvas(41, 73) == (80, 64) and not CorrBlock.is_transposed(64, 64)
    assert CorrBlock.bytes_per_edge(41, 73) == lib.dh_corr_pyramid_bytes(1, 80, 64)
    assert CorrBlock.supported(72, 80) and CorrBlock.canvas(72, 96) == (72, 64) and CorrBlock.canvas(65, 65) == (72, 64)
    assert CorrBlock.strip_bounds(72, 96) == [(0, 64), (64, 32)] and CorrBlock.strip_bounds(65, 130) == [(0, 64), (64, 64), (128, 2)]
    assert CorrBlock.strip_bounds(64, 96) is None and CorrBlock.strip_bounds(96, 64) is None and CorrBlock.strip_bounds(48, 64) is None
    assert CorrBlock.bytes_per_edge(72, 96) == 4 * lib.dh_corr_pyramid_bytes(1, 72, 64)
    assert CorrBlock.bytes_per_edge(96, 64) == lib.dh_corr_pyramid_bytes(1, 96, 64)          # (more rows than 64 alone need no strips)
    assert CorrBlock.bytes_per_edge(30, 40) == CorrBlock.bytes_per_edge(32, 64)


def test_reference_python_is_staged_as_an_archive_only():
    """oracle/build_ref.py stages the reference's callers for the -m gpu test as ONE git-ignored archive (no reference source
    file is unpacked into the tree)"""
    from oracle import build_ref
    if not os.path.isdir(os.path.join(build_ref.REF, "droid_slam")):
        pytest.skip("reference checkout not present")
    z = build_ref.stage_python()
    import zipfile
    names = zipfile.ZipFile(z).namelist()
    assert "droid_slam/factor_graph.py" in names and "droid_slam/depth_video.py" in names
    assert z.startswith(os.path.join(ROOT, "oracle", "_ref"))
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
    assert tracked == ""


def test_shipped_library_is_a_release_build_without_wrong_result_modes(built):
    """VERDICT r3 weak #10 / ADVICE: the timing ablations of the lookup (lookup_mode 2-5: wrong results by construction) and the
    prototype kernels (LDS-DMA / Winograd convolutions, first forms of the pyramid build and alt-correlation kernels) exist only
    in a -DDH_ABLATION build (DROID_HIP_ABLATION=1 python droid-slam_amd/build.py).  The library in the tree is the release
    build: the variants are not compiled in, and neither dh_set_option nor the environment can select them."""
    import subprocess
    lib = ctypes.CDLL(LIB)
    v = ctypes.c_int(-1)
    assert lib.dh_get_option(b"ablation_build", ctypes.byref(v)) == 0 and v.value == 0
    for mode in (2, 3, 4, 5):
        assert lib.dh_set_option(b"lookup_mode", mode) != 0
    for mode in (1, 6, 0):                                        # nt tap loads / the synchronous twin: same results
        assert lib.dh_set_option(b"lookup_mode", mode) == 0
    for name in (b"conv_dma", b"conv_wino", b"pyr_build_chunk", b"altcorr_v1", b"dma_var", b"conv_abl", b"conv_halo3"):
        assert lib.dh_set_option(name, 1) != 0 and lib.dh_set_option(name, 0) == 0
        assert lib.dh_get_option(name, ctypes.byref(v)) == 0 and v.value == 0
    # a stray environment variable is ignored as well (fresh process: the option store is initialised at first use)
    code = ("import ctypes; lib = ctypes.CDLL(%r); v = ctypes.c_int(-1); "
            "assert lib.dh_get_option(b'lookup_mode', ctypes.byref(v)) == 0; print(v.value); "
            "assert lib.dh_get_option(b'conv_wino', ctypes.byref(v)) == 0; print(v.value)" % LIB)
    env = dict(os.environ, DH_LOOKUP_MODE="3", DH_CONV_WINO="1", DH_CONV_DMA="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert out == ["0", "0"]
    # the prototype kernels are not in the code object
    syms = subprocess.run(["nm", "-D", "--defined-only", LIB], stdout=subprocess.PIPE, text=True).stdout
    blob = open(LIB, "rb").read()
    for kern in (b"conv3x3_wino_kernel", b"conv3x3_dma_kernel", b"conv3x3_halo3_kernel", b"altcorr_mfma_kernel", b"pyr_build_kernel"):
        assert kern not in blob, kern
    assert b"conv3x3_halo2_kernel" in blob and b"pyr_build_ring_kernel" in blob and b"altcorr_mfma2_kernel" in blob
