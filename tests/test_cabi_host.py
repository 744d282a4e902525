"""CPU tier: the C-ABI library loads without a GPU, exports every symbol include/advx.h
declares, and its HOST logic (integer geometry, tap tables, layout index maps) is
bit-exact against the oracle and the fixtures.  No compute entry point is called here."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from conftest import ROOT, load_golden
from oracle import geometry as G
from oracle import resample as R


@pytest.fixture(scope="module")
def lib():
    from adversarialvlm_amd.build import build_library
    build_library()
    from adversarialvlm_amd import _lib
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "advx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(advx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    from adversarialvlm_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (advx_[a-z0-9_]+)", out))
    assert set(declared) <= exported, sorted(set(declared) - exported)
    assert set(declared) == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.advx_version() >= 100


def test_product_path_has_no_cpu_fallback():
    from adversarialvlm_amd import ops
    from adversarialvlm_amd._lib import AdvxError
    with pytest.raises(AdvxError):
        ops.tanh_fwd(torch.zeros(3, 4, 4), 0.5)          # CPU tensor: must fail loudly
    # and nothing under the package imports the oracle
    pkg = os.path.join(ROOT, "adversarialvlm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_error_reporting(lib):
    from adversarialvlm_amd._lib import AdvxError
    from adversarialvlm_amd.plan import Plan
    with pytest.raises(AdvxError, match="unknown kind"):
        Plan(99, 10, 10)
    with pytest.raises(AdvxError, match="size out of range"):
        Plan.llava(0, 10)
    with pytest.raises(AdvxError):
        Plan.qwen2vl(336, 336, patch=0)


def test_mllama_geometry_bit_exact(lib):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("mllama_helpers.npz")
    for h, w, mt, ts, ch, cw, nh, nw in g["geometry"]:
        p = Plan.mllama(int(h), int(w), tile=int(ts), max_tiles=int(mt))
        s = p.stage(0)
        assert (s.can_h, s.can_w, s.res_h, s.res_w) == (int(ch), int(cw), int(nh), int(nw))
        assert p.info.aspect_ratio_id == G.mllama_aspect_ratio_id(int(ch) // int(ts), int(cw) // int(ts), int(mt))
        assert p.out_shape == (1, 1, int(mt), 3, int(ts), int(ts))


def test_qwen_geometry_bit_exact(lib):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("qwen2vl_reference.npz")
    for h, w, hb, wb in g["geometry"]:
        p = Plan.qwen2vl(int(h), int(w))
        assert (p.stage(0).res_h, p.stage(0).res_w) == (int(hb), int(wb))
        assert p.out_shape == ((int(hb) // 14) * (int(wb) // 14), 1176)
        assert p.info.num_tiles == (int(hb) // 14) * (int(wb) // 14)


def test_phi3_geometry_bit_exact(lib):
    from adversarialvlm_amd.plan import Plan
    g = load_golden("phi3_reference.npz")
    for h, w, oh, ow, ntok in g["geometry"]:
        p = Plan.phi3(int(h), int(w))
        assert (p.info.image_h, p.info.image_w, p.info.num_img_tokens) == (int(oh), int(ow), int(ntok))
        assert p.out_shape == (1, 7, 3, 336, 336)


@settings(max_examples=150, deadline=None)
@given(h=st.integers(8, 2400), w=st.integers(8, 2400))
def test_geometry_property_vs_oracle(lib, h, w):
    from adversarialvlm_amd.plan import Plan
    nh, nw, th, tw = G.mllama_geometry(h, w)
    s = Plan.mllama(h, w).stage(0)
    assert (s.res_h, s.res_w, s.can_h // 560, s.can_w // 560) == (nh, nw, th, tw)
    q = Plan.qwen2vl(h, w).stage(0)
    assert (q.res_h, q.res_w) == G.qwen_smart_resize(h, w)
    g = G.phi3_hd_geometry(h, w, 6)
    if g["out_h"] % 336 == 0 and g["out_w"] % 336 == 0 and 1 <= (g["out_h"] // 336) * (g["out_w"] // 336) <= 6:
        f = Plan.phi3(h, w)
        assert (f.info.image_h, f.info.image_w) == (g["out_h"], g["out_w"])
        assert f.info.num_img_tokens == G.phi3_num_img_tokens(g["out_h"], g["out_w"])
        a = f.stage(0)
        if g["trans"]:
            assert (a.res_h, a.res_w, a.off_y, a.off_x) == (g["new_w"], g["new_h"], 0, g["pad_top"])
        else:
            assert (a.res_h, a.res_w, a.off_y, a.off_x) == (g["new_h"], g["new_w"], g["pad_top"], 0)


ORACLE_TAPS = {0: R.aa_bilinear_taps, 1: R.bilinear_taps, 2: R.bicubic_taps}


@settings(max_examples=120, deadline=None)
@given(mode=st.integers(0, 2), n=st.integers(2, 700), m=st.integers(2, 700))
def test_tap_tables_bit_exact_vs_oracle(lib, mode, n, m):
    """fp32 tap tables computed by the library's host code == the oracle's restatement of
    ATen (which tests/test_oracle_resample.py pins to F.interpolate), bit for bit."""
    from adversarialvlm_amd.plan import taps_compute
    s, c, w = taps_compute(mode, n, m)
    so, co, wo = ORACLE_TAPS[mode](n, m)
    k = min(w.shape[1], wo.shape[1])
    assert np.array_equal(s, so) and np.array_equal(c, co)
    assert np.array_equal(w[:, :k], wo[:, :k])
    assert not w[:, k:].any() and not wo[:, k:].any()
    ts, tc, tw = taps_compute(mode, n, m, transposed=True)
    assert np.array_equal(R.taps_to_matrix((ts, tc, tw), m), R.taps_to_matrix((s, c, w), n).T)


@settings(max_examples=150, deadline=None)
@given(mode=st.integers(0, 2), n=st.integers(2, 700), m=st.integers(2, 700), same=st.booleans())
def test_device_tap_rows_are_the_full_rows_without_their_zero_ends(lib, mode, n, m, same):
    """advx_plan_upload stores rows without the zero-weight taps at their ends (a resize between equal sizes has
    bicubic rows (0, 1, 0, 0)); every dropped tap had weight exactly zero, every kept one is unchanged, in place."""
    from adversarialvlm_amd.plan import taps_compute
    if same:
        m = n
    for transposed in (False, True):
        s, c, w = taps_compute(mode, n, m, transposed=transposed)
        ds, dc, dw = taps_compute(mode, n, m, transposed=transposed, device_rows=True)
        cols = n if not transposed else m
        assert np.array_equal(R.taps_to_matrix((ds, dc, dw), cols), R.taps_to_matrix((s, c, w), cols))
        assert dw.shape[1] == max(1, int(dc.max()))
        for i in range(len(s)):
            row, drow = w[i, :c[i]], dw[i, :dc[i]]
            lead = ds[i] - s[i]
            assert lead >= 0 and lead + dc[i] <= max(c[i], 0) or dc[i] == 0
            if dc[i]:
                assert drow[0] != 0 and drow[-1] != 0
                assert np.array_equal(row[lead:lead + dc[i]], drow)
                assert not row[:lead].any() and not row[lead + dc[i]:].any()
            else:
                assert not row.any()
    if same and mode == 2:
        ds, dc, dw = taps_compute(mode, n, n, device_rows=True)
        assert dw.shape[1] == 1 and np.all(dc == 1) and np.array_equal(ds, np.arange(n)) and np.all(dw == 1.0)


@settings(max_examples=150, deadline=None)
@given(mode=st.integers(0, 2), n=st.integers(2, 700), m=st.integers(2, 700))
def test_device_side_builder_forms_the_same_transposed_table(lib, mode, n, m):
    """The crop window's transposed table is built on the device from tap_bounds_transposed + tap_weight (no row buffer);
    the same routine run on the host gives the table the host derives from the forward rows, bit for bit."""
    from adversarialvlm_amd.plan import taps_compute
    # (mode 0, antialiased bilinear, is the crop window's resize; tap_weight covers the other two as well)
    ts, tc, tw = taps_compute(mode, n, m, transposed=True)
    bs, bc, bw = taps_compute(mode, n, m, transposed=True, builder=True)
    assert np.array_equal(ts, bs) and np.array_equal(tc, bc) and np.array_equal(tw, bw)


def test_layout_index_maps_vs_reference_permutes(lib):
    """Tile / patch index maps are integer-exact against the reshape-permute-reshape of the
    reference (llama32processor.py:326-332, phi3processor.py:227, qwen2VLprocessor.py:249-267)."""
    from adversarialvlm_amd.plan import Plan
    rng = np.random.default_rng(0)
    # qwen
    p = Plan.qwen2vl(60, 90, min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64)
    sg = p.stage(0)
    for _ in range(300):
        c, y, x = int(rng.integers(3)), int(rng.integers(sg.can_h)), int(rng.integers(sg.can_w))
        idx = p.out_index(0, c, y, x)
        want = []
        for t in range(2):
            r, col = G.qwen_patch_index(c, t, y, x, p.info.grid_w)
            want.append(r * 1176 + col)
        assert idx == want
    # mllama tiles: emulate split_to_tiles on an index canvas
    p = Plan.mllama(1352, 1988)
    sg = p.stage(0)
    th, tw = sg.can_h // 560, sg.can_w // 560
    canvas = torch.arange(3 * sg.can_h * sg.can_w).reshape(3, sg.can_h, sg.can_w)
    tiles = canvas.reshape(3, th, 560, tw, 560).permute(1, 3, 0, 2, 4).reshape(-1)
    for _ in range(300):
        c, y, x = int(rng.integers(3)), int(rng.integers(sg.can_h)), int(rng.integers(sg.can_w))
        (i,) = p.out_index(0, c, y, x)
        assert int(tiles[i]) == int(canvas[c, y, x])
    # phi3: global view is tile 0 (stage 1), local tiles follow (stage 0)
    p = Plan.phi3(300, 500)
    sg = p.stage(0)
    canvas = torch.arange(3 * sg.can_h * sg.can_w).reshape(3, sg.can_h, sg.can_w)
    local = canvas.reshape(1, 3, sg.can_h // 336, 336, sg.can_w // 336, 336).permute(0, 2, 4, 1, 3, 5).reshape(-1)
    T = 3 * 336 * 336
    for _ in range(300):
        c, y, x = int(rng.integers(3)), int(rng.integers(sg.can_h)), int(rng.integers(sg.can_w))
        (i,) = p.out_index(0, c, y, x)
        assert i >= T and int(local[i - T]) == int(canvas[c, y, x])
    assert p.out_index(1, 2, 335, 335) == [T - 1]


@pytest.mark.parametrize("kind,hw", [("mllama", (336, 336)), ("mllama", (300, 700)), ("phi3", (336, 336)), ("phi3", (500, 900)),
                                     ("llava", (96, 80)), ("qwen2vl", (120, 200))])
def test_live_range_is_where_the_oracle_output_is_not_padding(lib, kind, hw):
    """advx_plan_live_range: outside [lo, hi) the processors return their constant zero tiles
    (llama32processor.py:344-346, phi3processor.py:232-235) - whose gradient the batch
    reduction skips and which ADVX_PAD_KEEP never rewrites; inside it an image (or the padding
    of its canvas, which is normalised and therefore not zero) is written."""
    from adversarialvlm_amd.plan import Plan
    from oracle.processors import LlavaOracle, MllamaOracle, Phi3Oracle, Qwen2VLOracle
    H, W = hw
    x = torch.rand(3, H, W, generator=torch.Generator().manual_seed(H + W)) * 0.8 + 0.1
    if kind == "mllama":
        plan, out = Plan.mllama(H, W), MllamaOracle().process(x)["pixel_values"]
    elif kind == "phi3":
        plan, out = Plan.phi3(H, W), Phi3Oracle().process(x)["pixel_values"]
    elif kind == "llava":
        plan, out = Plan.llava(H, W, 48, 48), LlavaOracle(48, 48).process(x)["pixel_values"]
    else:
        plan = Plan.qwen2vl(H, W, min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64)
        out = Qwen2VLOracle(min_pixels=28 * 28 * 4, max_pixels=28 * 28 * 64).process(x)["pixel_values"]
    flat = out.reshape(-1)
    lo, hi = plan.live_range()
    assert flat.numel() == plan.out_numel and 0 <= lo < hi <= plan.out_numel
    assert not bool((flat[:lo] != 0).any()) and not bool((flat[hi:] != 0).any())
    assert float((flat[lo:hi] != 0).float().mean()) > 0.99
    if kind in ("llava", "qwen2vl"):
        assert (lo, hi) == (0, plan.out_numel)


def test_header_is_plain_c():
    """include/advx.h is the drop-in boundary: it must compile as C99 on its own (no torch, no
    C++), so that cgo / JNI / ctypes / a C caller can bind it."""
    res = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Werror",
                          os.path.join(ROOT, "include", "advx.h")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_multi_plan_entry_points_validate_before_any_device_work(lib):
    """Argument checks of advx_emit_multi / advx_collect_multi / advx_forward_multi / advx_image_bwd_update /
    advx_update_flush come before the first HIP call: they can be exercised without a GPU and must
    return ADVX_E_* with a message, never crash."""
    import ctypes as C
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd.plan import Plan
    a, b, other = Plan.llava(40, 52, 32, 32), Plan.mllama(40, 52, tile=16), Plan.llava(41, 52, 32, 32)
    E_BADARG, E_SHAPE = -1, -2                                 # ADVX_E_BADARG, ADVX_E_SHAPE (include/advx.h)
    fake = lambda k: C.c_void_p(0x1000 * (k + 1))              # never dereferenced: checks fail first
    arr_p = lambda *v: (C.c_void_p * len(v))(*v)
    plans2 = arr_p(a.handle.value, b.handle.value)
    b2, w2 = (C.c_int32 * 2)(1, 1), (C.c_int64 * 2)(a.workspace_floats, b.workspace_floats)
    offs = (C.c_uint64 * 2)(0, 0)
    two_ws, same_ws = arr_p(0x1000, 0x2000), arr_p(0x1000, 0x1000)

    def emit(n, plans, batches, wss, wsf):
        return lib.advx_emit_multi(n, plans, fake(7), batches, None, None, 0, 0, offs, arr_p(0x3000, 0x4000), wss, wsf, 0, None)

    assert emit(0, plans2, b2, two_ws, w2) == E_BADARG and b"between 1 and 4" in lib.advx_last_error()
    assert emit(5, plans2, b2, two_ws, w2) == E_BADARG
    assert emit(2, plans2, b2, same_ws, w2) == E_BADARG and b"own workspace" in lib.advx_last_error()
    assert emit(2, plans2, (C.c_int32 * 2)(1, 0), two_ws, w2) == E_BADARG
    assert emit(2, plans2, b2, two_ws, (C.c_int64 * 2)(a.workspace_floats, 1)) == E_SHAPE
    assert emit(2, arr_p(a.handle.value, other.handle.value), b2, two_ws, w2) == E_SHAPE
    assert b"same image" in lib.advx_last_error()
    assert lib.advx_collect_multi(2, plans2, None, b2, fake(1), 0, two_ws, w2, None) == E_BADARG
    assert lib.advx_collect_multi(2, plans2, arr_p(0x5000, 0x6000), b2, None, 0, two_ws, w2, None) == E_BADARG
    assert lib.advx_forward_multi(None, fake(1), 40, 52, 0.5, 0, 0.0, None, fake(2), None, fake(3), fake(4), 2, plans2, b2, None,
                                  0, 0, offs, arr_p(0x3000, 0x4000), two_ws, w2, 0, None) == E_BADARG
    opt = L.OptScalars()
    assert lib.advx_image_bwd_update(fake(1), fake(2), fake(3), 40, 52, 0.5, 0, 0.0, None, 1.0, fake(4), 0, None, fake(5),
                                     fake(6), C.byref(opt), fake(7), fake(8), fake(9), 1, None) == E_BADARG
    assert lib.advx_image_bwd_update(fake(1), fake(2), fake(3), 0, 52, 0.5, 0, 0.0, None, 1.0, fake(4), 0, fake(5), fake(5),
                                     fake(6), C.byref(opt), fake(7), fake(8), fake(9), 1, None) == E_SHAPE
    assert lib.advx_update_flush(0, fake(1), fake(2), None) == E_BADARG


def test_round4_entry_points_validate_before_any_device_work(lib):
    """advx_collect_update / advx_ce_fwd / advx_ce_bwd: argument checks before the first HIP call; advx_ce_scratch_floats is plain
    host arithmetic (two floats per row chunk of 16 384 halfs / 8 192 floats, plus slack)."""
    import ctypes as C
    from adversarialvlm_amd import _lib as L
    from adversarialvlm_amd.plan import Plan
    E_BADARG, E_SHAPE = -1, -2
    fake = lambda k: C.c_void_p(0x1000 * (k + 1))              # never dereferenced: checks fail first
    a = Plan.llava(40, 52, 32, 32)
    opt = L.OptScalars()
    args = lambda go, batch, wsf: (a.handle, go, batch, fake(1), wsf, 40, 52, None, fake(2), fake(3), fake(4), 0.5, 1.0, fake(5), 0,
                                   fake(6), fake(7), fake(8), C.byref(opt), fake(9), fake(10), 1, None)
    assert lib.advx_collect_update(*args(None, 1, a.workspace_floats)) == E_BADARG
    assert lib.advx_collect_update(*args(fake(0), 0, a.workspace_floats)) == E_BADARG
    assert lib.advx_collect_update(*args(fake(0), 1, 1)) == E_SHAPE and b"workspace" in lib.advx_last_error()
    assert lib.advx_collect_update_supported(None, 40, 52, None) == 0
    # scratch of the cross entropy's forward
    assert lib.advx_ce_scratch_floats(0, 32000, 1) == 0 and lib.advx_ce_scratch_floats(8, 0, 1) == 0
    assert lib.advx_ce_scratch_floats(512, 32064, 1) == 2 * 512 * 2 + 4          # 32 064 halfs: two chunks of 16 384
    assert lib.advx_ce_scratch_floats(512, 32064, 0) == 2 * 512 * 4 + 4          # ... four of 8 192 floats
    assert lib.advx_ce_scratch_floats(512, 152064, 2) == 2 * 512 * 10 + 4        # Qwen2-VL
    assert lib.advx_ce_scratch_floats(3, 5, 2) == 2 * 3 + 4
    tg = (C.c_int64 * 4)(0, 1, 2, 3)
    fwd = lambda logits, io, T, rows, V, scratch: lib.advx_ce_fwd(logits, io, 6 * V, V, T, tg, rows, V, fake(1), fake(2), fake(3),
                                                                  scratch, None)
    assert fwd(None, 1, 2, 4, 100, fake(4)) == E_BADARG
    assert fwd(fake(0), 1, 2, 4, 100, None) == E_BADARG
    assert fwd(fake(0), 7, 2, 4, 100, fake(4)) == E_BADARG and b"io_dtype" in lib.advx_last_error()
    assert fwd(fake(0), 1, 3, 4, 100, fake(4)) == E_SHAPE                        # rows no multiple of T
    assert fwd(fake(0), 1, 2, 4, 0, fake(4)) == E_SHAPE
    assert fwd(fake(0), 1, 2, 4, 100, C.c_void_p(0x1004)) == E_BADARG and b"aligned" in lib.advx_last_error()
    bwd = lambda T, K, rows, V: lib.advx_ce_bwd(fake(0), 1, K * V, V, T, K, tg, rows, V, fake(1), fake(2), fake(3), fake(4), None)
    assert bwd(2, 1, 4, 100) == E_SHAPE                                          # fewer kept than supervised positions
    assert bwd(3, 3, 4, 100) == E_SHAPE
    assert lib.advx_ce_bwd(fake(0), 1, 300, 100, 2, 3, tg, 4, 100, None, fake(2), fake(3), fake(4), None) == E_BADARG


def test_plain_c_host_compiles_and_links_against_the_header(tmp_path):
    """CPU tier: tests/cabi/pair_steps.c - a host that is neither Python nor torch - builds with gcc against include/advx.h and
    links libadvx_hip.so (it RUNS in the GPU tier, tests/test_gpu_cabi_c_host.py): the header is plain C and every symbol it
    uses is exported."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc or the HIP headers are not here")
    from adversarialvlm_amd.build import build_library
    lib_dir = os.path.dirname(build_library())
    exe = str(tmp_path / "pair_steps")
    res = subprocess.run(["gcc", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cabi", "pair_steps.c"),
                          "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + lib_dir, "-ladvx_hip",
                          "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert os.path.exists(exe)


def test_build_sees_an_edit_to_every_header(tmp_path, monkeypatch):
    """CPU tier: `_stale()` follows every header advx.hip is compiled from - the list is found (glob), not written
    down, and each `#include "…"` of the sources is in it.  Checked on copies: the tree's own files are not touched."""
    import re
    import shutil
    from adversarialvlm_amd import build
    names = {os.path.basename(h) for h in build.headers()}
    for src in build.SOURCES + build.headers():
        with open(src) as f:
            for inc in re.findall(r'#include\s+"([^"]+)"', f.read()):
                assert os.path.basename(inc) in names, (src, inc)
    csrc, inc_dir = tmp_path / "pkg" / "csrc", tmp_path / "include"
    shutil.copytree(build.CSRC, csrc)
    inc_dir.mkdir()
    shutil.copy(build.PUBLIC_HEADER, inc_dir / "advx.h")
    out = tmp_path / "pkg" / "libadvx_hip.so"
    out.write_bytes(b"")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    monkeypatch.setattr(build, "SOURCES", [str(csrc / "advx.hip")])
    monkeypatch.setattr(build, "PUBLIC_HEADER", str(inc_dir / "advx.h"))
    monkeypatch.setattr(build, "OUT", str(out))
    files = build.SOURCES + build.headers()
    assert len(files) >= 9 and all(str(tmp_path) in f for f in files)
    t_lib = os.path.getmtime(out)
    for f in files:
        os.utime(f, (t_lib - 100, t_lib - 100))
    assert not build._stale()
    for f in files:
        os.utime(f, (t_lib + 100, t_lib + 100))
        assert build._stale(), f
        os.utime(f, (t_lib - 100, t_lib - 100))
    assert not build._stale()
