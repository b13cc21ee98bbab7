"""CPU: the kernels' per-pixel arithmetic (csrc/sgr_math.h compiled for the host,
tests/host_emul/emul.cpp) against the golden fixtures and the fp64 oracle.

This checks the hand-derived forward and adjoint math without a GPU; the `-m gpu`
tests check the same quantities through the real kernels and the C ABI."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_l2, rel_max
from inverserenderingofindoorscene_amd import tables

EMUL_DIR = os.path.join(ROOT, "tests", "host_emul")
NAMES = ("albedo", "normal", "rough", "axis", "lamb", "weight")
FP = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(EMUL_DIR, "libemul.so")
    src = os.path.join(EMUL_DIR, "emul.cpp")
    hdr = os.path.join(ROOT, "inverserenderingofindoorscene_amd", "csrc", "sgr_math.h")
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so])
    return ctypes.CDLL(so)


def _p(a):
    return None if a is None else a.ctypes.data_as(FP)


def _inputs(z):
    return {k: np.ascontiguousarray(z["in_" + k], dtype=np.float32) for k in NAMES}


def _forward(emul, z, cfg):
    x = _inputs(z)
    bn, K, R, C, eh, ew = cfg["bn"], cfg["K"], cfg["R"], cfg["C"], cfg["eh"], cfg["ew"]
    J = eh * ew
    dirs = tables.packed_direction_table(eh, ew)
    view = tables.view_vectors(C, R, cfg["fov"], cfg["cam"])
    env = np.empty((bn, 3, R, C, eh, ew), np.float32)
    d = np.empty((bn, 3, R, C), np.float32)
    s = np.empty((bn, 3, R, C), np.float32)
    emul.emul_fused_fwd(_p(x["albedo"]), _p(x["normal"]), _p(x["rough"]), _p(x["axis"]), _p(x["lamb"]), _p(x["weight"]),
                        _p(dirs), _p(view), _p(env), _p(d), _p(s), bn, K, R, C, J, cfg["imH"], cfg["imW"],
                        ctypes.c_float(cfg["F0"]), 1)
    return x, dirs, view, env, d, s


def _sep_tables(eh, ew):
    packed = tables.packed_direction_table(eh, ew)
    jpad = (eh * ew + 31) // 32 * 32
    ehp = (eh + 1) // 2 * 2
    rows = np.ascontiguousarray(packed[4 * jpad:4 * jpad + 8 * ehp])
    raw = packed[4 * jpad + 8 * ehp:]
    half = ew // 2
    cols = np.zeros((half, 8), np.float32)          # emulation layout: (ca, sa, ca^2, 2 ca sa, sa^2, 0, 0, 0) per azimuth
    cols[:, 0:2] = raw[:2 * half].reshape(half, 2)
    cols[:, 2:5] = raw[ew:ew + 4 * half].reshape(half, 4)[:, :3]
    return rows, np.ascontiguousarray(cols.reshape(-1))


FAST_CASES = ["g1_q4_k12", "g3_edges", "g10_cam_f0_q4", "g10_cam_f0_q1"]          # envWidth 16: the separable fast path applies


@pytest.mark.parametrize("name", FAST_CASES)
def test_fast_path_math_vs_golden(emul, name):
    """csrc/sgr_fast.inl's arithmetic (separable table, folded lambda, local-frame microfacet terms,
    A/Z form of the axis gradient) against the reference outputs."""
    from conftest import load_golden
    z, cfg = load_golden(name)
    x = _inputs(z)
    bn, K, R, C, eh, ew = cfg["bn"], cfg["K"], cfg["R"], cfg["C"], cfg["eh"], cfg["ew"]
    rows, cols = _sep_tables(eh, ew)
    view = tables.view_vectors(C, R, cfg["fov"], cfg["cam"])
    env = np.empty((bn, 3, R, C, eh, ew), np.float32)
    d = np.empty((bn, 3, R, C), np.float32)
    s = np.empty((bn, 3, R, C), np.float32)
    emul.emul_fast_fwd(_p(x["albedo"]), _p(x["normal"]), _p(x["rough"]), _p(x["axis"]), _p(x["lamb"]), _p(x["weight"]),
                       _p(rows), _p(cols), _p(view), _p(env), _p(d), _p(s), bn, K, R, C, eh, ew, cfg["imH"], cfg["imW"],
                       ctypes.c_float(cfg["F0"]), 1)
    for k, got in (("env", env), ("diffuse", d), ("spec", s)):
        e_ref = rel_l2(z["ref32_" + k], z["ref64_" + k])
        assert rel_l2(got, z["ref32_" + k]) < 1e-4, (name, k, rel_l2(got, z["ref32_" + k]))
        assert rel_max(got, z["ref32_" + k]) < 2e-4, (name, k, rel_max(got, z["ref32_" + k]))
        assert rel_l2(got, z["ref64_" + k]) < max(3 * e_ref, 3e-5), (name, k, rel_l2(got, z["ref64_" + k]), e_ref)
    ga, gl, gw = np.empty_like(x["axis"]), np.empty_like(x["lamb"]), np.empty_like(x["weight"])
    ct_env, ct_d, ct_s = (np.ascontiguousarray(z[k]) for k in ("ct_env", "ct_d", "ct_s"))
    emul.emul_fast_sg_bwd(_p(ct_env), _p(ct_d), _p(ct_s), _p(x["albedo"]), _p(x["normal"]), _p(x["rough"]), _p(x["axis"]),
                          _p(x["lamb"]), _p(x["weight"]), _p(rows), _p(cols), _p(view), _p(ga), _p(gl), _p(gw),
                          bn, K, R, C, eh, ew, cfg["imH"], cfg["imW"], ctypes.c_float(cfg["F0"]), 1)
    for k, got in (("axis", ga), ("lamb", gl), ("weight", gw)):
        e_ref = rel_l2(z["ref32_glin_" + k], z["ref64_glin_" + k])
        e = rel_l2(got, z["ref64_glin_" + k])
        assert e < max(3 * e_ref, 3e-5), (name, k, e, e_ref)


def test_premap_accuracy(emul):
    """tan(pi/2*0.999*x) of the kernels vs float64 on the whole decoder range, incl. x == 1."""
    x = np.concatenate([np.linspace(0, 1, 200001), 1 - np.logspace(-7, -1, 4001), [0.0, 1.0]]).astype(np.float32)
    y = np.empty_like(x)
    emul.emul_premap(_p(x), _p(y), x.size)
    # the reference's fp32 argument rounding, then exact tan
    arg = (np.float32(0.999) * x).astype(np.float32) * np.float32(np.pi / 2)
    truth = np.tan(arg.astype(np.float64))
    rel = np.abs(y - truth) / np.maximum(np.abs(truth), 1e-30)
    assert rel.max() < 4e-7, rel.max()
    ref = torch.tan(np.pi / 2 * (0.999 * torch.from_numpy(x))).numpy()      # torch's own fp32 tan
    assert (np.abs(y - ref) / np.maximum(np.abs(ref), 1e-30)).max() < 5e-7


def test_forward_math_vs_golden(emul, golden):
    name, z, cfg = golden
    _, _, _, env, d, s = _forward(emul, z, cfg)
    for k, got in (("env", env), ("diffuse", d), ("spec", s)):
        e_ref = rel_l2(z["ref32_" + k], z["ref64_" + k])       # the reference's own fp32 noise
        assert rel_l2(got, z["ref32_" + k]) < 1e-4, (name, k, rel_l2(got, z["ref32_" + k]))
        assert rel_max(got, z["ref32_" + k]) < 2e-4, (name, k)
        # vs the fp64 arbiter: no worse than ~the reference's own fp32 noise (the spec term is
        # ill-conditioned where alpha^2 is tiny; both fp32 evaluations sit on that floor)
        assert rel_l2(got, z["ref64_" + k]) < max(3 * e_ref, 3e-5), (name, k, rel_l2(got, z["ref64_" + k]), e_ref)


def test_sg_backward_math_vs_golden(emul, golden):
    name, z, cfg = golden
    x, dirs, view, _, _, _ = _forward(emul, z, cfg)
    bn, K, R, C, J = cfg["bn"], cfg["K"], cfg["R"], cfg["C"], cfg["eh"] * cfg["ew"]
    ga, gl, gw = np.empty_like(x["axis"]), np.empty_like(x["lamb"]), np.empty_like(x["weight"])
    ct_env, ct_d, ct_s = (np.ascontiguousarray(z[k]) for k in ("ct_env", "ct_d", "ct_s"))
    emul.emul_sg_bwd(_p(ct_env), _p(ct_d), _p(ct_s), _p(x["albedo"]), _p(x["normal"]), _p(x["rough"]), _p(x["axis"]),
                     _p(x["lamb"]), _p(x["weight"]), _p(dirs), _p(view), _p(ga), _p(gl), _p(gw), bn, K, R, C, J,
                     cfg["imH"], cfg["imW"], ctypes.c_float(cfg["F0"]), 1)
    for k, got in (("axis", ga), ("lamb", gl), ("weight", gw)):
        e_ref = rel_l2(z["ref32_glin_" + k], z["ref64_glin_" + k])
        e = rel_l2(got, z["ref64_glin_" + k])
        assert e < max(2 * e_ref, 1e-5), (name, k, e, e_ref)
        assert rel_l2(got, z["ref32_glin_" + k]) < max(3 * e_ref, 1e-4), (name, k)


def _nondegenerate_mask(z, cfg):
    """Pixels where the reference's normal gradient is meaningful: away from N || up (the local
    frame is singular there and torch returns O(1e20) values) and away from |N|^2 clamps."""
    n = torch.from_numpy(z["in_normal"])
    R, C = cfg["R"], cfg["C"]
    pn = torch.nn.functional.adaptive_avg_pool2d(n, (R, C))
    nn = (pn * pn).sum(1, keepdim=True)
    un = pn / nn.clamp(1e-6, 1).sqrt()
    ok = ((un[:, 1:2].abs() < 0.999) & (nn > 1e-4)).float()
    return torch.nn.functional.interpolate(ok, size=(cfg["imH"], cfg["imW"]), mode="nearest").numpy() > 0.5


def test_brdf_backward_math_vs_golden(emul, golden):
    name, z, cfg = golden
    x, dirs, view, env, _, _ = _forward(emul, z, cfg)
    bn, R, C, J = cfg["bn"], cfg["R"], cfg["C"], cfg["eh"] * cfg["ew"]
    gA, gN, gR = np.empty_like(x["albedo"]), np.empty_like(x["normal"]), np.empty_like(x["rough"])
    ct_d, ct_s = np.ascontiguousarray(z["ct_d"]), np.ascontiguousarray(z["ct_s"])
    env_ref = np.ascontiguousarray(z["ref32_env"])
    emul.emul_brdf_bwd(_p(ct_d), _p(ct_s), _p(x["albedo"]), _p(x["normal"]), _p(x["rough"]), _p(env_ref), _p(dirs),
                       _p(view), _p(gA), _p(gN), _p(gR), bn, R, C, J, cfg["imH"], cfg["imW"], ctypes.c_float(cfg["F0"]))
    mask = _nondegenerate_mask(z, cfg)
    for k, got, m in (("albedo", gA, np.ones_like(gA, bool)), ("normal", gN, np.broadcast_to(mask, gN.shape)),
                      ("rough", gR, np.ones_like(gR, bool))):
        ref64, ref32 = z["ref64_glin_" + k], z["ref32_glin_" + k]
        assert np.isfinite(got).all(), (name, k)
        e_ref = rel_l2(ref32[m], ref64[m])
        assert rel_l2(got[m], ref32[m]) < 3e-4, (name, k, rel_l2(got[m], ref32[m]))    # incl. the |N|^2 clamp kink
        if e_ref < 1e-3:
            e = rel_l2(got[m], ref64[m])
            assert e < max(3 * e_ref, 2e-5), (name, k, e, e_ref)


def test_c_tables_match_numpy():
    """sgr_fill_* helpers of the C ABI vs the numpy tables (needs libsgrender.so, no GPU)."""
    from inverserenderingofindoorscene_amd import _lib
    lib = _lib.load()
    for eh, ew in [(8, 16), (16, 32), (4, 8), (3, 5)]:
        want = tables.packed_direction_table(eh, ew)
        assert lib.sgr_dirs_floats(eh, ew) == want.size
        got = np.empty_like(want)
        assert lib.sgr_fill_direction_table(got.ctypes.data, eh, ew) == 0
        assert np.abs(got - want).max() <= 1.2e-7
    for R, C, fov in [(120, 160, 57.0), (6, 8, 42.75), (1, 1, 57.0)]:
        want = tables.view_vectors(C, R, fov)
        got = np.empty_like(want)
        assert lib.sgr_fill_view_vectors(got.ctypes.data, R, C, ctypes.c_float(fov), None) == 0
        assert np.abs(got - want).max() <= 1.2e-7
    # a camera off the origin (constructor kwarg cameraPos, models.py:408,428-430): the C-side table's camera branch
    for R, C, fov, cam in [(120, 160, 57.0, (0.1, -0.05, 0.2)), (8, 12, 42.75, (0.1, -0.05, 0.2)), (6, 8, 57.0, (-0.3, 0.25, -0.5))]:
        want = tables.view_vectors(C, R, fov, cam)
        assert np.abs(want - tables.view_vectors(C, R, fov)).max() > 1e-2      # the camera does move the table
        got = np.empty_like(want)
        cam32 = np.asarray(cam, np.float32)
        assert lib.sgr_fill_view_vectors(got.ctypes.data, R, C, ctypes.c_float(fov), cam32.ctypes.data) == 0
        assert np.abs(got - want).max() <= 1.2e-7


@pytest.mark.parametrize("name", ["g10_cam_f0_q4", "g10_cam_f0_q1"])
def test_view_table_with_camera_equals_the_reference_bits(name):
    """Fixture g10 carries the reference's own `renderingLayer(cameraPos=[0.1,-0.05,0.2], fov=42.75).v` (oracle/make_golden.py: rl_view):
    the product's host table and the oracle's are the same bits."""
    from conftest import load_golden
    from oracle import sg_oracle as O
    z, cfg = load_golden(name)
    assert cfg["cam"] == [0.1, -0.05, 0.2] and cfg["F0"] == 0.04
    ref = z["ref_view"]
    assert np.array_equal(tables.view_vectors(cfg["C"], cfg["R"], cfg["fov"], cfg["cam"]), ref)
    assert np.array_equal(O.view_vectors(cfg["C"], cfg["R"], cfg["fov"], cfg["cam"]), ref)
    assert not np.array_equal(tables.view_vectors(cfg["C"], cfg["R"], cfg["fov"]), ref)


# --------------------------------------------------------------------------- #
# decoder heads as the kernels' prologue (premap = 3): the formulas of heads_two_lobes / heads_bwd_lobe (csrc/sgr_pk.inl)
# --------------------------------------------------------------------------- #
def _tanh_abs32(x):
    """1 - 2 / (e^{2x} + 1) in fp32 (v_exp_f32 + v_rcp_f32 on the device): the prologue's tanh."""
    f = np.float32
    with np.errstate(over="ignore"):
        e = np.exp2(x.astype(f) * f(2.8853900817779268)).astype(f)
        r = (f(1.0) / (e + f(1.0))).astype(f)
    return (f(1.0) - f(2.0) * r).astype(f)


def test_heads_prologue_formulas_against_the_oracle():
    """The prologue's restatement of models.py:336-346 -- tanh as 1 - 2/(e^{2x}+1), a / max(|a|, 1e-6) as a * min(rsqrt(|a|^2), 1e6),
    the [0, 1] clamp behind torch's op-by-op rounding -- and its chain rule, in fp32 numpy against the fp64 oracle and its autograd:
    absolute error a few 1e-7 forward (what the tan pre-map then sees is within an ulp or two of the standalone pass), gradients to
    1e-5 of their norm."""
    from oracle import sg_oracle as O
    f = np.float32
    rng = np.random.default_rng(5)
    bn, K, R, C = 2, 6, 5, 7
    xa = (1.5 * rng.standard_normal((bn, 3 * K, R, C))).astype(f)
    xl = (1.5 * rng.standard_normal((bn, K, R, C))).astype(f)
    xw = (1.5 * rng.standard_normal((bn, 3 * K, R, C))).astype(f)
    for x in (xa, xl, xw):                                   # saturated entries either side, and huge ones (e^{2x} = inf / 0)
        flat = x.reshape(-1)
        flat[::17] = 7.0
        flat[5::19] = -7.0
        flat[3::101] = 60.0
        flat[7::103] = -60.0
    t64 = [torch.from_numpy(v).double().requires_grad_(True) for v in (xa, xl, xw)]
    a64, l64, w64, _ = O.light_heads(*t64)

    # forward
    ta = _tanh_abs32(xa).reshape(bn, K, 3, R, C)
    a = (f(1.01) * ta).astype(f)
    n2 = (a * a).sum(axis=2, keepdims=True, dtype=f)
    with np.errstate(divide="ignore"):
        inv = np.minimum((f(1.0) / np.sqrt(n2, dtype=f)).astype(f), f(1e6))
    y = (a * inv).astype(f)
    unit = lambda t: np.clip((f(0.5) * ((f(1.01) * t).astype(f) + f(1.0)).astype(f)).astype(f), f(0.0), f(1.0))
    lam, w = unit(_tanh_abs32(xl)), unit(_tanh_abs32(xw))
    assert np.abs(y - a64.detach().numpy()).max() < 5e-7
    assert np.abs(lam - l64.detach().numpy()).max() < 3e-7 and np.abs(w - w64.detach().numpy()).max() < 3e-7
    assert np.isfinite(y).all() and np.isfinite(lam).all() and np.isfinite(w).all()

    # chain rule (heads_bwd_lobe) against autograd through the oracle
    ga = rng.standard_normal(a64.shape).astype(f)
    gl = rng.standard_normal(l64.shape).astype(f)
    gw = rng.standard_normal(w64.shape).astype(f)
    tot = (a64 * torch.from_numpy(ga).double()).sum() + (l64 * torch.from_numpy(gl).double()).sum() + (w64 * torch.from_numpy(gw).double()).sum()
    r64 = torch.autograd.grad(tot, t64)
    s_a = (f(1.0) - ta * ta).astype(f)
    live = n2 >= f(1e-12)
    invb = np.where(live, (f(1.0) / np.sqrt(np.maximum(n2, f(1e-30)), dtype=f)).astype(f), f(1e6))
    dot = np.where(live, (a * ga).sum(axis=2, keepdims=True, dtype=f) * invb * invb, f(0.0))
    gxa = ((ga - a * dot) * invb * (f(1.01) * s_a)).astype(f).reshape(bn, 3 * K, R, C)

    def unit_bwd(x, g):
        t = _tanh_abs32(x)
        pre = (f(0.5) * ((f(1.01) * t).astype(f) + f(1.0)).astype(f)).astype(f)
        return np.where((pre >= 0) & (pre <= 1), g * (f(0.505) * (f(1.0) - t * t)), f(0.0)).astype(f)
    gxl, gxw = unit_bwd(xl, gl), unit_bwd(xw, gw)
    for name, mine, ref in (("x_axis", gxa, r64[0]), ("x_lamb", gxl, r64[1]), ("x_weight", gxw, r64[2])):
        assert rel_l2(mine, ref.numpy()) < 1e-5, (name, rel_l2(mine, ref.numpy()))
