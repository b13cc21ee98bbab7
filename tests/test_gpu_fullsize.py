"""GPU: parity AT SIZE, whole images.  BASELINE config 2 (batch 16, 240x320 -> 120x160, SGNum 12, 8x16): ALL sixteen images,
every env cell, forward and backward w.r.t. the SG parameters AND the BRDF maps, against the fp64 oracle; one image against the
reference-made fixture g7_cfg2_one_image.npz (oracle/make_golden_fullsize.py: the unmodified reference in fp32 and fp64), which
supplies the reference's own fp32 error e_ref as the yardstick; the clamp kink at size against reference fp32 (g9_ratio1_unit_normals.npz); one full image of config 5 (480x640 -> 240x320, SGNum 24,
16x32) with e_ref from the reference-made fixture g8_cfg5_small.npz (oracle/make_golden_cfg5.py), which is also compared
against directly; and a fixed-seed randomised shape sweep.  Every tolerance is ``max(2 e_ref, 1e-4)`` (BASELINE.md section 3 /
north_star), capped at 2e-4 for the normal / roughness gradients, whose reference fp32 evaluation is itself only good to 1e-3
at size (the adjoint here evaluates the GGX denominator without the reference's cancellation, sgr_math.h: brdf_dir_bwd).

The GPU runs the full batch through the HIP kernels; the fp64 oracle (device-generic torch) is evaluated ON THE GPU as well, image
by image: about a second per image of config 2 instead of twenty on the host, so nothing is windowed any more."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, NAMES6, oracle_fwd_bwd, oracle_with_noise, rel_l2, rel_max, scalar_close, tol2

pytestmark = pytest.mark.gpu
NAMES = NAMES6
SG = ("axis", "lamb", "weight")
BRDF = ("albedo", "normal", "rough")


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


def _fixture_with_eref(name):
    z = np.load(os.path.join(GOLDEN_DIR, name))
    cfg = {k: v for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist())}
    for k in ("bn", "imH", "imW", "R", "C", "K", "eh", "ew", "seed"):
        cfg[k] = int(cfg[k])
    keys = ("env", "diffuse", "spec") + tuple(f"glin_{k}" for k in SG + BRDF)
    e_ref = {k: rel_l2(z["ref32_" + k], z["ref64_" + k]) for k in keys}
    return z, cfg, e_ref


@pytest.fixture(scope="module")
def g7():
    return _fixture_with_eref("g7_cfg2_one_image.npz")


@pytest.fixture(scope="module")
def g8():
    return _fixture_with_eref("g8_cfg5_small.npz")


def _regenerated_inputs(O, z, cfg):
    """The inputs and cotangents of a results-only fixture (g7, g8, g9), regenerated from the seed out of NumPy's frozen legacy stream
    (oracle.synthetic_inputs_np) -- and the stored checksums HOLD: rounds 3-4 drew them from torch's CPU generator and skipped the
    comparison when a machine's stream differed, which would have switched the strongest tests off silently after a torch update."""
    inp = O.synthetic_inputs_np(cfg["bn"], cfg["imH"], cfg["imW"], cfg["R"], cfg["C"], cfg["K"], cfg["eh"], cfg["ew"], seed=cfg["seed"])
    chk = np.array([inp[k].double().sum().item() for k in NAMES])
    assert np.allclose(chk, z["in_checksums"], rtol=1e-12), ("regenerated inputs differ from the ones the fixture was made with", chk, z["in_checksums"])
    cts = O.synthetic_cotangents_np(cfg["bn"], cfg["R"], cfg["C"], cfg["eh"], cfg["ew"], cfg["seed"] + 7)
    assert np.allclose([c.double().sum().item() for c in cts], z["ct_checksums"], rtol=1e-12)
    return inp, cts


def _fwd_bwd(sgr, inp, cts, R, C, eh=8, ew=16, wrt=SG):
    x = {k: inp[k].cuda() for k in NAMES}
    for k in wrt:
        x[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    grads = torch.autograd.grad([env, d, s], [x[k] for k in wrt], grad_outputs=[c.cuda() for c in cts])
    return env.detach(), d.detach(), s.detach(), dict(zip(wrt, grads))


def _regular_normals(inp, b, R, C):
    """Mask (image resolution) of the pixels whose pooled normal is not (anti)parallel to `up` and not near zero: there the local
    frame is singular and the reference's normal gradient is O(1e20) garbage -- excluded, as in the
    fixture tests."""
    n = inp["normal"][b:b + 1]
    pn = torch.nn.functional.adaptive_avg_pool2d(n, (R, C))
    nn = (pn * pn).sum(1, keepdim=True)
    un = pn / nn.clamp(1e-6, 1).sqrt()
    ok = ((un[:, 1:2].abs() < 0.999) & (nn > 1e-4)).float()
    return torch.nn.functional.interpolate(ok, size=tuple(n.shape[2:]), mode="nearest") > 0.5


def test_one_image_vs_reference_fixture(sgr, g7):
    """The reference itself at full size: fp32 values, and fp64 values as the arbiter; SG and BRDF-map gradients."""
    from oracle import sg_oracle as O
    z, cfg, e_ref = g7
    inp, cts = _regenerated_inputs(O, z, cfg)
    R, C, eh, ew = cfg["R"], cfg["C"], cfg["eh"], cfg["ew"]
    env, d, s, grads = _fwd_bwd(sgr, inp, cts, R, C, eh, ew, wrt=SG + BRDF)
    se, ss = [int(v) for v in z["strides"]]
    got = dict(env=env[:, :, ::se, ::se], diffuse=d, spec=s, **{f"glin_{k}": gk[..., ::ss, ::ss] for k, gk in grads.items()})
    report = {}
    for k, v in got.items():
        v = v.cpu()
        e32, e64 = rel_l2(v, z["ref32_" + k]), rel_l2(v, z["ref64_" + k])
        report[k] = (e32, e64, e_ref[k])
        loose = k in ("glin_normal", "glin_rough")      # the reference's own fp32 gradients are good to 1e-3 there
        assert e64 <= (2.0 * e_ref[k] if loose else tol2(e_ref[k], 1e-5)), (k, "vs reference fp64", e64, e_ref[k])
        assert e32 <= (3.0 * e_ref[k] if loose else 1e-4), (k, "vs reference fp32", e32, e_ref[k])
    print("config 2, one image vs the reference-made fixture: (vs ref32, vs ref64, e_ref)", {k: tuple(f"{x:.2e}" for x in v) for k, v in report.items()})
    # max-abs / max|ref| against reference fp32 (<= 2e-4) -- or twice the reference's OWN max-norm error against its fp64 run where
    # that is larger: on this image the reference's fp32 specular term is 2.7e-4 off its fp64 one in the max norm, the kernel 6e-5
    for k, v in (("diffuse", d), ("spec", s)):
        own = rel_max(z["ref32_" + k], z["ref64_" + k])
        # the widened bound belongs to the fp32 comparison ALONE (a distance between two fp32 evaluations, each `own` from the
        # truth); against the reference's fp64 run SURVEY 8c's 2e-4 stands as written -- the kernel is at ~6e-5 there
        assert rel_max(v.cpu(), z["ref32_" + k]) <= max(2e-4, 2.0 * own), (k, rel_max(v.cpu(), z["ref32_" + k]), own)
        assert rel_max(v.cpu(), z["ref64_" + k]) <= 2e-4, (k, rel_max(v.cpu(), z["ref64_" + k]), own)
    assert abs(env.double().norm().item() - float(z["ref32_env_norm"][0])) < 1e-5 * float(z["ref32_env_norm"][0])
    for k in SG:
        n = float(z[f"ref32_glin_{k}_norm"][0])
        assert abs(grads[k].double().norm().item() - n) < 1e-4 * n, k


def test_clamp_kink_at_size_vs_reference_fp32(sgr):
    """Fixture g9 (round 5; oracle/make_golden_fullsize.py): BRDF maps AT the env-grid resolution (120x160, no pooling) with unit input
    normals -- every pixel sits on the |N|^2 == 1 kink of the two-sided clamp (models.py:467-468), where the reference's fp32 and fp64
    normal gradients differ by O(1) (0.99 rel-L2 over this image) and fp64 is no arbiter.  The reference's fp32 VALUES are the semantics:
    normal / roughness gradients against reference fp32 over ALL pixels within max(3 e_ref, 1e-4), e_ref = the reference's own fp32-vs-fp64
    error on the pixels where its two runs take the same clamp branch (`agree`, 66 % of the image: 4.6e-4 / 2.8e-4) -- and against reference
    fp64 on those pixels within 2 e_ref.  (Rounds 3-4 held this case to a bare 2e-3 against the fp32 ORACLE on small random shapes.)"""
    from oracle import sg_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, "g9_ratio1_unit_normals.npz"))
    cfg = {k: v for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist())}
    for k in ("bn", "imH", "imW", "R", "C", "K", "eh", "ew", "seed"):
        cfg[k] = int(cfg[k])
    inp, cts = _regenerated_inputs(O, z, cfg)
    R, C, eh, ew = cfg["R"], cfg["C"], cfg["eh"], cfg["ew"]
    assert (cfg["imH"], cfg["imW"]) == (R, C)
    env, d, s, grads = _fwd_bwd(sgr, inp, cts, R, C, eh, ew, wrt=SG + BRDF)
    se, ss = [int(v) for v in z["strides"]]
    agree = torch.from_numpy(z["agree"])
    regular = _regular_normals(inp, 0, R, C)                       # N parallel to up: the reference's gradient is O(1e20) garbage there
    report = {}
    # values and the gradients the kink does not touch: the usual bounds
    got = dict(env=env[:, :, ::se, ::se], diffuse=d, spec=s, glin_albedo=grads["albedo"], **{f"glin_{k}": grads[k][..., ::ss, ::ss] for k in SG})
    for k, v in got.items():
        r32, r64 = torch.from_numpy(z["ref32_" + k]), torch.from_numpy(z["ref64_" + k])
        e_ref = rel_l2(r32, r64)
        e32, e64 = rel_l2(v.cpu(), r32), rel_l2(v.cpu(), r64)
        report[k] = (e32, e64, e_ref)
        # vs reference fp64 within twice the reference's own fp32 error; vs reference fp32 within 1e-4 -- or that same 2 e_ref where the
        # reference's fp32 run is itself farther than 5e-5 from its fp64 one (the specular term of this image: 1.8e-4)
        assert e64 <= tol2(e_ref, 1e-5) and e32 <= tol2(e_ref), (k, e32, e64, e_ref)
    for k in ("normal", "rough"):
        a = grads[k].cpu()
        r32, r64 = torch.from_numpy(z[f"ref32_glin_{k}"]), torch.from_numpy(z[f"ref64_glin_{k}"])
        m_all = regular.expand_as(a)
        m_agree = (regular & agree).expand_as(a)
        e_ref = rel_l2(r32[m_agree], r64[m_agree])                  # the reference's own fp32 error where fp64 is an arbiter
        e32_all = rel_l2(a[m_all], r32[m_all])
        e64_agree = rel_l2(a[m_agree], r64[m_agree])
        report["glin_" + k] = (e32_all, e64_agree, e_ref, rel_l2(r32[m_all], r64[m_all]))
        assert e32_all <= max(3.0 * e_ref, 1e-4), (k, "vs reference fp32, all pixels", e32_all, e_ref)
        assert e64_agree <= max(2.0 * e_ref, 2e-4), (k, "vs reference fp64 where the clamp branches agree", e64_agree, e_ref)
    print("clamp kink at size (g9): (vs ref32, vs ref64 [agreeing pixels for normal / rough], e_ref[, ref32 vs ref64 over all pixels])",
          {k: tuple(f"{x:.2e}" for x in v) for k, v in report.items()})


def test_config2_all_sixteen_images_forward_backward(sgr, g7):
    """Every image of the contract batch, every cell, against the fp64 oracle evaluated on the GPU; SG and BRDF-map gradients;
    tolerances from the reference's own fp32 error at this size (fixture g7)."""
    from oracle import sg_oracle as O
    _, _, e_ref = g7
    bn, imH, imW, R, C, K = 16, 240, 320, 120, 160, 12
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, seed=20202)
    g = torch.Generator().manual_seed(11)
    cts = [torch.randn((bn, 3, R, C, 8, 16), generator=g), torch.randn((bn, 3, R, C), generator=g), torch.randn((bn, 3, R, C), generator=g)]
    env, d, s, grads = _fwd_bwd(sgr, inp, cts, R, C, wrt=SG + BRDF)
    assert all(torch.isfinite(t).all() for t in [env, d, s] + list(grads.values()))
    worst = {}
    for b in range(bn):
        ref = oracle_fwd_bwd(O, inp, cts, 8, 16, SG + BRDF, torch.float64, "cuda", b)
        ok_n = _regular_normals(inp, b, R, C).cuda().expand(1, 3, imH, imW)
        errs = dict(env=rel_l2(env[b:b + 1], ref["env"]), diffuse=rel_l2(d[b:b + 1], ref["diffuse"]), spec=rel_l2(s[b:b + 1], ref["spec"]))
        for k in SG + BRDF:
            a, r = grads[k][b:b + 1], ref["g_" + k]
            errs["glin_" + k] = rel_l2(a[ok_n], r[ok_n]) if k == "normal" else rel_l2(a, r)
        for k, e in errs.items():
            lim = min(tol2(e_ref[k]), 2e-4)      # normal / roughness: 2 e_ref would be 2e-3 (the reference's fp32 gradients carry 1e-3)
            assert e <= lim, (b, k, e, e_ref[k])
            worst[k] = max(worst.get(k, 0.0), e)
        del ref
    print("config 2, worst rel-L2 over 16 WHOLE images vs the fp64 oracle:", {k: f"{v:.2e}" for k, v in worst.items()},
          "reference's own fp32 error (g7):", {k: f"{v:.2e}" for k, v in e_ref.items()})


def test_config5_small_vs_reference_fixture(sgr, g8):
    """SGNum 24 on the 16x32 grid against the unmodified reference (12 x 16 cells): values and all six gradients."""
    from oracle import sg_oracle as O
    z, cfg, e_ref = g8
    R, C, K, eh, ew = cfg["R"], cfg["C"], cfg["K"], cfg["eh"], cfg["ew"]
    inp, cts = _regenerated_inputs(O, z, cfg)
    env, d, s, grads = _fwd_bwd(sgr, inp, cts, R, C, eh, ew, wrt=SG + BRDF)
    st = int(z["env_stride"][0])
    got = dict(env=env[:, :, ::st, ::st], diffuse=d, spec=s, **{f"glin_{k}": gk for k, gk in grads.items()})
    for k, v in got.items():
        v = v.cpu()
        e32, e64 = rel_l2(v, z["ref32_" + k]), rel_l2(v, z["ref64_" + k])
        loose = k in ("glin_normal", "glin_rough")
        assert e64 <= (2.0 * e_ref[k] if loose else tol2(e_ref[k], 2e-5)), (k, "vs reference fp64", e64, e_ref[k])
        assert e32 <= (3.0 * e_ref[k] if loose else 1e-4), (k, "vs reference fp32", e32, e_ref[k])


def test_config5_one_full_image(sgr, g8):
    """BASELINE configs[4]: 480x640 maps, env grid 240x320, SGNum 24, 16x32 directions; one whole image against
    the fp64 oracle on the GPU, SG and BRDF-map gradients; e_ref from the reference-made fixture at these parameters (g8)."""
    from oracle import sg_oracle as O
    _, _, e_ref = g8
    bn, imH, imW, R, C, K, eh, ew = 1, 480, 640, 240, 320, 24, 16, 32
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=20205)
    g = torch.Generator().manual_seed(12)
    cts = [torch.randn((bn, 3, R, C, eh, ew), generator=g), torch.randn((bn, 3, R, C), generator=g), torch.randn((bn, 3, R, C), generator=g)]
    env, d, s, grads = _fwd_bwd(sgr, inp, cts, R, C, eh, ew, wrt=SG + BRDF)
    assert all(torch.isfinite(t).all() for t in [env, d, s] + list(grads.values()))
    ref = oracle_fwd_bwd(O, inp, cts, eh, ew, SG + BRDF, torch.float64, "cuda", 0)
    ok_n = _regular_normals(inp, 0, R, C).cuda().expand(1, 3, imH, imW)
    errs = dict(env=rel_l2(env, ref["env"]), diffuse=rel_l2(d, ref["diffuse"]), spec=rel_l2(s, ref["spec"]))
    for k in SG + BRDF:
        a, r = grads[k], ref["g_" + k]
        errs["glin_" + k] = rel_l2(a[ok_n], r[ok_n]) if k == "normal" else rel_l2(a, r)
    print("config 5, one WHOLE image vs the fp64 oracle:", {k: f"{v:.2e}" for k, v in errs.items()}, "e_ref (g8):", {k: f"{v:.2e}" for k, v in e_ref.items()})
    for k, e in errs.items():
        lim = min(tol2(e_ref[k]), 2e-4)
        assert e <= lim, (k, e, e_ref[k])


def test_config5_fused_objective_one_full_image(sgr):
    """The fused light objective at config 5's size and parameters (24 lobes, 16x32: the half-wave statistics forward and the
    four-lane-group backward): values and SG gradients against the fp64 oracle on the GPU, yardstick = the same objective through
    the oracle in fp32; and against the unfused HIP pipeline."""
    from oracle import sg_oracle as O
    bn, imH, imW, R, C, K, eh, ew = 1, 480, 640, 240, 320, 24, 16, 32
    assert sgr.light_objective_supported(K, R, C, eh, ew)
    inp = O.synthetic_inputs(bn, imH, imW, R, C, K, eh, ew, seed=20205)
    ind = torch.ones(bn, 1, 1, 1)
    x = {k: v.cuda() for k, v in inp.items()}
    for k in SG:
        x[k].requires_grad_(True)
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"], x["env_gt"],
                              ind.cuda(), 1.0, 10.0)
    grads = torch.autograd.grad(obj[0], [x[k] for k in SG])

    def objective(dtype):
        xo = {k: v.to("cuda", dtype) for k, v in inp.items()}
        for k in SG:
            xo[k].requires_grad_(True)
        eo, do, so = O.render_from_sg(xo["albedo"], xo["normal"], xo["rough"], xo["axis"], xo["lamb"], xo["weight"], eh, ew)
        ro, _, _, _ = O.render_loss(do, so, xo["im"], xo["seg"], R, C)
        co, _, _, _ = O.recon_loss(eo, xo["env_gt"], xo["seg"], ind.to("cuda", dtype), R, C)
        return ro.detach(), co.detach(), torch.autograd.grad(ro + 10.0 * co, [xo[k] for k in SG])

    ro, co, g64 = objective(torch.float64)
    ro32, co32, g32 = objective(torch.float32)
    assert scalar_close(obj[1].item(), ro.item(), ro32.item() - ro.item()), (obj[1].item(), ro.item(), ro32.item())
    assert scalar_close(obj[2].item(), co.item(), co32.item() - co.item()), (obj[2].item(), co.item(), co32.item())
    errs = {}
    for k, a, r, r32 in zip(SG, grads, g64, g32):
        assert torch.isfinite(a).all(), k
        errs[k] = (rel_l2(a, r), rel_l2(r32, r))
        assert errs[k][0] <= tol2(errs[k][1]), (k, errs[k])
    print("config 5, fused objective, one whole image: SG gradient error vs fp64 oracle / the fp32 oracle's own:", {k: tuple(f"{v:.2e}" for v in e) for k, e in errs.items()})
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    r2, _ = sgr.render_loss(d, s, x["im"], x["seg"], R, C)
    c2 = sgr.recon_loss(env, x["env_gt"], x["seg"], ind.cuda(), R, C)
    g3 = torch.autograd.grad(r2 + 10.0 * c2, [x[k] for k in SG])
    for k, ga, gb in zip(SG, grads, g3):
        assert rel_l2(ga, gb) < 1e-4, (k, rel_l2(ga, gb))


def _rand_case(g):
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())
    bn, R, C, q = ri(1, 3), ri(3, 13), ri(3, 17), (1, 2)[ri(0, 1)]
    wide = bool(ri(0, 1))      # the 16x32-style grid / up to 24 lobes in half of the cases (round 3: fused objective kernels exist for those too)
    return dict(bn=bn, R=R, C=C, q=q, K=ri(1, 24 if wide else 12), eh=ri(1, 9), ew=32 if wide else 16, benign=bool(ri(0, 1)))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_randomised_shapes_fixed_seeds(sgr, seed):
    """8 random (batch, grid, pooling ratio, lobe count, envHeight) cases per seed through the fused forward + backward and
    the fused light objective, against the fp64 oracle; tolerance max(2 e32, 1e-4) with e32 the fp32 noise of the oracle's
    restatement on the same inputs (conftest.oracle_with_noise)."""
    from oracle import sg_oracle as O
    g = torch.Generator().manual_seed(seed)
    worst = 0.0
    for case in range(8):
        c = _rand_case(g)
        bn, R, C, K, eh, ew = c["bn"], c["R"], c["C"], c["K"], c["eh"], c["ew"]
        inp = O.synthetic_inputs(bn, R * c["q"], C * c["q"], R, C, K, eh, ew, seed=1000 * (seed + 1) + case, benign=c["benign"])
        ind = (torch.rand(bn, 1, 1, 1, generator=g) < 0.8).float()
        x = {k: v.cuda() for k, v in inp.items()}
        for k in SG:
            x[k].requires_grad_(True)
        layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
        ct = [torch.randn(t.shape, generator=g) for t in (env, d, s)]
        gr = torch.autograd.grad([env, d, s], [x[k] for k in SG], grad_outputs=[t.cuda() for t in ct])
        r64, _, e32 = oracle_with_noise(O, inp, ct, eh, ew, SG, "cuda")
        errs = dict(env=rel_l2(env.detach(), r64["env"]), diffuse=rel_l2(d.detach(), r64["diffuse"]), spec=rel_l2(s.detach(), r64["spec"]),
                    **{f"g_{k}": rel_l2(a, r64["g_" + k]) for k, a in zip(SG, gr)})
        for k, e in errs.items():
            assert e <= tol2(e32[k]), (seed, case, c, k, e, e32[k])
            worst = max(worst, e / tol2(e32[k]))
        # the fused light objective: its gradient against the oracle's, yardstick = the same objective's gradient in fp32
        obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"],
                                  x["env_gt"], ind.cuda(), 1.0, 10.0)
        go2 = torch.autograd.grad(obj[0], [x[k] for k in SG])

        def objective(dtype):
            xo = {k: v.to("cuda", dtype) for k, v in inp.items()}
            for k in SG:
                xo[k].requires_grad_(True)
            eo, do, so = O.render_from_sg(xo["albedo"], xo["normal"], xo["rough"], xo["axis"], xo["lamb"], xo["weight"], eh, ew)
            ro, _, _, _ = O.render_loss(do, so, xo["im"], xo["seg"], R, C)
            co, _, _, _ = O.recon_loss(eo, xo["env_gt"], xo["seg"], ind.to("cuda", dtype), R, C)
            return ro.detach(), co.detach(), torch.autograd.grad(ro + 10.0 * co, [xo[k] for k in SG])

        ro, co, g64 = objective(torch.float64)
        ro32, co32, g32 = objective(torch.float32)
        assert scalar_close(obj[1].item(), ro.item(), ro32.item() - ro.item()), (seed, case, c, obj[1].item(), ro.item(), ro32.item())
        assert scalar_close(obj[2].item(), co.item(), co32.item() - co.item()), (seed, case, c, obj[2].item(), co.item(), co32.item())
        for k, a, r, r32 in zip(SG, go2, g64, g32):
            assert torch.isfinite(a).all(), (seed, case, c)
            e, e_o = rel_l2(a, r), rel_l2(r32, r)
            assert e <= tol2(e_o), (seed, case, c, "objective", k, e, e_o)
    print(f"seed {seed}: worst error / tolerance = {worst:.2f}")
