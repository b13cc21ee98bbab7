"""GPU: a fixed-seed randomised parity sweep (round 4's `tools/fuzz_parity.py` was a tool with a flat 5e-4 bound).

Thirty random cases -- batch 1-3, env grids 3..13 x 3..17, pooling ratio 1 or 2, 8x16-style or 16x32-style direction grids with 1..16 rows,
1..24 lobes, the decoder range (`stress`) or the benign one, some images without a ground-truth env -- through the fused layer (forward and
backward w.r.t. the SG parameters) and the fused light objective (wrapperBRDFLight.py:167-207: both loss values, the gradients, and the
forward-only route), each quantity held to ``max(2 e_ref, 1e-4)`` rel-L2 against the fp64 oracle (BASELINE.md section 3 / north_star), where
``e_ref`` is the fp32 evaluation of the same algorithm by the oracle on the same inputs -- the proxy for the reference's own fp32 error
where no reference-made fixture exists (conftest.oracle_with_noise; within 1.5x of the real thing where both exist: g1-g3, g7-g9).  Loss
values: ``|got - ref| <= max(2 |ref32 - ref64|, 1e-5 |ref|)`` (conftest.scalar_close).  Both oracles run on the GPU (device-generic torch)."""
import pytest
import torch

from conftest import oracle_with_noise, rel_l2, scalar_close, tol2

pytestmark = pytest.mark.gpu
SG = ("axis", "lamb", "weight")
N_CASES = 30
# Ratio-1 cases (q = 1: the BRDF maps AT the env-grid resolution, unit input normals): every pixel sits on the |N|^2 == 1 side of the two-sided
# clamp (models.py:467-468) with |N|^2 = 1 +- one ulp, and the GGX denominator ndh^2 (a^2 - 1) + 1 (models.py:499) is then uncertain by
# eps / a^2 -- 1.4e-3 relative at roughness -0.8 -- in ANY fp32 evaluation.  The reference's own fp32-vs-fp64 error of the specular image on
# such inputs is 1.78e-4 (reference-made fixture g9_ratio1_unit_normals, 120x160; 4.98e-5 on the pooled fixture g7), and it reaches the
# objective's gradients through the render-loss cotangent 2 (rendered - im) seg.  The fp32 oracle's own error on the same inputs -- the proxy
# e_ref of this file -- is ONE sample of that noise and came out at 4e-5 where the kernels' sample was 1.3e-4 (case 16 with F0 = 0.115; the
# fused and the unfused HIP routes agree to 1e-7 there, each pixel but one within 1e-7).  So for q = 1 the objective's gradient bound has
# the reference-made yardstick as its floor: max(2 e_proxy, 2 x 1.78e-4); the layer-level quantities keep max(2 e_proxy, 1e-4).
E_REF_SPEC_RATIO1 = 1.78e-4


def _cases():
    g = torch.Generator().manual_seed(20250)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())
    out = []
    for case in range(N_CASES):
        bn, R, C, q = ri(1, 3), ri(3, 13), ri(3, 17), (1, 2)[ri(0, 1)]
        ew = (16, 32)[ri(0, 1)]
        K, eh = ri(1, 24 if ri(0, 2) == 0 else 12), ri(1, 9 if ew == 16 else 16)
        ind = [float(ri(0, 4) > 0) for _ in range(bn)]
        out.append(dict(case=case, bn=bn, R=R, C=C, q=q, K=K, eh=eh, ew=ew, benign=bool(ri(0, 1)), ind=ind))
    # round 6: the constructor kwargs fov / F0 / cameraPos (models.py:408) vary too; drawn from a SECOND stream so that the thirty
    # shapes above stay the ones rounds 4-5 ran.  Every third case keeps the reference's defaults.
    g2 = torch.Generator().manual_seed(20260)
    ru = lambda lo, hi: round(float(lo + (hi - lo) * torch.rand(1, generator=g2).item()), 4)
    for c in out:
        if c["case"] % 3 == 0:
            c.update(fov=57.0, F0=0.05, cam=[0.0, 0.0, 0.0])
        else:
            c.update(fov=ru(35.0, 75.0), F0=ru(0.02, 0.12), cam=[ru(-0.3, 0.3), ru(-0.3, 0.3), ru(-0.3, 0.4)])
    return out


CASES = _cases()


@pytest.fixture(scope="module")
def sgr():
    import inverserenderingofindoorscene_amd as pkg
    from inverserenderingofindoorscene_amd import _lib
    _lib.load()
    return pkg


def _objective_oracle(O, inp, ind, R, C, eh, ew, dtype, fov=57.0, F0=0.05, cam=(0.0, 0.0, 0.0)):
    x = {k: v.to("cuda", dtype) for k, v in inp.items()}
    for k in SG:
        x[k] = x[k].clone().requires_grad_(True)
    env, d, s = O.render_from_sg(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], eh, ew, fov, F0, cam)
    r, _, _, _ = O.render_loss(d, s, x["im"], x["seg"], R, C)
    c, _, _, _ = O.recon_loss(env, x["env_gt"], x["seg"], ind.to("cuda", dtype), R, C)
    g = torch.autograd.grad(r + 10.0 * c, [x[k] for k in SG])
    return r.detach(), c.detach(), [t.detach() for t in g]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("c", CASES, ids=[f"case{c['case']}_bn{c['bn']}_{c['R']}x{c['C']}_q{c['q'] ** 2}_K{c['K']}_{c['eh']}x{c['ew']}" for c in CASES])
def test_random_case_vs_oracle(sgr, c):
    from oracle import sg_oracle as O
    bn, R, C, q, K, eh, ew = c["bn"], c["R"], c["C"], c["q"], c["K"], c["eh"], c["ew"]
    inp = O.synthetic_inputs(bn, R * q, C * q, R, C, K, eh, ew, seed=7000 + c["case"], benign=c["benign"])
    ind = torch.tensor(c["ind"]).reshape(bn, 1, 1, 1)
    x = {k: v.cuda() for k, v in inp.items()}
    for k in SG:
        x[k].requires_grad_(True)
    fov, F0, cam = c["fov"], c["F0"], c["cam"]
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, fov=fov, F0=F0, cameraPos=cam, envWidth=ew, envHeight=eh)
    # ---- the fused layer: values and SG gradients --------------------------------------------------------------------------------
    env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=True)
    g = torch.Generator().manual_seed(99 + c["case"])
    cts = [torch.randn(env.shape, generator=g), torch.randn(d.shape, generator=g), torch.randn(s.shape, generator=g)]
    grads = torch.autograd.grad([env, d, s], [x[k] for k in SG], grad_outputs=[t.cuda() for t in cts])
    r64, _, e32 = oracle_with_noise(O, inp, cts, eh, ew, SG, "cuda", None, fov, F0, cam)
    for k, v in (("env", env), ("diffuse", d), ("spec", s)):
        assert rel_l2(v.detach(), r64[k]) <= tol2(e32[k]), (c, k, rel_l2(v.detach(), r64[k]), e32[k])
    for k, a in zip(SG, grads):
        assert torch.isfinite(a).all(), (c, k)
        assert rel_l2(a, r64["g_" + k]) <= tol2(e32["g_" + k]), (c, "g_" + k, rel_l2(a, r64["g_" + k]), e32["g_" + k])
    # ---- the fused light objective ------------------------------------------------------------------------------------------------
    obj = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"], x["env_gt"], ind.cuda(), 1.0, 10.0)
    g_obj = torch.autograd.grad(obj[0], [x[k] for k in SG])
    ro, co, go = _objective_oracle(O, inp, ind, R, C, eh, ew, torch.float64, fov, F0, cam)
    r3, c3, g3 = _objective_oracle(O, inp, ind, R, C, eh, ew, torch.float32, fov, F0, cam)
    assert scalar_close(obj[1].item(), ro.item(), r3.item() - ro.item()), (c, "renderErr", obj[1].item(), ro.item(), r3.item())
    assert scalar_close(obj[2].item(), co.item(), c3.item() - co.item()), (c, "reconstErr", obj[2].item(), co.item(), c3.item())
    for k, a, b, b32 in zip(SG, g_obj, go, g3):
        assert torch.isfinite(a).all(), (c, k)
        if float(b.norm()) == 0.0:      # every image of the case without a ground-truth env AND no live render pixel: nothing to compare
            assert float(a.abs().max()) == 0.0
            continue
        bound = tol2(rel_l2(b32, b)) if q > 1 else max(tol2(rel_l2(b32, b)), 2.0 * E_REF_SPEC_RATIO1)
        assert rel_l2(a, b) <= bound, (c, "objective g_" + k, rel_l2(a, b), rel_l2(b32, b))
    with torch.no_grad():               # the forward-only route (no gradient kernel) returns the same values
        ng = sgr.light_objective(layer, x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], x["im"], x["seg"], x["env_gt"], ind.cuda(), 1.0, 10.0)
    assert abs(ng[0].item() - obj[0].item()) <= 2e-6 * abs(obj[0].item()), (c, ng[0].item(), obj[0].item())
    assert torch.equal(ng[1], obj[1])
