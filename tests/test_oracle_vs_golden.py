"""CPU: the oracle restatement against fixtures produced by the unmodified reference
(oracle/make_golden.py).  fp64 oracle vs the reference run in fp64 must agree to
round-off; fp32 oracle vs reference fp32 to fp32 round-off."""
import numpy as np
import pytest
import torch

from conftest import rel_l2, rel_max
from oracle import sg_oracle as O

NAMES = ("albedo", "normal", "rough", "axis", "lamb", "weight")


def _run(z, cfg, dtype):
    x = {k: torch.from_numpy(z["in_" + k]).to(dtype).requires_grad_(True) for k in NAMES}
    env, d, s = O.render_from_sg(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"],
                                 cfg["eh"], cfg["ew"], cfg["fov"], cfg["F0"], cfg["cam"])
    ct = {k: torch.from_numpy(z[k]).to(dtype) for k in ("ct_env", "ct_d", "ct_s")}
    lin = (env * ct["ct_env"]).sum() + (d * ct["ct_d"]).sum() + (s * ct["ct_s"]).sum()
    g_lin = torch.autograd.grad(lin, [x[k] for k in NAMES], retain_graph=True)
    im = torch.from_numpy(z["in_im"]).to(dtype)
    seg = torch.from_numpy(z["in_seg"]).to(dtype)
    env_gt = torch.from_numpy(z["in_env_gt"]).to(dtype)
    rerr, ren, num, den = O.render_loss(d, s, im, seg, cfg["R"], cfg["C"])
    ind = torch.ones(cfg["bn"], 1, 1, 1, dtype=dtype)
    cerr, env_sc, _, _ = O.recon_loss(env, env_gt, seg, ind, cfg["R"], cfg["C"])
    g_tot = torch.autograd.grad(rerr + 10.0 * cerr, [x[k] for k in NAMES], allow_unused=True)
    g_tot = [torch.zeros_like(x[k]) if g is None else g for g, k in zip(g_tot, NAMES)]
    return dict(env=env, diffuse=d, spec=s, rendered=ren, env_scaled=env_sc, render_err=rerr.reshape(1),
                recon_err=cerr.reshape(1), **{"glin_" + k: g for k, g in zip(NAMES, g_lin)},
                **{"gtot_" + k: g for k, g in zip(NAMES, g_tot)})


def test_oracle_fp64_matches_reference_fp64(golden):
    name, z, cfg = golden
    out = _run(z, cfg, torch.float64)
    for k, v in out.items():
        ref = z["ref64_" + k]
        tol = 1e-12 if ref.dtype == np.float64 else 2e-7      # env-sized refs are stored rounded to fp32
        assert rel_l2(v.detach(), ref) < tol, (name, k, rel_l2(v.detach(), ref))


def test_oracle_fp32_matches_reference_fp32(golden):
    name, z, cfg = golden
    out = _run(z, cfg, torch.float32)
    # forward: the two fp32 evaluations differ only in summation order
    for k in ("env", "diffuse", "spec", "rendered", "render_err", "recon_err"):
        assert rel_l2(out[k].detach(), z["ref32_" + k]) < 5e-6, (name, k)
    # backward: both sit on the reference's own fp32 noise floor; compare through fp64
    for k in NAMES:
        e_or = rel_l2(out["glin_" + k].detach(), z["ref64_glin_" + k])
        e_ref = rel_l2(z["ref32_glin_" + k], z["ref64_glin_" + k])
        assert e_or < max(3.0 * e_ref, 1e-5), (name, k, e_or, e_ref)


def test_tables_match_reference_bits(golden):
    """lamb_tan / weight_tan are pure functions of the inputs: the oracle's pre-map equals the
    reference's bit for bit in fp32."""
    name, z, cfg = golden
    lam = O.premap(torch.from_numpy(z["in_lamb"]))
    w = O.premap(torch.from_numpy(z["in_weight"]))
    assert np.array_equal(lam.numpy(), z["ref32_lamb_tan"])
    assert np.array_equal(w.numpy(), z["ref32_weight_tan"])


def test_oracle_light_heads_vs_reference_decoders():
    """oracle.light_heads against the outputs and autograd gradients of the reference's decoderLight modules
    (tests/golden/g4_heads.npz, made by oracle/make_golden_heads.py with a hook on dconvFinal)."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN_DIR, rel_l2
    from oracle import sg_oracle as O
    z = np.load(os.path.join(GOLDEN_DIR, "g4_heads.npz"))
    x = {k: torch.from_numpy(z["x_" + k]).double().requires_grad_(True) for k in ("axis", "lamb", "weight")}
    a, l, w, packed = O.light_heads(x["axis"], x["lamb"], x["weight"])
    outs = dict(axis=a, lamb=l, weight=w)
    for k, y in outs.items():
        assert (y.detach() - torch.from_numpy(z["y_" + k]).double()).abs().max().item() < 5e-7, k
    K = a.shape[1]
    assert torch.equal(packed[:, :3 * K], a.reshape(a.shape[0], 3 * K, *a.shape[3:])) and torch.equal(packed[:, 3 * K:4 * K], l)
    assert torch.equal(packed[:, 4 * K:], w)
    tot = sum((y * torch.from_numpy(z["ct_" + k]).double()).sum() for k, y in outs.items())
    g = torch.autograd.grad(tot, [x["axis"], x["lamb"], x["weight"]])
    for k, gi in zip(("axis", "lamb", "weight"), g):
        assert rel_l2(gi, z["gx_" + k]) < 1e-6, (k, rel_l2(gi, z["gx_" + k]))


def test_broadcast_restatement_matches_looped_oracle():
    """render_from_sg_broadcast (the reference's tensor formulation, used by bench.py's eager-GPU baseline)
    against the looped oracle, values and SG gradients, fp64."""
    import torch
    from conftest import rel_l2
    from oracle import sg_oracle as O
    inp = O.synthetic_inputs(2, 12, 16, 6, 8, 5, 4, 8, seed=9, dtype=torch.float64)
    outs = []
    for fn in (O.render_from_sg, O.render_from_sg_broadcast):
        x = {k: inp[k].clone() for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")}
        for k in ("axis", "lamb", "weight"):
            x[k].requires_grad_(True)
        env, d, s = fn(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], 4, 8)
        g = torch.autograd.grad((env * 0.01).sum() + (d * d).sum() + (s * 3).sum(), [x["axis"], x["lamb"], x["weight"]])
        outs.append((env, d, s) + tuple(g))
    for a, b in zip(*outs):
        assert rel_l2(a.detach(), b.detach()) < 1e-12
