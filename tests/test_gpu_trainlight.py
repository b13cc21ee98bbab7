"""GPU: the synthetic cascade-0 training loop (BASELINE config 3) runs, its losses fall, and its first
step's losses match the oracle evaluated on the same tensors."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused,hip_heads", [(True, True), (False, False), (True, False)])
def test_synthetic_trainlight_loop_descends_and_matches_oracle(fused, hip_heads):
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import train_light_synthetic as T
    from oracle import sg_oracle as O

    bn, imH, imW, R, C, K = 2, 48, 64, 24, 32, 12
    hist, _ = T.train(bn=bn, steps=6, imH=imH, imW=imW, R=R, C=C, K=K, verbose=False, fused=fused, hip_heads=hip_heads)
    assert all(torch.isfinite(torch.tensor(h)).all() for h in hist)
    assert hist[-1][0] < hist[0][0], hist                       # Adam makes progress on the objective

    # first-step losses against the oracle (fp64) on identical tensors
    batch = T.make_batch(bn, imH, imW, R, C, 8, 16, torch.device("cuda"), 0)
    g = torch.Generator().manual_seed(1)
    params = [(torch.randn(s, generator=g) * 0.5) for s in ((bn, K, 3, R, C), (bn, K, R, C), (bn, 3 * K, R, C))]
    axis, lam, w = T.decoder_heads(*[p.double() for p in params])
    cpu = {k: v.cpu().double() for k, v in batch.items()}
    env, d, s = O.render_from_sg(cpu["albedo"], cpu["normal"], cpu["rough"], axis, lam, w)
    rerr, _, _, _ = O.render_loss(d, s, cpu["im"], cpu["seg"], R, C)
    cerr, _, _, _ = O.recon_loss(env, cpu["env_gt"], cpu["seg"], cpu["env_ind"], R, C)
    assert abs(hist[0][1] - rerr.item()) < 5e-5 * rerr.item(), (hist[0][1], rerr.item())      # relative: the fp32 example vs the fp64 oracle (measured ~1e-6)
    assert abs(hist[0][2] - cerr.item()) < 5e-5 * cerr.item(), (hist[0][2], cerr.item())
