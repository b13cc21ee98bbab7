import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from conftest import rel_l2, oracle_fwd_bwd
import inverserenderingofindoorscene_amd as sgr
from oracle import sg_oracle as O
for fix in ("g8_cfg5_small.npz", "g7_cfg2_one_image.npz"):
    z = np.load('tests/golden/' + fix)
    cfg = {k: v for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist())}
    for k in ("bn", "imH", "imW", "R", "C", "K", "eh", "ew", "seed"): cfg[k] = int(cfg[k])
    R, C, K, eh, ew = cfg["R"], cfg["C"], cfg["K"], cfg["eh"], cfg["ew"]
    inp = O.synthetic_inputs(cfg["bn"], cfg["imH"], cfg["imW"], R, C, K, eh, ew, seed=cfg["seed"])
    g = torch.Generator().manual_seed(cfg["seed"] + 7)
    cts = [torch.randn((1, 3, R, C, eh, ew), generator=g), torch.randn((1, 3, R, C), generator=g), torch.randn((1, 3, R, C), generator=g)]
    x = {k: inp[k].cuda().requires_grad_(True) for k in ("albedo", "normal", "rough", "axis", "lamb", "weight")}
    layer = sgr.renderingLayer(imWidth=C, imHeight=R, envWidth=ew, envHeight=eh)
    for need_env in (True, False):
        env, d, s = layer.forwardSG(x["albedo"], x["normal"], x["rough"], x["axis"], x["lamb"], x["weight"], need_env=need_env)
        outs, c = ([env, d, s], cts) if need_env else ([d, s], cts[1:])
        gr = torch.autograd.grad(outs, [x["albedo"], x["normal"], x["rough"]], grad_outputs=[t.cuda() for t in c])
        if not need_env: continue
        st = int(z["strides"][1]) if "strides" in z else 1
        for k, got in zip(("albedo", "normal", "rough"), gr):
            got = got[..., ::st, ::st].cpu()
            print(fix, k, "gpu vs ref64 %.2e   ref32 vs ref64 %.2e" % (rel_l2(got, z["ref64_glin_" + k]), rel_l2(z["ref32_glin_" + k], z["ref64_glin_" + k])))
