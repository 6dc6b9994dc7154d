"""Dev tool: the batch-4 / 550k training forward on the GPU against the oracle-backed host run, stage by stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_training_step import _setup
from oracle import cpu_backend
from oracle import oracle as orc
L = orc.lib(); L.orc_set_threads.restype = int; L.orc_set_threads(64)

def run(device):
    cfg, m, crit, batch = _setup(device, full=True, batch4=True)
    cap = {}
    dec = m.forward_decoder
    def dec_w(cl, cf, ql, pc, geo, pei):
        cap["pei"] = pei.detach().cpu().numpy().copy()
        cap["geo_reached"] = [int((g >= 0).sum()) for g in geo]
        cap["geo_sum"] = [float(torch.where(g >= 0, g, torch.zeros_like(g)).double().sum()) for g in geo]
        cap["cf"] = cf.detach().cpu().numpy().copy()
        r = dec(cl, cf, ql, pc, geo, pei)
        cap["dec"] = r.detach().cpu().numpy().copy()
        return r
    m.forward_decoder = dec_w
    np.random.seed(3)
    out = m(batch, 5)
    loss, info = crit(out, batch, 5)
    return out, float(loss), info, cap

with cpu_backend.installed():
    oc, lc, ic, cc = run("cpu")
og, lg, ig, cg = run("cuda")
print("loss", lc, lg); print(ic); print(ig)
print("sem maxdiff", float((og["semantic_scores"].cpu() - oc["semantic_scores"]).abs().max()))
fg_g, fg_c = og["fg_idxs"].cpu().numpy(), oc["fg_idxs"].numpy()
print("fg sub sizes", fg_g.shape, fg_c.shape, "equal", fg_g.shape == fg_c.shape and bool((fg_g == fg_c).all()))
print("pei equal", [bool((cg["pei"][b] == cc["pei"][b]).all()) for b in range(4)])
print("geo reached", cg["geo_reached"], cc["geo_reached"]); print("geo sum", cg["geo_sum"], cc["geo_sum"])
print("ctx feats maxdiff", np.abs(cg["cf"] - cc["cf"]).max(), "dec maxdiff", np.abs(cg["dec"] - cc["dec"]).max())
for l in range(4):
    for b in range(4):
        a, c = og["mask_predictions"][l]["mask_logits"][b].detach().cpu(), oc["mask_predictions"][l]["mask_logits"][b].detach()
        print(l, b, tuple(a.shape), tuple(c.shape), float((a - c).abs().max()) if a.shape == c.shape else None)
