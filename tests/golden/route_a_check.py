"""Route A of INTEGRATION.md, exercised: the UNMODIFIED reference model (imported from /root/reference, build
container only) running on top of this repo's drop-in modules -- geoformer_amd.dropin.install() registers
geoformer_amd.spconv, PG_OP, pointnet2._ext and faiss under the names the reference imports; the reference's own
wrappers lib/pointgroup_ops/functions/pointgroup_ops.py and lib/pointnet2/pointnet2_utils.py sit in between, untouched.

There is no GPU in the build container, so the native layer under the drop-ins is stood in for by the oracle
(oracle.cpu_backend.installed(), test infrastructure), and the reference's CUDA-only idioms (`.cuda()`,
torch.cuda.FloatTensor) are made CPU no-ops the same way the golden generator does.  The run must reproduce the
committed fixture tests/golden/geoformer_s8k_eval.npz, which the reference produced over the generator's own shims:
same semantic scores, foreground set, FPS picks, geodesic distances, decoder output, mask logits and proposals.

    python tests/golden/route_a_check.py        (exit code 0 = identical within the fixtures' tolerances)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.argv = ["route_a_check", "--config", os.path.join(REF, "config/test_geoformer_scannet.yaml")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import geoformer_amd.dropin as dropin  # noqa: E402
from oracle import cpu_backend  # noqa: E402

mods = dropin.install()
# CPU stand-ins for what only exists on a GPU box (test-side patches; the product keeps refusing CPU tensors)
dropin._chk = lambda t, dtype, name: None
_sqrt = torch.sqrt
torch.sqrt = lambda t, *a, **k: (torch.from_numpy(np.sqrt(t.detach().numpy())) if (not a and not k and t.dtype == torch.float32 and not t.requires_grad) else _sqrt(t, *a, **k))  # IEEE sqrt like a GPU (see ref_shims.py)
torch.Tensor.cuda = lambda self, *a, **k: self
torch.cuda.FloatTensor = lambda *shape: torch.zeros(*shape, dtype=torch.float32)
torch.cuda.IntTensor = lambda *shape: torch.zeros(*shape, dtype=torch.int32)
for dummy in ("trimesh", "tensorboardX"):
    import types

    sys.modules.setdefault(dummy, types.ModuleType(dummy))
sys.path.insert(0, REF)
os.chdir(REF)

with cpu_backend.installed():
    import spconv  # noqa: E402

    assert spconv is mods["spconv"] and spconv.__name__ == "geoformer_amd.spconv"
    from model.geoformer.geoformer import GeoFormer  # noqa: E402  (the reference's class, unmodified)
    from model.geoformer import geodesic_utils  # noqa: E402
    import lib.pointgroup_ops.functions.pointgroup_ops as ref_pg  # noqa: E402  (reference wrapper over PG_OP)
    import lib.pointnet2.pointnet2_utils as ref_p2  # noqa: E402  (reference wrapper over pointnet2._ext)

    assert ref_pg.PG_OP is mods["PG_OP"] and ref_p2._ext is mods["pointnet2._ext"]
    from geoformer_amd import scene  # noqa: E402
    from tests.util import synthetic_state_dict  # noqa: E402

    z = np.load(os.path.join(HERE, "geoformer_s8k_eval.npz"))
    torch.manual_seed(0)
    m = GeoFormer()
    m.load_state_dict(synthetic_state_dict(m.state_dict(), int(z["weight_seed"])))
    m.eval()
    batch = scene.make_batch([scene.make_small_scene(int(z["scene_points"]), int(z["scene_seed"]))])
    cap = {}
    orig_agg, orig_dec = m.forward_aggregator, m.forward_decoder

    def agg(*a, **k):
        r = orig_agg(*a, **k)
        cap["context_locs"], cap["context_feats"], cap["pre_enc_inds"] = [t.detach().numpy().copy() for t in r]
        return r

    def dec(context_locs, context_feats, query_locs, pc_dims, geo_dists, pre_enc_inds):
        cap["geo"] = geo_dists[0].numpy().copy()
        r = orig_dec(context_locs, context_feats, query_locs, pc_dims, geo_dists, pre_enc_inds)
        cap["dec_outputs"] = r.detach().numpy().copy()
        return r

    m.forward_aggregator, m.forward_decoder = agg, dec
    np.random.seed(int(z["numpy_seed"]))
    with torch.no_grad():
        out = m(batch, 300, training=False)

checks = []


def check(name, ok):
    checks.append((name, bool(ok)))
    print(("ok   " if ok else "FAIL ") + name)


check("semantic_scores <= 1e-4", np.abs(out["semantic_scores"].numpy() - z["semantic_scores"]).max() < 1e-4)
check("fg_idxs identical", (out["fg_idxs"].numpy() == z["fg_idxs"]).all())
check("FPS picks identical", (cap["pre_enc_inds"] == z["pre_enc_inds"]).all())
check("context_locs identical", np.abs(cap["context_locs"] - z["context_locs"]).max() == 0)
check("context_feats <= 1e-4", np.abs(cap["context_feats"] - z["context_feats"]).max() < 1e-4)
geo = cap["geo"]
check("geodesic reach sets", ((geo >= 0).sum(1) == z["geo_reached"]).all())
check("geodesic distances bit-exact (sample)", (geo[::8, ::4] == z["geo_sub"]).all())
check("decoder output <= 1e-4", np.abs(cap["dec_outputs"] - z["dec_outputs"]).max() < 1e-4)
mp = out["mask_predictions"][-1]
check("cls_logits <= 1e-4", np.abs(mp["cls_logits"].numpy() - z["cls_logits"]).max() < 1e-4)
check("mask_logits <= 1e-4 (sample)", np.abs(mp["mask_logits"][0].numpy()[::8, ::4] - z["mask_logits_sub"]).max() < 1e-4)
cls_final, scores_final, masks_final = out["proposal_scores"]
check("proposal classes identical", (np.asarray(cls_final) == z["proposal_cls"]).all())
check("proposal scores <= 1e-4", np.abs(np.asarray(scores_final) - z["proposal_scores"]).max() < 1e-4)
sys.exit(0 if all(ok for _, ok in checks) else 1)
