"""GPU: the module-shaped mirrors (PG_OP / pointnet2._ext / faiss / spconv) under the reference's import names."""
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_install_registers_reference_import_names(hip):
    from geoformer_amd import dropin

    mods = dropin.install()
    import faiss
    import faiss.contrib.torch_utils  # noqa: F401
    import PG_OP
    import pointnet2._ext as _ext
    import spconv
    from spconv.modules import SparseModule  # noqa: F401

    assert spconv is mods["spconv"] and hasattr(spconv, "SubMConv3d") and hasattr(spconv, "SparseInverseConv3d")
    for name in ("voxelize_idx", "voxelize_fp", "voxelize_bp", "point_recover_fp", "point_recover_bp",
                 "ballquery_batch_p", "bfs_cluster", "roipool_fp", "roipool_bp", "get_iou", "sec_mean", "sec_min",
                 "sec_max"):
        assert hasattr(PG_OP, name)  # pointgroup_ops_api.cpp:6-23
    for name in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn", "three_interpolate",
                 "three_interpolate_grad", "ball_query", "group_points", "group_points_grad"):
        assert hasattr(_ext, name)  # bindings.cpp:8-21
    with pytest.raises(RuntimeError):
        _ext.furthest_point_sampling(torch.zeros(1, 8, 3), 4)  # "CPU not supported"
    with pytest.raises(RuntimeError):
        _ext.gather_points(torch.zeros(1, 3, 8).cuda().transpose(1, 2), torch.zeros(1, 4, dtype=torch.int32).cuda())
    assert faiss.GpuIndexFlatConfig().device == 0


def test_pg_op_mirror(hip, oracle):
    from geoformer_amd import dropin, scene

    pg = dropin.install()["PG_OP"]
    sc = scene.make_small_scene(3000, 4)
    b = scene.make_batch([sc])
    coords = b["locs"]
    oc, im, om = coords.new(), torch.IntTensor(coords.shape[0]).zero_(), torch.IntTensor()
    pg.voxelize_idx(coords, oc, im, om, 1, 4)  # resizes its outputs like the native
    roc, rim, rom = oracle.voxelize_idx(coords.numpy(), 4)
    assert (oc.numpy() == roc).all() and (im.numpy() == rim).all() and (om.numpy() == rom).all()
    feats = _dev(np.concatenate([sc["rgb"], sc["xyz"]], 1))
    out = torch.zeros((om.shape[0], 6), device="cuda")
    pg.voxelize_fp(feats, out, om.cuda(), 4, om.shape[0], om.shape[1] - 1, 6)
    assert (out.cpu().numpy() == oracle.voxelize_fp(feats.cpu().numpy(), rom, True)).all()
    # point_recover_fp: every point receives its voxel's row (sum mode of the transposed map)
    rec = torch.zeros((coords.shape[0], 6), device="cuda")
    pg.point_recover_fp(out, rec, om.cuda(), om.shape[0], om.shape[1] - 1, 6)
    assert torch.equal(rec.cpu(), out.cpu()[im.long()])


def test_faiss_shim_exact_knn(hip, oracle):
    from geoformer_amd import dropin, scene

    fa = dropin.install()["faiss"]
    pts = scene.make_small_scene(6000, 9)["xyz"]
    index = fa.GpuIndexFlatL2(fa.StandardGpuResources(), 3, fa.GpuIndexFlatConfig())
    x = _dev(pts)
    D = torch.zeros((x.shape[0], 64), device="cuda")
    I = torch.zeros((x.shape[0], 64), dtype=torch.int64, device="cuda")
    index.add(x)
    index.search(x, 64, D, I)
    index.reset()
    rD, rI = oracle.knn(pts, pts, 64)
    assert (I.cpu().numpy() == rI).all()
    assert (D.cpu().numpy() == rD).all()  # squared L2, like faiss
