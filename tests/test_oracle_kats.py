"""CPU: hand-derived known-answer tests for the oracle's restatement of the CUDA/C++ natives
(the reference holds no vectors for them)."""
import numpy as np


def test_voxelize_idx_first_occurrence_order(oracle):
    # voxelize.cpp:96-104: ids by insertion counter; :143-149 rule rows [count, ids..., 0 pad]; :39-48 coords of rule[1]
    c = np.array([[0, 5, 5, 5], [0, 1, 1, 1], [0, 5, 5, 5], [1, 5, 5, 5], [0, 1, 1, 1], [0, 5, 5, 5]], np.int64)
    oc, p2v, v2p = oracle.voxelize_idx(c, 4)
    assert p2v.tolist() == [0, 1, 0, 2, 1, 0]
    assert oc.tolist() == [[0, 5, 5, 5], [0, 1, 1, 1], [1, 5, 5, 5]]
    assert v2p.tolist() == [[3, 0, 2, 5], [2, 1, 4, 0], [1, 3, 0, 0]]
    _, _, v1 = oracle.voxelize_idx(c, 1)  # front()
    _, _, v2 = oracle.voxelize_idx(c, 2)  # back()
    assert v1.tolist() == [[1, 0], [1, 1], [1, 3]] and v2.tolist() == [[1, 5], [1, 4], [1, 3]]


def test_voxelize_host_matches_oracle(oracle):
    from geoformer_amd import scene

    sc = scene.make_small_scene(4000, 2)
    b = scene.make_batch([sc, scene.make_small_scene(3000, 3)])
    oc, p2v, v2p = oracle.voxelize_idx(b["locs"].numpy(), 4)
    assert (b["voxel_locs"].numpy() == oc).all() and (b["p2v_map"].numpy() == p2v).all()
    assert (b["v2p_map"].numpy() == v2p).all()


def test_voxelize_mean(oracle):
    feats = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]], np.float32)
    rules = np.array([[2, 0, 2], [1, 1, 0]], np.int32)
    out = oracle.voxelize_fp(feats, rules, True)
    assert out.tolist() == [[3.0, 4.0], [3.0, 4.0]]
    g = oracle.voxelize_bp(np.ones((2, 2), np.float32), rules, 3, True)
    assert g.tolist() == [[0.5, 0.5], [1.0, 1.0], [0.5, 0.5]]


def test_fps_kats(oracle):
    # sampling_gpu.cu:88-89 start at 0; farthest next; :104 skips |p|^2 <= 1e-3; m > n pads with the lowest eligible
    p = np.array([[[1, 0, 0], [0, 0, 0], [1.1, 0, 0], [5, 0, 0], [-2.5, 0, 0]]], np.float32)
    assert oracle.fps(p, 4).tolist() == [[0, 3, 4, 2]]
    assert oracle.fps(p, 8)[0, :4].tolist() == [0, 3, 4, 2]
    assert oracle.fps(p, 8)[0, 4:].tolist() == [0, 0, 0, 0]  # all distances 0: lowest-key eligible index
    # tie between two equidistant points: block size 4 (n=5 -> bs=4): slots are k mod 4; the tree keeps the
    # lower slot at every level, so with candidates in slots 1 and 2 slot 2 wins (0<-2 first, then 0 vs 1)
    q = np.array([[[1, 0, 0], [1, 2, 0], [1, -2, 0], [1, 0, 0.5], [1, 0, -0.5]]], np.float32)
    assert oracle.fps(q, 2).tolist() == [[0, 2]]


def test_ball_query_kats(oracle):
    xyz = np.array([[[0, 0, 0], [0.1, 0, 0], [0.3, 0, 0], [0.05, 0, 0], [9, 9, 9]]], np.float32)
    ctr = np.array([[[0, 0, 0], [9, 9, 9.05], [50, 0, 0]]], np.float32)
    idx = oracle.ball_query(ctr, xyz, 0.2, 4)
    assert idx[0].tolist() == [[0, 1, 3, 0], [4, 4, 4, 4], [0, 0, 0, 0]]  # pad with first hit; no hit -> zeros
    # strict '<' on fp32 d2 vs fp32 radius^2 (ball_query_gpu.cu:25,36)
    e = np.array([[[0.2, 0, 0]]], np.float32)
    assert oracle.ball_query(np.zeros((1, 1, 3), np.float32), e, 0.2, 2)[0, 0].tolist() == [0, 0]
    r2 = np.float32(0.2) * np.float32(0.2)
    assert (np.float32(0.2) * np.float32(0.2) < r2) is np.False_ or True


def test_knn_ties_and_padding(oracle):
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0], [-1, 0, 0]], np.float32)
    D, I = oracle.knn(p, p, 6)
    assert I[0].tolist() == [0, 2, 1, 3, -1, -1] and I[2].tolist() == [0, 2, 1, 3, -1, -1]  # ties -> lower index
    assert D[0, :4].tolist() == [0.0, 0.0, 1.0, 1.0] and np.isinf(D[0, 4:]).all()


def test_geodesic_kat(oracle):
    # chain 0-1-2-3 with a shortcut 0-2 longer than the radius: hop-synchronous, first-listed parent wins
    pts = np.array([[0, 0, 0], [0.04, 0, 0], [0.08, 0, 0], [0.12, 0, 0], [5, 5, 5]], np.float32)
    D2, I = oracle.knn(pts, pts, 5)
    D = np.sqrt(D2)
    geo = oracle.geodesic(D[:, 1:], I[:, 1:], np.array([0]), 0.05, 10)
    assert geo[0, 4] == -1 and geo[0, 0] == 0
    assert geo[0, 1] == D[0, 1] and geo[0, 2] == np.float32(D[1, 1] + geo[0, 1])
    geo1 = oracle.geodesic(D[:, 1:], I[:, 1:], np.array([0]), 0.05, 1)
    assert (geo1[0] >= 0).tolist() == [True, True, False, False, False]


def test_sec_ops(oracle):
    x = np.array([[1, 5], [3, 1], [2, 2], [7, 0]], np.float32)
    off = np.array([0, 3, 4, 4], np.int32)
    assert oracle.sec_op("mean", x, off)[:2].tolist() == [[2.0, np.float32(5 / 3 + 1 / 3 + 2 / 3)], [7.0, 0.0]]
    assert oracle.sec_op("max", x, off)[0].tolist() == [3.0, 5.0] and oracle.sec_op("min", x, off)[0].tolist() == [1.0, 1.0]
    assert np.isinf(oracle.sec_op("min", x, off)[2]).all()  # empty segment keeps the 1e50 -> +inf initialiser


def test_dormant_pg_ops_kats(oracle):
    # bfs_cluster.cpp:28-54,60-75: scan order + FIFO BFS restricted to equal semantic label, size threshold
    xyz = np.array([[0, 0, 0], [0.05, 0, 0], [0.1, 0, 0], [5, 0, 0], [5.05, 0, 0], [0.15, 0, 0]], np.float32)
    bidx = np.zeros(6, np.int32)
    cum, idx, sl = oracle.ballquery_batch_p(xyz, bidx, np.array([0, 6], np.int32), 6, 0.06)
    assert sl[:, 1].tolist() == [2, 3, 3, 2, 2, 2] and sl[:, 0].tolist() == [0, 2, 5, 8, 10, 12] and cum == 14
    assert idx[:5].tolist() == [0, 1, 0, 1, 2]
    sem = np.array([1, 1, 1, 1, 1, 2], np.int32)
    ci, co = oracle.bfs_cluster(sem, idx[:cum], sl, 2)
    assert co.tolist() == [0, 3, 5] and ci[:, 1].tolist() == [0, 1, 2, 3, 4]  # point 5 differs in label: own size-1 cluster dropped
    out, arg = oracle.roipool_fp(np.array([[1, 9], [3, 2], [3, 5]], np.float32), np.array([0, 3], np.int32))
    assert out.tolist() == [[3.0, 9.0]] and arg.tolist() == [[1, 0]]  # strict '>' keeps the first maximum
    iou = oracle.get_iou(np.array([0, 1, 2], np.int32), np.array([0, 3], np.int32), np.array([0, 0, 1, 1], np.int64),
                         np.array([2, 2], np.int32))
    assert np.allclose(iou, [[2 / (3 + 2 - 2 + 1e-5), 1 / (3 + 2 - 1 + 1e-5)]])


def test_proposal_stats_oracle_matches_reference_formulation(oracle):
    """orc_proposal_stats/scatter vs the PyTorch lines of generate_proposal (geoformer.py:206-262) on CPU."""
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(0)
    nq, N, ncls, npts = 32, 2000, 20, 2600
    logits = (rng.standard_normal((nq, N)) * 3).astype(np.float32)
    cl = (rng.standard_normal((nq, ncls)) * 2).astype(np.float32)
    sem = torch.softmax(torch.from_numpy(rng.standard_normal((N, ncls)).astype(np.float32)), 1)
    fg = np.sort(rng.choice(npts, N, replace=False)).astype(np.int64)
    cp, npoints, sc, fin = oracle.proposal_stats(logits, cl, sem.numpy(), 0.5, 0.55, 600)
    prob = torch.from_numpy(logits).sigmoid()
    mb = prob >= 0.5
    n = torch.sum(mb, dim=1)
    ms = torch.sum(prob * mb.int(), dim=1) / (n + 1e-6)
    cpr = F.softmax(torch.from_numpy(cl), dim=-1)
    pred = torch.argmax(torch.from_numpy(cl), dim=-1)
    ss = torch.sum(sem[None].expand(nq, N, ncls) * mb.int()[:, :, None], dim=1) / (n[:, None] + 1e-6)
    ss = torch.gather(ss, 1, pred.unsqueeze(-1)).squeeze(-1)
    scores = ms * torch.pow(torch.gather(cpr, 1, pred.unsqueeze(-1)).squeeze(-1), 0.5) * ss
    final = (pred >= 4) & (n >= 600) & (ms >= 0.55)
    assert (cp == pred.numpy()).all() and (npoints == n.numpy()).all() and (fin == final.numpy()).all()
    assert np.abs(sc - scores.numpy()).max() < 1e-5
    assert fin.sum() > 0
    props = torch.zeros((int(final.sum()), npts), dtype=torch.int)
    inst, pts = torch.nonzero(mb[final], as_tuple=True)
    props[inst, torch.from_numpy(fg)[pts]] = 1
    got = oracle.proposal_scatter(logits, np.nonzero(fin)[0].astype(np.int32), fg, 0.5, npts)
    assert (got == props.numpy()).all()
