"""GPU: DeviceFeeder (pinned staging, copy stream, GPU voxelisation one batch ahead) hands over the same batch dicts
as the host collate + blocking copies of the reference's drivers (datasets/scannetv2_inst.py:389-455, test.py:56),
and the forward on them gives the same outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_feeder_batches_equal_host_collate(hip):
    from geoformer_amd import scene
    from geoformer_amd.feeder import DeviceFeeder

    specs = [[(8192, 3)], [(5000, 4), (7000, 5)], [(20000, 6)], [(3000, 7)], [(9000, 8)]]
    scenes = [[scene.make_small_scene(n, s) for n, s in sp] for sp in specs]
    feeder = DeviceFeeder((scene.collate_raw(sc) for sc in scenes), "cuda")
    got = list(feeder)
    assert len(got) == len(scenes)
    for g, sc in zip(got, scenes):
        ref = scene.make_batch(sc)
        assert set(ref) <= set(g)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert g[k].is_cuda and g[k].dtype == v.dtype and torch.equal(g[k].cpu(), v), k
            else:
                assert (np.asarray(g[k]) == np.asarray(v)).all(), k


def test_feeder_forward_equals_blocking_path(hip):
    from geoformer_amd import scene
    from geoformer_amd.feeder import DeviceFeeder
    from geoformer_amd.model import GeoFormer, load_config
    from tests.util import synthetic_state_dict

    m = GeoFormer(load_config("test_geoformer_scannet.yaml"))
    m.load_state_dict(synthetic_state_dict(m.state_dict(), 5))
    m.cuda()
    m.eval()
    scenes = [[scene.make_small_scene(8192, 11 + i)] for i in range(3)]
    outs = []
    for batch in DeviceFeeder((scene.collate_raw(sc) for sc in scenes), "cuda"):
        with torch.no_grad():
            outs.append(m(batch, 0, training=False)["semantic_scores"].clone())
    for o, sc in zip(outs, scenes):
        b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in scene.make_batch(sc).items()}
        with torch.no_grad():
            ref = m(b, 0, training=False)["semantic_scores"]
        assert torch.equal(o, ref)


def test_feeder_reports_producer_errors(hip):
    from geoformer_amd.feeder import DeviceFeeder

    def bad():
        yield {"locs": torch.zeros((4, 4), dtype=torch.int64) - 1, "offsets": torch.tensor([0, 4], dtype=torch.int32)}

    with pytest.raises(Exception):
        list(DeviceFeeder(bad(), "cuda"))
