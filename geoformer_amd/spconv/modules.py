"""``spconv.modules`` mirror: the marker base class the reference imports
(``from spconv.modules import SparseModule``, geoformer_modules.py:6)."""
import torch.nn as nn


class SparseModule(nn.Module):
    """Modules that consume/produce a SparseConvTensor (everything else in a
    SparseSequential is applied to ``.features``)."""
