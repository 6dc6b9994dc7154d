"""The build's ``model/geoformer`` counterpart (same class names and forward() signatures)."""
from .config import cfg, load_config, set_config  # noqa: F401
from .geoformer import GeoFormer, cal_geodesic, get_batch_offsets  # noqa: F401
from .criterion import InstSetCriterion  # noqa: F401,E402
from .geoformer_fs import GeoFormerFS  # noqa: F401,E402
from .criterion_fs import FSInstSetCriterion  # noqa: F401,E402
