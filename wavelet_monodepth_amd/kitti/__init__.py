from .depth_decoder import DepthDecoder, DepthWaveProgressiveDecoder  # noqa: F401
from .sparse_decoder import SparseDepthWaveProgressiveDecoder  # noqa: F401
from .network_constructors import make_depth_decoder, make_depth_encoder  # noqa: F401
