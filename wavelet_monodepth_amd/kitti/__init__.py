from .depth_decoder import DepthDecoder, DepthWaveProgressiveDecoder  # noqa: F401
