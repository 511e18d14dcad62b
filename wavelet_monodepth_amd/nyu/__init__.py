from .densedepth_decoder import DecoderWave  # noqa: F401
