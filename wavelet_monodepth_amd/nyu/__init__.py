from .densedepth_decoder import DecoderWave, SparseDecoderWave  # noqa: F401
