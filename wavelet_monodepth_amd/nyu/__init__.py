from .densedepth_decoder import Decoder, Decoder224, DecoderWave, DecoderWave224, SparseDecoderWave  # noqa: F401
from .model import Model, NyuResnetEncoder  # noqa: F401
