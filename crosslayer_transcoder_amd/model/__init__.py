from .clt import CrossLayerTranscoder, CrosslayerDecoder, Decoder, Encoder  # noqa: F401
from .jumprelu import JumpReLU, ReLU  # noqa: F401
from .topk import BatchTopK, PerLayerBatchTopK, PerLayerTopK  # noqa: F401
