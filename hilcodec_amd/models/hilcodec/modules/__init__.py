from .conv import (CausalSTFT, ConvParams, NormConv1d, NormConvTranspose1d, SConv1d, SConvTranspose1d,
                   get_extra_padding_for_conv1d)
from .seanet import L2Norm, Scale, SEANetDecoder, SEANetEncoder, SEANetResnetBlock, SpecBlock
