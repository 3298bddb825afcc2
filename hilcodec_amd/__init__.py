"""hilcodec_amd — MI355X-native (gfx950) HILCodec encode -> RVQ -> decode forward path.

Importing the package loads `lib/libhilcodec_amd.so` (hand-written HIP kernels behind a C ABI,
`include/hilcodec_amd.h`) and fails loudly if it is missing: there is no CPU or PyTorch fallback."""
from . import _lib  # noqa: F401  (raises if the HIP library is not built)
from .models.hilcodec.models import HILCodec  # noqa: F401

__all__ = ["HILCodec"]
