from .models import HILCodec
