"""Import shim for the real reference (aask1357/hilcodec at /root/reference) — BUILD CONTAINER ONLY.

Test infrastructure: used by `oracle/make_golden.py` and `tests/test_oracle_vs_reference.py` to run
the reference's own PyTorch modules on CPU.  `/root/reference` does not exist on the GPU box, so
everything that calls `load_reference()` is skipped there; nothing in the product imports this.

The reference's package `__init__` files pull in trainer-only third-party modules that are not
installed here (librosa, torchaudio, tensorboard, pesq, pystoi, soundfile); they are stubbed in
`sys.modules` — none of them is touched by the forward path.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("HILCODEC_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "hilcodec", "models.py"))


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _Stub(self.__name__ + "." + k)
        sys.modules[m.__name__] = m
        return m

    def __call__(self, *a, **k):
        return None


_loaded = None


def load_reference():
    """Returns a namespace with the reference's offline and streaming classes."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    for n in ["librosa", "librosa.filters", "librosa.util", "torchaudio", "torchaudio.transforms",
              "torchaudio.functional", "tensorboard", "torch.utils.tensorboard", "pesq", "pystoi",
              "soundfile"]:
        if n not in sys.modules:
            sys.modules[n] = _Stub(n)
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    # the reference's top-level packages are called `models`, `modules`, `utils`, `functional`, `optim`
    saved_path = list(sys.path)
    sys.path.insert(0, REFERENCE_ROOT)
    warnings.filterwarnings("ignore", category=FutureWarning)
    try:
        from models.hilcodec.models import HILCodec as OfflineHILCodec
        from models.hilcodec.streaming import HILCodec as StreamingHILCodec
        from models.hilcodec import vector_quantize as vq_new
        from modules import vector_quantize as vq_old
        import importlib
        ws = importlib.import_module("modules.weight_standardization")
    finally:
        sys.path[:] = saved_path
    ns = types.SimpleNamespace(OfflineHILCodec=OfflineHILCodec, StreamingHILCodec=StreamingHILCodec,
                               vq_new=vq_new, vq_old=vq_old, ws=ws)
    _loaded = ns
    return ns


def build_offline(ref, mk: dict, sd: dict):
    """Reference offline model with `sd` loaded and the codebooks marked initialised
    (kmeans_init=True leaves `initted=False`, `vector_quantize.py:86,139-140`)."""
    import copy
    model = ref.OfflineHILCodec(sample_rate=24000, channels_audio=1, **copy.deepcopy(mk)).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    # `weight_scale` (weight standardisation) is a buffer the constructor fills from norm_kwargs['scale']
    missing = [k for k in missing if not k.endswith("_extra_state") and not k.endswith(".weight_scale")]
    assert not missing and not unexpected, (missing, unexpected)
    for layer in model.quantizer.layers:
        layer.initted = True
    return model


def build_streaming(ref, mk: dict, offline_model, plain_weights: bool = False):
    """Reference streaming model (`models/hilcodec/streaming.py:651`) filled from an offline model
    with the offline->streaming correspondence that `scripts/HILCodec Onnx.ipynb` cell 1
    establishes, then `remove_weight_reparameterizations()` (`streaming.py:740-747`).

    The correspondence is expressed as (streaming module, offline module) pairs:
    streaming convs are bare (weight-normed) `nn.Conv1d`, offline ones sit two wrappers deep
    (`SConv1d.conv.conv` / `SConvTranspose1d.convtr.convtr`).

    `plain_weights`: the offline model's convs already carry plain `weight`s (re-parameterisation removed or folded by
    the caller): the streaming model's weight_norm hooks are removed FIRST and the plain weights copied in."""
    import copy
    mk2 = copy.deepcopy(mk)
    for k in ("spec_learnable", "causal", "pad_mode"):
        mk2.pop(k)
    model = ref.StreamingHILCodec(24000, **mk2).eval()

    def inner(m):
        return m.convtr.convtr if hasattr(m, "convtr") else m.conv.conv

    pairs = []          # (dst conv, src wrapper)
    scalars = []        # (dst Parameter, src Parameter)

    def resblock_pairs(dst, src):
        pairs.extend([(dst.block[0].pointwise[1], src.block[1]), (dst.block[0].depthwise, src.block[2]),
                      (dst.block[1].pointwise[1], src.block[4]), (dst.block[1].depthwise, src.block[5])])
        scalars.append((dst.res_scale_param, src.res_scale_param))

    se, oe = model.encoder, offline_model.encoder
    pairs.append((se.conv_pre, oe.conv_pre[1]))
    for s in range(len(oe.blocks)):
        for d, o in zip(se.blocks[s], oe.blocks[s]):
            resblock_pairs(d, o)
        pairs.append((se.spec_blocks[s].layer, oe.spec_blocks[s].layer))
        scalars.append((se.spec_blocks[s].scale_param, oe.spec_blocks[s].scale_param))
        pairs.append((se.downsample_pointwise[s][1], oe.downsample[s][2]))
        pairs.append((se.downsample_depthwise[s], oe.downsample[s][3]))
    pairs.append((se.spec_post.layer, oe.spec_post.layer))
    scalars.append((se.spec_post.scale_param, oe.spec_post.scale_param))
    pairs.append((se.conv_post_depthwise, oe.conv_post[1]))
    pairs.append((se.conv_post_pointwise, oe.conv_post[2]))

    sdec, seq = model.decoder, offline_model.decoder.model
    pairs.append((sdec.conv_pre_pointwise, seq[0]))
    pairs.append((sdec.conv_pre_depthwise, seq[1]))
    pos = 2
    for i in range(len(sdec.blocks)):
        pos += 2                                   # Scale/Identity + ELU
        pairs.append((sdec.upsample_depthwise[i], seq[pos]))
        pairs.append((sdec.upsample_pointwise[i], seq[pos + 1]))
        pos += 2
        for d in sdec.blocks[i]:
            resblock_pairs(d, seq[pos])
            pos += 1
    pairs.append((sdec.conv_post, seq[pos + 2]))

    if plain_weights:
        from torch.nn.utils import remove_weight_norm
        import torch.nn as nn
        for module in model.modules():
            if isinstance(module, (nn.Conv1d, nn.ConvTranspose1d)) and hasattr(module, "weight_g"):
                remove_weight_norm(module)
    for dst, src in pairs:
        dst.load_state_dict(inner(src).state_dict())
    for dst, src in scalars:
        dst.data.copy_(src.data)
    for q, dq, src in zip(model.quantizer.layers, model.dequantizer.layers, offline_model.quantizer.layers):
        for tgt in (q, dq):
            tgt.embed.data.copy_(src.embed.data)
            tgt.ema_num.data.copy_(src.ema_num.data)
    model.eval()
    if plain_weights:
        model.norm = "none"                       # hooks are gone already; merge_scaling still has to run
    model.remove_weight_reparameterizations()
    return model
