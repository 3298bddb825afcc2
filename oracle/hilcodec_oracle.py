"""ORACLE — CPU restatement of the reference HILCodec encode -> RVQ -> decode forward path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the shipped package (`hilcodec_amd/`) never does
and fails loudly when its HIP library is missing.

What it restates (all `file:line` relative to the reference checkout, aask1357/hilcodec):

* offline path  — `models/hilcodec/models.py:111-118` (`HILCodec.forward`), built from
  `models/hilcodec/modules/seanet.py:249-479`, `models/hilcodec/modules/conv.py:61-358`,
  `models/hilcodec/vector_quantize.py:132-243`, `modules/vector_quantize.py:141-195,400-419,490-516`.
* streaming path — `models/hilcodec/streaming.py:25-157` (index-only RVQ, Dequantizer),
  `:160-365` (DWSBlock/ResBlock/SpecBlock with merged scales), `:368-648` (Encoder/Decoder with
  explicit caches), `models/hilcodec/causal_layers.py:72-188`, and the offline->streaming weight
  mapping of `scripts/HILCodec Onnx.ipynb` cell 1 + `streaming.py:740-747`.

The arithmetic is the reference's own arithmetic: plain fp32 PyTorch CPU ops (`F.conv1d`,
`F.conv_transpose1d`, `F.pad`, `F.elu`, `F.normalize`, matmul, `min/max(dim)`), written as a
flat functional program over the reference's *state-dict key layout* instead of its nn.Module
tree.  Parity pinning: `tests/test_oracle_vs_reference.py` runs this file against the real
reference modules imported from `/root/reference` (in the build container, where it exists) and
`tests/test_golden.py` checks it against `tests/golden/*.npz`, vectors generated from the real
reference by `oracle/make_golden.py`.  The reference ships no trained encoder/decoder weights
(`.MISSING_LARGE_BLOBS`), so its `onnx/input_speech.wav -> hil_speech_quantized.npy ->
hil_speech_output.wav` golden triple cannot be reproduced end-to-end: for trained-weight
end-to-end the parity is UNPINNED; the trained *codebooks* (`onnx/*_deq{i}.onnx`) together with
`hil_speech_quantized.npy` pin the Dequantizer only (tests/test_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

SD = Dict[str, Tensor]

WAV_STD = 0.1122080159                                  # seanet.py:264, streaming.py:380
SPEC_MEANS = [-4.554, -4.315, -4.021, -3.726, -3.477]   # seanet.py:265
SPEC_STDS = [2.830, 2.837, 2.817, 2.796, 2.871]         # seanet.py:266


# --------------------------------------------------------------------------------------
# weight re-parameterisations (folded on the host, once)
# --------------------------------------------------------------------------------------
def fold_weight_norm(v: Tensor, g: Tensor) -> Tensor:
    """`torch.nn.utils.weight_norm(dim=0)` hook: `conv.py:26-41` -> `torch._weight_norm(v, g, 0)`.
    For ConvTranspose1d weights `[Cin,1,k]` dim 0 is the *input* channel (== the channel for
    depthwise), exactly as the reference's hook computes it."""
    return torch._weight_norm(v, g, 0)


def fold_weight_standardization(v: Tensor, g: Optional[Tensor], scale: Optional[Tensor] = None,
                                eps: float = 1e-7) -> Tensor:
    """`modules/weight_standardization.py:30-41` with `dim=0`:
    `w = gain*scale*(v-mean)*rsqrt(max(var*fan_in, eps))`, var/mean over all axes but 0, biased."""
    axes = list(range(1, v.dim()))
    fan_in = 1.0
    for a in axes:
        fan_in *= v.size(a)
    var, mean = torch.var_mean(v, dim=axes, unbiased=False, keepdim=True)
    w = (v - mean) * torch.rsqrt(torch.clamp(var * fan_in, min=eps))
    if g is not None:
        if scale is not None:
            g = g * scale
        w = g * w
    return w


def conv_weight(sd: SD, prefix: str) -> Tuple[Tensor, Optional[Tensor]]:
    """Effective (weight, bias) of one reference conv given its state-dict prefix.  Accepts the
    weight_norm layout (`weight_g/_v`), the already-removed layout (`weight`) and the
    weight-standardisation layout (`weight_g/_v` [+ `weight_scale`] with `norm_type` marker)."""
    bias = sd.get(prefix + ".bias")
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"], bias
    v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
    if sd.get("__norm__", "weight_norm") == "weight_standardization":
        return fold_weight_standardization(v, g, sd.get(prefix + ".weight_scale")), bias
    return fold_weight_norm(v, g), bias


def with_weight_standardization(sd: SD, scale: Optional[float] = None) -> SD:
    """Copy of a `weight_g/_v` state dict marked as belonging to `HILCodec(norm="weight_standardization",
    norm_kwargs={"scale": scale})` (`conv.py:36-37`): the marker `conv_weight` reads, plus the `weight_scale` buffer the
    reference's constructor would have registered for every conv (`modules/weight_standardization.py:88-92`)."""
    out = dict(sd)
    out["__norm__"] = "weight_standardization"
    if scale is not None:
        for k in sd:
            if k.endswith(".weight_v"):
                out[k[:-len("weight_v")] + "weight_scale"] = torch.ones(1) * scale
    return out


# --------------------------------------------------------------------------------------
# primitive layers (offline, `conv.py`)
# --------------------------------------------------------------------------------------
def extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int) -> int:
    """`conv.py:61-68`."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


def sconv1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1, groups: int = 1,
            causal: bool = True, dilation: int = 1) -> Tensor:
    """`SConv1d.forward` `conv.py:222-236` (pad_mode 'constant')."""
    k = w.shape[-1]
    padding_total = (k - 1) * dilation - (stride - 1)
    extra = extra_padding_for_conv1d(x.shape[-1], k, stride, padding_total)
    if causal:
        x = F.pad(x, (padding_total, extra))
    else:
        pr = padding_total // 2
        x = F.pad(x, (padding_total - pr, pr + extra))
    return F.conv1d(x, w, b, stride=stride, dilation=dilation, groups=groups)


def sconvtr1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, groups: int) -> Tensor:
    """`SConvTranspose1d.forward` `conv.py:260-282`, causal, trim_right_ratio 1."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
    padding_total = k - stride
    return y[..., : y.shape[-1] - padding_total]


def causal_stft_mag(wav: Tensor, basis: Tensor, hop: int, pad: bool, clamp: bool) -> Tensor:
    """Offline `CausalSTFT.forward` `conv.py:348-358` (pad=True, clamp_min(1e-12) before sqrt) and
    streaming `CausalSTFT.forward` `causal_layers.py:135-144` (no pad, no clamp)."""
    n_fft = basis.shape[-1]
    if pad:
        wav = F.pad(wav, (n_fft - 1, 0))
    y = F.conv1d(wav, basis, None, stride=hop)
    B, C, T = y.shape
    y = y.view(B, 2, C // 2, T).square().sum(dim=1)
    if clamp:
        y = y.clamp_min(1e-12)
    return y.sqrt()


def elu(x: Tensor) -> Tensor:
    return F.elu(x, alpha=1.0)


# --------------------------------------------------------------------------------------
# offline encoder / decoder (`seanet.py`)
# --------------------------------------------------------------------------------------
def resblock(sd: SD, prefix: str, x: Tensor, res_scale: float, idx: Optional[int]) -> Tensor:
    """`SEANetResnetBlock.forward` `seanet.py:129-148` with skip='identity'.
    idx=None reproduces the streaming decoder's `pre_scale = 1`."""
    pre_scale = (1 + idx * res_scale ** 2) ** -0.5 if idx is not None else 1.0
    y = x * pre_scale
    for pw, dw in (("1", "2"), ("4", "5")):
        w, b = conv_weight(sd, f"{prefix}.block.{pw}.conv.conv")
        y = F.conv1d(elu(y), w, b)
        w, b = conv_weight(sd, f"{prefix}.block.{dw}.conv.conv")
        y = sconv1d(y, w, b, groups=w.shape[0])
    scale = res_scale * sd[f"{prefix}.res_scale_param"]
    return y * scale + x


def spec_block(sd: SD, prefix: str, x: Tensor, wav: Tensor, hop: int, mean: float, std: float,
               res_scale: float) -> Tensor:
    """`SpecBlock.forward` `seanet.py:220-246` (spec='stft', compression='log', inout_norm)."""
    y = causal_stft_mag(wav, sd[f"{prefix}.spec.weight"], hop, pad=True, clamp=True)
    y = y.clamp_min(1e-5).log()
    y = (y - mean) / std
    w, b = conv_weight(sd, f"{prefix}.layer.conv.conv")
    y = F.conv1d(y, w, b)
    scale = sd[f"{prefix}.scale_param"] * res_scale
    return x + y * scale


def encoder_forward(sd: SD, x: Tensor, mk: dict) -> Tensor:
    """`SEANetEncoder.forward` `seanet.py:368-378`; x `[B,1,T]` -> `[B,dim,T/320]`."""
    rs = mk["res_scale_enc"]
    nres = mk["n_residual_enc"]
    ratios = list(reversed(mk["strides"]))
    wav = x
    w, b = conv_weight(sd, "encoder.conv_pre.1.conv.conv")
    x = sconv1d((1 / WAV_STD) * x, w, b)                               # seanet.py:280-286
    hop = 1
    for s, r in enumerate(ratios):
        x = spec_block(sd, f"encoder.spec_blocks.{s}", x, wav, hop, SPEC_MEANS[s], SPEC_STDS[s], rs)
        for j in range(1, nres + 1):                                   # idx = j (spec != "")
            x = resblock(sd, f"encoder.blocks.{s}.{j - 1}", x, rs, j)
        hop *= r
        x = x * (1 + nres * rs ** 2) ** -0.5                           # seanet.py:322-340
        w, b = conv_weight(sd, f"encoder.downsample.{s}.2.conv.conv")
        x = F.conv1d(elu(x), w, b)
        w, b = conv_weight(sd, f"encoder.downsample.{s}.3.conv.conv")
        x = sconv1d(x, w, b, stride=r, groups=w.shape[0])
    x = spec_block(sd, "encoder.spec_post", x, wav, hop, SPEC_MEANS[-1], SPEC_STDS[-1], rs)
    w, b = conv_weight(sd, "encoder.conv_post.1.conv.conv")            # seanet.py:350-358
    x = sconv1d(elu(x), w, b, groups=w.shape[0])
    w, b = conv_weight(sd, "encoder.conv_post.2.conv.conv")
    x = F.conv1d(x, w, b)
    if mk.get("encoder_l2norm", True):                                 # L2Norm seanet.py:151-162
        x = F.normalize(x, p=2.0, dim=1, eps=1e-12) * (x.shape[1] ** 0.5)
    return x


def decoder_forward(sd: SD, z: Tensor, mk: dict, streaming_variant: bool = False) -> Tensor:
    """`SEANetDecoder.forward` `seanet.py:477-479` (nn.Sequential built at :409-475).

    streaming_variant=True reproduces the two *decoder* deviations of the reference's streaming
    model: (a) `pre_scale = 1` in every decoder ResBlock (`streaming.py:576-583`),
    (b) only the final conv's weight — not its bias — is scaled by wav_std (`streaming.py:609-617`)."""
    rs = mk["res_scale_dec"]
    nres = mk["n_residual_dec"]
    w, b = conv_weight(sd, "decoder.model.0.conv.conv")
    x = F.conv1d(z, w, b)
    w, b = conv_weight(sd, "decoder.model.1.conv.conv")
    x = sconv1d(x, w, b, groups=w.shape[0])
    idx = 2
    stage_scale = (1 + nres * rs ** 2) ** -0.5
    for i, r in enumerate(mk["strides"]):
        if i > 0:
            x = x * stage_scale
        idx += 2
        w, b = conv_weight(sd, f"decoder.model.{idx}.convtr.convtr")
        x = sconvtr1d(elu(x), w, b, stride=r, groups=w.shape[0])
        idx += 1
        w, b = conv_weight(sd, f"decoder.model.{idx}.conv.conv")
        x = F.conv1d(x, w, b)
        idx += 1
        for j in range(nres):
            x = resblock(sd, f"decoder.model.{idx}", x, rs, None if streaming_variant else j)
            idx += 1
    idx += 2
    x = elu(x * stage_scale)
    w, b = conv_weight(sd, f"decoder.model.{idx}.conv.conv")
    if streaming_variant:
        x = sconv1d(x, w * WAV_STD, b)
    else:
        x = sconv1d(x, w, b) * WAV_STD                                 # seanet.py:462-466
    if mk.get("final_activation", "Tanh") == "Tanh":
        x = torch.tanh(x)
    return x


# --------------------------------------------------------------------------------------
# residual VQ (eval branch)
# --------------------------------------------------------------------------------------
def codebook_argmin(flat: Tensor, embed: Tensor) -> Tensor:
    """`EuclideanCodebook.forward` eval branch, `models/hilcodec/vector_quantize.py:132-152`:
    `dist = (-2*flat) @ embed.T + sum(embed**2)`, argmin = first minimum."""
    e = embed.t()
    distance = -2 * flat @ e + e.pow(2).sum(0, keepdim=True)
    return distance.min(dim=-1).indices


def codebook_argmax_neg(flat: Tensor, embed: Tensor) -> Tensor:
    """Older form, `modules/vector_quantize.py:141-160` and `streaming.py:51-65`:
    `dist = -(|x|^2 - 2 x@e + |e|^2)`, argmax = first maximum."""
    e = embed.t()
    distance = -(flat.pow(2).sum(1, keepdim=True) - 2 * flat @ e + e.pow(2).sum(0, keepdim=True))
    return distance.max(dim=-1).indices


def rvq_forward(sd: SD, z: Tensor, n: Optional[int], num_quantizers: int,
                variant: str = "hilcodec", prefix: str = "quantizer.layers.{i}.embed"
                ) -> Tuple[Tensor, np.ndarray, Tensor, Tensor]:
    """`ResidualVQ.forward` `models/hilcodec/vector_quantize.py:199-243` (eval, channel_last=False).
    Returns `(quantized [B,C,T], num_replaces int64[Nq], mse loss, indices [B,n,T] int64)`.
    variant='legacy' uses the `modules/vector_quantize.py` distance/argmax form (same indices
    except at exact fp ties)."""
    if n is not None:
        assert 1 <= n <= num_quantizers, f"'n' must be in range of 1 <= n <= {num_quantizers}"
    high = n if n is not None else num_quantizers
    residual = z.transpose(1, 2)
    shape = residual.shape
    out = None
    indices = []
    pick = codebook_argmin if variant == "hilcodec" else codebook_argmax_neg
    for i in range(high):
        embed = sd[prefix.format(i=i)]
        ind = pick(residual.reshape(-1, shape[-1]), embed).view(*shape[:-1])
        q = F.embedding(ind, embed)
        indices.append(ind)
        residual = residual - q
        out = q if out is None else out + q
    out = out.transpose(1, 2)
    loss = F.mse_loss(z, out)
    return out, np.zeros(num_quantizers, dtype=np.int64), loss, torch.stack(indices, dim=1)


def rvq_train_step(state: Dict[str, Tensor], z: Tensor, n: Optional[int], num_quantizers: int, decay: float,
                   ema_num_threshold: float = 0.0, bucket_hook=None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Training branch of `EuclideanCodebook.forward` / `ResidualVQ.forward`
    (`models/hilcodec/vector_quantize.py:132-176,199-243`, channel_last=False), single process, expiry only
    reported (the replacement vectors are random: `:21-29,101-112`).  `state` holds `layers.{i}.embed`,
    `layers.{i}.ema_embed`, `layers.{i}.ema_num` and is updated IN PLACE like the module's buffers.
    `bucket_hook(bucket)` stands for `dist.all_reduce(bucket)` (`:158-162`).
    Returns `(quantized [B,C,T] (values of the straight-through output), loss, indices [B,n,T], expired masks [n,K])`."""
    high = n if n is not None else num_quantizers
    residual = z.transpose(1, 2)
    shape = residual.shape
    out = None
    indices, expired = [], []
    for i in range(high):
        embed = state[f"layers.{i}.embed"]
        flatten = residual.reshape(-1, shape[-1])
        ind = codebook_argmin(flatten, embed)
        onehot = F.one_hot(ind, embed.shape[0]).type(embed.dtype)
        q = F.embedding(ind.view(*shape[:-1]), embed)
        num_curr = onehot.sum(dim=0)
        embed_curr = onehot.t() @ flatten.float()
        if bucket_hook is not None:
            bucket = torch.cat([num_curr, embed_curr.view(-1)])
            bucket_hook(bucket)
            num_curr = bucket[:embed.shape[0]]
            embed_curr = bucket[embed.shape[0]:].reshape(embed.shape)
        state[f"layers.{i}.ema_num"].mul_(decay).add_(num_curr, alpha=(1 - decay))
        state[f"layers.{i}.ema_embed"].mul_(decay).add_(embed_curr, alpha=(1 - decay))
        embed.copy_(state[f"layers.{i}.ema_embed"] / state[f"layers.{i}.ema_num"].unsqueeze(1))
        expired.append(state[f"layers.{i}.ema_num"] < ema_num_threshold if ema_num_threshold != 0.0
                       else torch.zeros_like(state[f"layers.{i}.ema_num"], dtype=torch.bool))
        indices.append(ind.view(*shape[:-1]))
        residual = residual - q
        out = q if out is None else out + q
    out = out.transpose(1, 2)
    loss = F.mse_loss(z, out)
    out = out + z - z.detach()            # straight-through estimator (`:234-235`): (q + x) - x in fp32
    return out, loss, torch.stack(indices, dim=1), torch.stack(expired)


def legacy_rvq_train_step(state: Dict[str, Tensor], z: Tensor, n: Optional[int], num_quantizers: int, decay: float,
                          eps: float = 1e-7, ema_num_threshold: float = 0.0, bucket_hook=None
                          ) -> Tuple[Tensor, Tensor, Tensor]:
    """Training branch of the OLDER stack: `EuclideanCodebook.forward` `modules/vector_quantize.py:141-195` inside
    `ResidualVQ.forward` `:487-516` (channel_last=False, `gradient_flow=False` layers), single process, expiry only
    reported.  Differences to `rvq_train_step`: argmax of the negated distance (`:152-158`), and with
    `ema_num_threshold <= 0` the counts are Laplace-smoothed before the division (`:183-190`).
    `state`: `layers.{i}.embed / ema_embed / ema_num`, updated in place.
    Returns `(values of the straight-through output [B,C,T], loss, expired masks [n,K])`."""
    high = n if n is not None else num_quantizers
    residual = z.detach()
    out = 0.
    expired = []
    for i in range(high):
        embed = state[f"layers.{i}.embed"]
        x = residual.transpose(1, 2)                                  # VectorQuantize: 'b c t -> b t c' (`:402-404`)
        flatten = x.reshape(-1, x.shape[-1])
        ind = codebook_argmax_neg(flatten, embed)
        onehot = F.one_hot(ind, embed.shape[0]).type(embed.dtype)
        q = F.embedding(ind.view(*x.shape[:-1]), embed)
        num_new = onehot.sum(dim=0)
        embed_new = flatten.t().float() @ onehot                      # [C, K] (`:178`)
        if bucket_hook is not None:
            bucket = torch.cat([num_new, embed_new.reshape(-1)])
            bucket_hook(bucket)
            num_new = bucket[:embed.shape[0]]
            embed_new = bucket[embed.shape[0]:].reshape(embed.shape[1], embed.shape[0])
        ema_num, ema_embed = state[f"layers.{i}.ema_num"], state[f"layers.{i}.ema_embed"]
        ema_num.mul_(decay).add_(num_new, alpha=(1 - decay))
        ema_embed.mul_(decay).add_(embed_new.t(), alpha=(1 - decay))
        if ema_num_threshold <= 0.0:
            denom = (ema_num + eps) / (ema_num.sum() + embed.shape[0] * eps) * ema_num.sum()
        else:
            denom = ema_num
        embed.copy_(ema_embed / denom.unsqueeze(1))
        expired.append(ema_num < ema_num_threshold if ema_num_threshold != 0.0
                       else torch.zeros_like(ema_num, dtype=torch.bool))
        q = q.transpose(1, 2)
        residual = residual - q
        out = out + q
    loss = F.mse_loss(z, out)
    return out + z - z.detach(), loss, torch.stack(expired)


def rvq_gaps_fp64(sd: SD, z: Tensor, indices: Tensor, prefix: str = "quantizer.layers.{i}.embed") -> Tensor:
    """fp64 best-vs-second-best distance gap per (b, stage, t) along the *given* index path
    (used by the parity tests to tell a genuine mismatch from a sub-ulp near-tie)."""
    residual = z.transpose(1, 2).double()
    gaps = []
    for i in range(indices.shape[1]):
        embed = sd[prefix.format(i=i)].double()
        d = torch.cdist(residual.reshape(-1, residual.shape[-1]), embed).pow(2)
        top2 = d.topk(2, dim=-1, largest=False).values
        gaps.append((top2[:, 1] - top2[:, 0]).view(residual.shape[:-1]))
        residual = residual - F.embedding(indices[:, i], embed)
    return torch.stack(gaps, dim=1)


def dequantize(sd: SD, indices: Tensor, n: int, prefix: str = "quantizer.layers.{i}.embed") -> Tensor:
    """`Dequantizer.forward` `streaming.py:148-157`: indices `[n,B,T]` -> `[B,T,C]`."""
    out = torch.zeros(1, dtype=torch.float32)
    for i in range(n):
        out = out + F.embedding(indices[i], sd[prefix.format(i=i)])
    return out


def codec_forward(sd: SD, x: Tensor, mk: dict, n: Optional[int] = None):
    """`HILCodec.forward` `models/hilcodec/models.py:111-118`."""
    z = encoder_forward(sd, x, mk)
    q, num_replaces, loss, idx = rvq_forward(sd, z, n, mk["vq_kwargs"]["num_quantizers"])
    wav = decoder_forward(sd, q, mk)
    return wav.float(), num_replaces, loss, dict(z=z, q=q, indices=idx)


# --------------------------------------------------------------------------------------
# streaming model (`streaming.py`, `causal_layers.py`)
# --------------------------------------------------------------------------------------
def stream_prepare(sd: SD, mk: dict) -> SD:
    """Offline state-dict -> merged streaming parameters: the mapping of
    `scripts/HILCodec Onnx.ipynb` cell 1 followed by `remove_weight_reparameterizations`
    (`streaming.py:740-747`): remove_weight_norm, then every `merge_scaling`
    (`:472-480` conv_pre /= wav_std, `:321-344` SpecBlock, `:240-250` ResBlock, `:609-617`)."""
    p: SD = {}
    rs_e, rs_d = mk["res_scale_enc"], mk["res_scale_dec"]
    ratios = list(reversed(mk["strides"]))

    def put(dst: str, src: str):
        w, b = conv_weight(sd, src)
        p[dst + ".weight"] = w.clone()
        if b is not None:
            p[dst + ".bias"] = b.clone()

    def put_resblock(dst: str, src: str, rs: float):
        put(dst + ".0.pw", src + ".block.1.conv.conv")
        put(dst + ".0.dw", src + ".block.2.conv.conv")
        put(dst + ".1.pw", src + ".block.4.conv.conv")
        put(dst + ".1.dw", src + ".block.5.conv.conv")
        scale = rs * sd[src + ".res_scale_param"]
        p[dst + ".1.dw.weight"].mul_(scale)
        p[dst + ".1.dw.bias"].mul_(scale)

    def put_spec(dst: str, src: str, mean: float, std: float):
        put(dst + ".layer", src + ".layer.conv.conv")
        w = p[dst + ".layer.weight"]
        bias2 = w.sum((1, 2)).mul(-mean / std)
        w.div_(std)
        scale = rs_e * sd[src + ".scale_param"]
        w.mul_(scale)
        p[dst + ".layer.bias"] = bias2.mul_(scale)
        p[dst + ".basis"] = sd[src + ".spec.weight"]

    put("enc.conv_pre", "encoder.conv_pre.1.conv.conv")
    p["enc.conv_pre.weight"].div_(WAV_STD)
    for s in range(len(ratios)):
        for j in range(mk["n_residual_enc"]):
            put_resblock(f"enc.blocks.{s}.{j}", f"encoder.blocks.{s}.{j}", rs_e)
        put_spec(f"enc.spec.{s}", f"encoder.spec_blocks.{s}", SPEC_MEANS[s], SPEC_STDS[s])
        put(f"enc.down.{s}.pw", f"encoder.downsample.{s}.2.conv.conv")
        put(f"enc.down.{s}.dw", f"encoder.downsample.{s}.3.conv.conv")
    put_spec("enc.spec_post", "encoder.spec_post", SPEC_MEANS[-1], SPEC_STDS[-1])
    put("enc.post.dw", "encoder.conv_post.1.conv.conv")
    put("enc.post.pw", "encoder.conv_post.2.conv.conv")

    put("dec.pre.pw", "decoder.model.0.conv.conv")
    put("dec.pre.dw", "decoder.model.1.conv.conv")
    idx = 2
    for i in range(len(mk["strides"])):
        idx += 2
        put(f"dec.up.{i}.dw", f"decoder.model.{idx}.convtr.convtr"); idx += 1
        put(f"dec.up.{i}.pw", f"decoder.model.{idx}.conv.conv"); idx += 1
        for j in range(mk["n_residual_dec"]):
            put_resblock(f"dec.blocks.{i}.{j}", f"decoder.model.{idx}", rs_d); idx += 1
    idx += 2
    put("dec.post", f"decoder.model.{idx}.conv.conv")
    p["dec.post.weight"].mul_(WAV_STD)
    for i in range(mk["vq_kwargs"]["num_quantizers"]):
        p[f"vq.{i}.embed"] = sd[f"quantizer.layers.{i}.embed"]
    return p


def causal_conv1d(x: Tensor, cache: Tensor, w: Tensor, b: Optional[Tensor], stride: int, groups: int):
    """`CausalConv1d.forward` `causal_layers.py:160-165`."""
    x = torch.cat((cache, x), dim=2)
    pad = (w.shape[-1] - 1) - (stride - 1)
    cache = x[:, :, -pad:]
    return F.conv1d(x, w, b, stride, 0, 1, groups), cache


def causal_convtr1d(x: Tensor, cache: Tensor, w: Tensor, b: Optional[Tensor], stride: int, groups: int):
    """`CausalConvTranspose1d.forward` `causal_layers.py:168-188` (k = 2*stride: causal_padding 1,
    padding = stride, output_padding = 0)."""
    k = w.shape[-1]
    rf = k - 1
    cpad = rf // stride
    padding = cpad * stride
    out_pad = stride - 1 + padding - rf
    x = torch.cat([cache, x], dim=2)
    cache = x[:, :, -cpad:]
    return F.conv_transpose1d(x, w, b, stride, padding, out_pad, groups, 1), cache


def stream_cache_shapes(mk: dict) -> Tuple[List[Tuple[int, int]], List[Tuple[int, int]]]:
    """(channels, length) of every encoder / decoder cache, in the order of
    `Encoder.initialize_cache` `streaming.py:458-470` and `Decoder.initialize_cache` `:599-607`."""
    ce, cd = mk["channels_enc"], mk["channels_dec"]
    ratios = list(reversed(mk["strides"]))
    nfft_post = mk["n_fft_base"] * 2 ** len(ratios)
    enc = [(1, nfft_post - 1)]
    c = ce
    for r in ratios:
        for _ in range(mk["n_residual_enc"]):
            enc += [(c, mk["residual_kernel_size"] - 1)] * 2
        enc.append((2 * c, 2 * r - 1 - (r - 1)))
        c *= 2
    enc.append((c, mk["last_kernel_size"] - 1))
    c = cd * 2 ** len(ratios)
    dec = [(c, mk["kernel_size"] - 1)]
    for r in mk["strides"]:
        dec.append((c, 1))
        for _ in range(mk["n_residual_dec"]):
            dec += [(c // 2, mk["residual_kernel_size"] - 1)] * 2
        c //= 2
    dec.append((c, mk["last_kernel_size"] - 1))
    return enc, dec


def stream_init_cache(mk: dict, batch: int) -> Tuple[List[Tensor], List[Tensor]]:
    enc, dec = stream_cache_shapes(mk)
    return ([torch.zeros(batch, c, l) for c, l in enc], [torch.zeros(batch, c, l) for c, l in dec])


def _stream_resblock(p: SD, prefix: str, x: Tensor, caches: Sequence[Tensor], pre_scale: float):
    """`ResBlock.forward` `streaming.py:252-276` (merged)."""
    skip = x
    x = x * pre_scale
    new = []
    for i in range(2):
        x = F.conv1d(elu(x), p[f"{prefix}.{i}.pw.weight"])
        w = p[f"{prefix}.{i}.dw.weight"]
        x, c = causal_conv1d(x, caches[i], w, p[f"{prefix}.{i}.dw.bias"], 1, w.shape[0])
        new.append(c)
    return x + skip, new


def _stream_spec(p: SD, prefix: str, x: Tensor, wav: Tensor, hop: int) -> Tensor:
    """`SpecBlock.forward` `streaming.py:346-365` (merged)."""
    y = causal_stft_mag(wav, p[prefix + ".basis"], hop, pad=False, clamp=False)
    y = y.clamp_min(1e-5).log()
    y = F.conv1d(y, p[prefix + ".layer.weight"], p[prefix + ".layer.bias"])
    return y + x


def stream_encoder(p: SD, mk: dict, x: Tensor, cache_in: Sequence[Tensor]):
    """`Encoder.forward` `streaming.py:482-517`: x `[B,1,320m]` -> (`[B,m,dim]`, caches)."""
    rs = mk["res_scale_enc"]
    nres = mk["n_residual_enc"]
    ratios = list(reversed(mk["strides"]))
    cache_out: List[Tensor] = []
    wav_cache_len = cache_in[0].shape[-1]
    wav = torch.cat((cache_in[0], x), dim=2)
    cache_out.append(wav[:, :, -wav_cache_len:])
    k_pre = p["enc.conv_pre.weight"].shape[-1]
    x = F.conv1d(wav[:, :, wav_cache_len - (k_pre - 1):], p["enc.conv_pre.weight"], p["enc.conv_pre.bias"])
    idx = 1
    hop = 1
    scale = (1 + nres * rs ** 2) ** -0.5
    for s, r in enumerate(ratios):
        n_fft = p[f"enc.spec.{s}.basis"].shape[-1]
        x = _stream_spec(p, f"enc.spec.{s}", x, wav[:, :, wav_cache_len - (n_fft - 1):], hop)
        for j in range(1, nres + 1):
            x, c = _stream_resblock(p, f"enc.blocks.{s}.{j - 1}", x, cache_in[idx:idx + 2],
                                    (1 + j * rs ** 2) ** -0.5)
            cache_out.extend(c)
            idx += 2
        hop *= r
        x = F.conv1d(elu(x * scale), p[f"enc.down.{s}.pw.weight"])
        w = p[f"enc.down.{s}.dw.weight"]
        x, c = causal_conv1d(x, cache_in[idx], w, p[f"enc.down.{s}.dw.bias"], r, w.shape[0])
        cache_out.append(c)
        idx += 1
    x = _stream_spec(p, "enc.spec_post", x, wav, hop)
    w = p["enc.post.dw.weight"]
    x, c = causal_conv1d(elu(x), cache_in[idx], w, None, 1, w.shape[0])
    cache_out.append(c)
    x = F.conv1d(x, p["enc.post.pw.weight"], p["enc.post.pw.bias"])
    x = F.normalize(x, p=2.0, dim=1, eps=1e-12) * (x.shape[1] ** 0.5)
    return x.transpose(1, 2), cache_out


def stream_quantize(p: SD, x: Tensor, n: int) -> Tensor:
    """`ResidualVQ.forward` `streaming.py:89-100`: x `[B,T,C]` -> indices `[n,B,T]` int64."""
    residual = x
    B, T, C = x.shape
    indices = []
    for i in range(n):
        embed = p[f"vq.{i}.embed"]
        ind = codebook_argmax_neg(residual.reshape(B * T, C), embed).view(B, T)
        residual = residual - F.embedding(ind, embed)
        indices.append(ind)
    return torch.stack(indices, dim=0)


def stream_dequantize(p: SD, indices: Tensor, n: int) -> Tensor:
    return dequantize(p, indices, n, prefix="vq.{i}.embed")


def stream_decoder(p: SD, mk: dict, q: Tensor, cache_in: Sequence[Tensor]):
    """`Decoder.forward` `streaming.py:619-648`: q `[B,m,dim]` -> (`[B,1,320m]`, caches)."""
    rs = mk["res_scale_dec"]
    nres = mk["n_residual_dec"]
    x = q.transpose(1, 2)
    cache_out: List[Tensor] = []
    x = F.conv1d(x, p["dec.pre.pw.weight"])
    w = p["dec.pre.dw.weight"]
    x, c = causal_conv1d(x, cache_in[0], w, p["dec.pre.dw.bias"], 1, w.shape[0])
    cache_out.append(c)
    idx = 1
    scale = (1 + nres * rs ** 2) ** -0.5
    for i, r in enumerate(mk["strides"]):
        w = p[f"dec.up.{i}.dw.weight"]
        x, c = causal_convtr1d(elu(x), cache_in[idx], w, None, r, w.shape[0])
        x = F.conv1d(x, p[f"dec.up.{i}.pw.weight"], p[f"dec.up.{i}.pw.bias"])
        cache_out.append(c)
        idx += 1
        for j in range(nres):
            x, c = _stream_resblock(p, f"dec.blocks.{i}.{j}", x, cache_in[idx:idx + 2], 1.0)
            cache_out.extend(c)
            idx += 2
        x = x * scale
    x, c = causal_conv1d(elu(x), cache_in[idx], p["dec.post.weight"], p["dec.post.bias"], 1, 1)
    cache_out.append(c)
    return torch.tanh(x), cache_out
