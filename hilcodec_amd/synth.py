"""Portable, seed-addressed synthetic data for HILCodec parity tests and benchmarks.

There are no trained encoder/decoder weights in the reference tree
(`/root/reference/.MISSING_LARGE_BLOBS`), so parity and throughput are measured on
deterministic synthetic weights.  Everything here is integer-hash based (splitmix64 on a
`(seed, index)` counter, top 24 bits -> fp32) so that the same numbers come out on every
machine and numpy version: no libm call, no torch/numpy RNG stream.

The state-dict produced by :func:`synth_state_dict` uses the reference's *offline* key layout
(`models/hilcodec/models.py:24` module tree, keys listed in SURVEY.md §5 "Checkpoint"), i.e. the
same thing `torch.load(NNNNN.pth)['model']` holds.
"""
from __future__ import annotations

import zlib
from typing import Dict, List, Optional

import numpy as np
import torch

WAV_STD = 0.1122080159  # models/hilcodec/modules/seanet.py:264

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """`n` fp32 values in [0, 1) addressed by (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([(seed * 0x100000001B3 + stream * 0x9E3779B1 + 0x1234567) & 0xFFFFFFFFFFFFFFFF],
                                    dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64((idx * np.uint64(0xD1342543DE82EF95) + base) & _M64)
    return ((h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))).astype(np.float32)


def uniform(seed: int, n: int, lo: float, hi: float, stream: int = 0) -> np.ndarray:
    u = uniform01(seed, n, stream).astype(np.float64)
    return (lo + (hi - lo) * u).astype(np.float32)


def normalish(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """Unit-variance, zero-mean, bell-shaped fp32 values (Irwin-Hall of 4 uniforms)."""
    acc = np.zeros(n, dtype=np.float64)
    for j in range(4):
        acc += uniform01(seed, n, stream * 4 + j + 101).astype(np.float64)
    return ((acc - 2.0) * (3.0 ** 0.5)).astype(np.float32)  # var of sum = 4/12


def key_seed(seed: int, key: str) -> int:
    return (seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF


def synth_clips(batch: int, samples: int = 24000, seed: int = 1234, first: int = 0) -> torch.Tensor:
    """`[batch, 1, samples]` fp32 clips: clamp(wav_std * N(0,1)-ish, -1, 1), clip i seeded by seed+i.

    SURVEY.md §8(d): the reference normalises by wav_std (`seanet.py:264,280-286`) and its callers
    clamp inputs to +-1 (notebook cell 2)."""
    out = np.empty((batch, 1, samples), dtype=np.float32)
    for i in range(batch):
        x = normalish(seed + first + i, samples) * np.float32(WAV_STD)
        out[i, 0] = np.clip(x, -1.0, 1.0)
    return torch.from_numpy(out)


def sweep_clip(samples: int = 24000, sample_rate: int = 24000) -> torch.Tensor:
    """Deterministic 100 Hz -> 8 kHz exponential sine sweep `[1,1,samples]` (sanity clip)."""
    t = np.arange(samples, dtype=np.float64) / sample_rate
    dur = samples / sample_rate
    f0, f1 = 100.0, 8000.0
    k = (f1 / f0) ** (1.0 / dur)
    phase = 2 * np.pi * f0 * (k ** t - 1.0) / np.log(k)
    return torch.from_numpy((0.25 * np.sin(phase)).astype(np.float32)).view(1, 1, -1)


# --------------------------------------------------------------------------------------
# Model hyper-parameters (configs/hilcodec_speech.yaml:2-38, configs/hilcodec_music.yaml:2-38)
# --------------------------------------------------------------------------------------
def adversarial_clips(samples: int = 24000) -> torch.Tensor:
    """`[6,1,samples]`: digital silence, silence -> signal, +-1 square wave (full-scale clipping), a single impulse,
    DC 0.5, a full-scale 3 kHz sine — the inputs that drive the clamp branches of the log-spectrogram
    (conv.py:357, seanet.py:232: every bin at p = 0) and the ELU's deep-negative / large-positive tails.
    Only the reference's outputs for them are stored (tests/golden/realistic.npz)."""
    t = torch.arange(samples, dtype=torch.float32)
    x = torch.zeros(6, 1, samples)
    x[1, 0, samples // 2:] = torch.from_numpy(normalish(4711, samples - samples // 2)) * 0.1      # silence -> signal
    x[2, 0] = torch.where((t // 37) % 2 == 0, torch.tensor(1.0), torch.tensor(-1.0))                  # +-1 square wave
    x[3, 0, 12345] = 1.0                                                                                 # lone impulse
    x[4, 0] = 0.5                                                                                         # DC
    x[5, 0] = torch.sin(t * (2 * 3.141592653589793 * 3000.0 / 24000.0))                                  # full-scale sine
    return x


def model_kwargs(name: str = "hil_speech") -> dict:
    nq = {"hil_speech": 8, "hil_music": 12}[name]
    dropout_index = {"hil_speech": [2, 4, 8], "hil_music": [2, 4, 8, 12]}[name]
    return dict(
        channels_enc=64, channels_dec=96, n_fft_base=64, n_residual_enc=2, n_residual_dec=3,
        res_scale_enc=0.5773502691896258, res_scale_dec=0.5773502691896258,
        strides=[8, 5, 4, 2], kernel_size=5, last_kernel_size=5, residual_kernel_size=5,
        dilation_base=1, skip="identity", final_activation="Tanh", act_all=False,
        encoder_l2norm=True, causal=True, zero_init=True, inout_norm=True, pad_mode="constant",
        spec="stft", spec_compression="log", spec_learnable=False,
        vq_kwargs=dict(dim=128, codebook_size=1024, num_quantizers=nq, kmeans_init=True,
                       decay=0.99, ema_num_threshold=0.5, ema_num_initial=0.5,
                       dropout=True, dropout_index=dropout_index),
    )


def offline_param_shapes(mk: dict) -> "Dict[str, tuple]":
    """Shapes of every tensor in the reference's offline state-dict, by key, in module order.

    Mirrors the constructors `seanet.py:249-366` (encoder) and `seanet.py:381-475` (decoder) and
    `vector_quantize.py:63-92,179-196`; keys verified against the real reference in
    `tests/test_oracle_vs_reference.py`."""
    ce, cd = mk["channels_enc"], mk["channels_dec"]
    nfft = mk["n_fft_base"]
    k, lk, rk = mk["kernel_size"], mk["last_kernel_size"], mk["residual_kernel_size"]
    dim = mk["vq_kwargs"]["dim"]
    strides: List[int] = list(mk["strides"])
    shapes: Dict[str, tuple] = {}

    def conv(prefix: str, cout: int, cin_per_group: int, ks: int, bias: bool):
        if bias:
            shapes[prefix + ".bias"] = (cout,)
        shapes[prefix + ".weight_g"] = (cout, 1, 1)
        shapes[prefix + ".weight_v"] = (cout, cin_per_group, ks)

    def resblock(prefix: str, c: int):
        shapes[prefix + ".res_scale_param"] = (1,)
        conv(prefix + ".block.1.conv.conv", c, c, 1, False)
        conv(prefix + ".block.2.conv.conv", c, 1, rk, True)
        conv(prefix + ".block.4.conv.conv", c, c, 1, False)
        conv(prefix + ".block.5.conv.conv", c, 1, rk, True)

    # ---- encoder
    conv("encoder.conv_pre.1.conv.conv", ce, 1, k, True)
    ratios = list(reversed(strides))
    mult = 1
    for s, r in enumerate(ratios):
        for j in range(mk["n_residual_enc"]):
            resblock(f"encoder.blocks.{s}.{j}", mult * ce)
        mult *= 2
    mult = 1
    for s, r in enumerate(ratios):
        n = mult * nfft
        shapes[f"encoder.spec_blocks.{s}.scale_param"] = (1,)
        shapes[f"encoder.spec_blocks.{s}.spec.weight"] = (n + 2, 1, n)
        conv(f"encoder.spec_blocks.{s}.layer.conv.conv", mult * ce, n // 2 + 1, 1, False)
        mult *= 2
    mult = 1
    for s, r in enumerate(ratios):
        conv(f"encoder.downsample.{s}.2.conv.conv", 2 * mult * ce, mult * ce, 1, False)
        conv(f"encoder.downsample.{s}.3.conv.conv", 2 * mult * ce, 1, 2 * r, True)
        mult *= 2
    n = mult * nfft
    shapes["encoder.spec_post.scale_param"] = (1,)
    shapes["encoder.spec_post.spec.weight"] = (n + 2, 1, n)
    conv("encoder.spec_post.layer.conv.conv", mult * ce, n // 2 + 1, 1, False)
    conv("encoder.conv_post.1.conv.conv", mult * ce, 1, lk, False)
    conv("encoder.conv_post.2.conv.conv", dim, mult * ce, 1, True)

    # ---- decoder (nn.Sequential indices, seanet.py:409-475)
    mult = 2 ** len(strides)
    conv("decoder.model.0.conv.conv", mult * cd, dim, 1, False)
    conv("decoder.model.1.conv.conv", mult * cd, 1, k, True)
    idx = 2
    for i, r in enumerate(strides):
        idx += 2  # scale_layer, act
        c = mult * cd
        shapes[f"decoder.model.{idx}.convtr.convtr.weight_g"] = (c, 1, 1)
        shapes[f"decoder.model.{idx}.convtr.convtr.weight_v"] = (c, 1, 2 * r)
        idx += 1
        conv(f"decoder.model.{idx}.conv.conv", c // 2, c, 1, True)
        idx += 1
        for j in range(mk["n_residual_dec"]):
            resblock(f"decoder.model.{idx}", c // 2)
            idx += 1
        mult //= 2
    idx += 2  # scale_layer, act
    conv(f"decoder.model.{idx}.conv.conv", 1, cd, lk, True)

    # ---- quantizer
    vq = mk["vq_kwargs"]
    for i in range(vq["num_quantizers"]):
        shapes[f"quantizer.layers.{i}.embed"] = (vq["codebook_size"], vq["dim"])
        shapes[f"quantizer.layers.{i}.ema_embed"] = (vq["codebook_size"], vq["dim"])
        shapes[f"quantizer.layers.{i}.ema_num"] = (vq["codebook_size"],)
    return shapes


def stft_basis(n_fft: int) -> torch.Tensor:
    """The fixed `[n_fft+2, 1, n_fft]` DFT*hann conv basis, built exactly like the reference
    (`models/hilcodec/modules/conv.py:323-345`, `causal_layers.py:109-128`): fp32 torch ops."""
    import math
    window = torch.hann_window(n_fft)
    n = torch.arange(n_fft, dtype=torch.float32).view(1, 1, n_fft)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float32).view(-1, 1, 1)
    cos = torch.cos(-2 * math.pi / n_fft * k * n)
    sin = torch.sin(-2 * math.pi / n_fft * k * n)
    return torch.cat([cos, sin], dim=0) * window


def synth_state_dict(name: str = "hil_speech", seed: int = 7,
                     mk: Optional[dict] = None) -> "Dict[str, torch.Tensor]":
    """Deterministic non-degenerate offline state-dict (reference key layout).

    * `weight_v` ~ U(-1,1); `weight_g` ~ U(0.8,1.2) (weight_norm then gives rows of norm g);
      the 1x1 convs that feed a residual branch keep activations O(1).
    * `bias` ~ U(-0.2,0.2); `res_scale_param`, `scale_param` ~ U(0.5,1) (the reference zero-inits
      them, `seanet.py:124-125,217-218`, which would make every residual branch dead).
    * `embed` ~ 0.3*0.95^i-scaled bell-shaped codebooks (the scale at which a random 1024-entry
      codebook actually reduces the residual of a sqrt(128)-norm vector, stage after stage);
      `ema_*` follow the reference's init relation.
    * decoder transposed-conv and final-conv gains are raised so the decoded waveform has a
      natural amplitude (std ~0.1) instead of shrinking towards 0."""
    mk = mk or model_kwargs(name)
    shapes = offline_param_shapes(mk)
    sd: Dict[str, torch.Tensor] = {}
    ema_init = mk["vq_kwargs"]["ema_num_initial"]
    for key, shp in shapes.items():
        n = int(np.prod(shp))
        ks = key_seed(seed, key)
        if key.endswith("spec.weight"):
            t = stft_basis(shp[2])
        elif key.endswith("weight_v"):
            t = torch.from_numpy(uniform(ks, n, -1.0, 1.0)).view(shp)
        elif key.endswith("weight_g"):
            gain = 1.0
            if ".convtr." in key:
                gain = 2.5      # each output sample only sees 2 of the 2r unit-norm taps
            elif key.startswith("decoder") and shp[0] == 1:
                gain = 4.0      # final 96->1 conv: one unit-norm row over 96*5 taps
            t = torch.from_numpy(uniform(ks, n, 0.8 * gain, 1.2 * gain)).view(shp)
        elif key.endswith(".bias"):
            t = torch.from_numpy(uniform(ks, n, -0.2, 0.2)).view(shp)
        elif key.endswith("scale_param"):
            t = torch.from_numpy(uniform(ks, n, 0.5, 1.0)).view(shp)
        elif key.endswith(".embed"):
            i = int(key.split(".")[2])
            t = torch.from_numpy(normalish(ks, n) * np.float32(0.3 * 0.95 ** i)).view(shp)
        elif key.endswith(".ema_embed"):
            continue
        elif key.endswith(".ema_num"):
            t = torch.full(shp, float(ema_init))
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    for key in list(shapes):
        if key.endswith(".ema_embed"):
            sd[key] = sd[key[:-len("ema_embed")] + "embed"].clone() * ema_init
    # keep module order identical to the reference (ema_embed sits between embed and ema_num)
    return {k: sd[k] for k in shapes}
