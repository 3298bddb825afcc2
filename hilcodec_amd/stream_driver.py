"""Streaming driver with the protocol of the reference's `test_onnx.py`, on the
MI355X kernels instead of ONNXRuntime sessions:

  encoder pass : waveform -> hop-sized chunks (hop = 320 * num_frames) -> encoder(+caches) -> RVQ
                 -> indices int16 `[n, B, T]` saved as `{name}_quantized.npy`        (test_onnx.py:50-100)
  decoder pass : indices -> `num_frames` at a time -> Dequantizer -> decoder(+caches) -> waveform
                                                                                     (test_onnx.py:103-139)
  StageClock   : per-stage RTF = audio seconds / stage seconds (HIP events)          (test_onnx.py:20-47)

    python -m hilcodec_amd.stream_driver -n hil_speech -q 8 -f 1 --enc --dec --input in.wav --outdir out/

`--input` is a 16-bit PCM mono WAV (read with the stdlib `wave` module) or a float32 `.npy`; weights come
from `--checkpoint` (the reference's `NNNNN.pth`, key 'model') or, without it, from the deterministic synthetic
generator used by the tests (there are no trained encoder/decoder weights in the reference tree)."""
from __future__ import annotations

import argparse
import contextlib
import os
import wave
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import synth, wire


class StageClock:
    """Per-stage real-time factor of a streaming run — what `test_onnx.py:20-47` reports (RTF = audio seconds per
    wall second of a stage), measured the GPU way: a stage is bracketed by two HIP events on the launch stream and
    the host only synchronises once, when the figures are read, so timing a run does not serialise its hops."""

    STAGES = ("encoder", "decoder")

    def __init__(self, sample_rate: int):
        self.sample_rate = sample_rate
        self.samples = 0                      # audio samples the run covered (set by encode_stream / decode_stream)
        self._spans = {s: [] for s in self.STAGES}

    class _Span:
        def __init__(self, spans):
            self._spans = spans

        def __enter__(self):
            self.begin = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.begin.record()
            return self

        def __exit__(self, *exc):
            self.end.record()
            self._spans.append((self.begin, self.end))
            return False

    def stage(self, name: str) -> "StageClock._Span":
        return StageClock._Span(self._spans[name])

    def seconds(self, name: str) -> float:
        spans = self._spans[name]
        if spans:
            spans[-1][1].synchronize()
        return sum(b.elapsed_time(e) for b, e in spans) * 1e-3

    def report(self) -> dict:
        audio = self.samples / self.sample_rate
        out = {"wav_seconds": audio}
        for name in self.STAGES:
            sec = self.seconds(name)
            if sec > 0:
                out[f"{name}_seconds"] = sec
                out[f"{name}_rtf"] = audio / sec
        return out

    def summary(self) -> str:
        r = self.report()
        lines = [f"audio {r['wav_seconds']:.2f} s"]
        for name in self.STAGES:
            if f"{name}_rtf" in r:
                lines.append(f"{name:8s} {r[name + '_seconds'] * 1e3:9.2f} ms   {r[name + '_rtf']:10.1f} x real time")
        return "\n".join(lines)


def read_wav(path: str, sr: int) -> np.ndarray:
    """16-bit PCM mono WAV -> float32 in [-1, 1) (what librosa.load returns for such a file at its native rate)."""
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32).reshape(-1)
    with wave.open(path, "rb") as w:
        if w.getnchannels() != 1 or w.getsampwidth() != 2:
            raise ValueError("expected 16-bit mono PCM")
        if w.getframerate() != sr:
            raise ValueError(f"sample rate {w.getframerate()} != {sr} (no resampler on this path)")
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return (pcm.astype(np.float32) / 32768.0)


def write_wav(path: str, wav: np.ndarray, sr: int) -> None:
    pcm = np.clip(np.round(wav * 32767.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


@torch.no_grad()
def encode_stream(model, wav: Tensor, num_quantizers: int, num_frames: int = 1, hop_size: int = 320,
                  cache_enc: Optional[Sequence[Tensor]] = None, timer: Optional[StageClock] = None
                  ) -> Tuple[Tensor, List[Tensor]]:
    """wav `[B,1,L]` on the GPU -> (indices int16 `[n,B,L//320]`, final encoder caches).  The tail that does
    not fill a chunk is dropped, like `length = len(wav) // hop_size * hop_size` (test_onnx.py:53)."""
    hop = hop_size * num_frames
    length = wav.shape[-1] // hop * hop
    wav = wav[:, :, :length].contiguous()
    cache = list(cache_enc) if cache_enc is not None else model.encoder.initialize_cache(wav)
    chunks = []
    if timer:
        timer.samples = length
    with (timer.stage("encoder") if timer else contextlib.nullcontext()):
        for i in range(0, length, hop):
            x, cache = model.encoder(wav[:, :, i:i + hop].contiguous(), *cache)
            chunks.append(model.quantizer(x, num_quantizers))            # [n,B,F]
    idx = torch.cat(chunks, dim=2) if chunks else torch.zeros(num_quantizers, wav.shape[0], 0, dtype=torch.int64)
    return idx.to(torch.int16), cache


@torch.no_grad()
def decode_stream(model, indices: Tensor, num_quantizers: int, num_frames: int = 1,
                  cache_dec: Optional[Sequence[Tensor]] = None, timer: Optional[StageClock] = None
                  ) -> Tuple[Tensor, List[Tensor]]:
    """indices `[n,B,T]` (int16 or int64) -> (wav `[B,1,320*T]`, final decoder caches)."""
    dev = next(model.parameters()).device
    indices = indices.to(dev)
    cache = list(cache_dec) if cache_dec is not None else model.decoder.initialize_cache(
        torch.zeros(indices.shape[1], 1, device=dev))
    outs = []
    with (timer.stage("decoder") if timer else contextlib.nullcontext()):
        for i in range(0, indices.shape[2], num_frames):
            q = model.dequantizer(indices[:num_quantizers, :, i:i + num_frames].contiguous(), num_quantizers)
            w, cache = model.decoder(q, *cache)
            outs.append(w)
    wav = torch.cat(outs, dim=2)
    if timer:
        timer.samples = wav.shape[-1]
    return wav, cache


def build_streaming_model(name: str, checkpoint: Optional[str], device) -> torch.nn.Module:
    from .models.hilcodec.streaming import HILCodec
    mk = synth.model_kwargs(name)
    smk = {k: v for k, v in mk.items() if k not in ("spec_learnable", "causal", "pad_mode")}
    model = HILCodec(24000, **smk).eval()
    if checkpoint:
        sd = torch.load(checkpoint, map_location="cpu")
        sd = sd.get("model", sd)
    else:
        sd = synth.synth_state_dict(name, seed=7)
    model.load_offline_state_dict(sd)
    model.remove_weight_reparameterizations()
    return model.to(device)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-n", "--name", default="hil_speech")
    ap.add_argument("-q", "--num_quantizers", type=int, default=8)
    ap.add_argument("-f", "--num_frames", type=int, default=1)
    ap.add_argument("-H", "--hop_size", type=int, default=320)
    ap.add_argument("--enc", action="store_true")
    ap.add_argument("--dec", action="store_true")
    ap.add_argument("--sr", type=int, default=24_000)
    ap.add_argument("--input", default=None)
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--checkpoint", default=None)
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    model = build_streaming_model(a.name, a.checkpoint, dev)
    timer = StageClock(a.sr)
    qpath = os.path.join(a.outdir, f"{a.name}_quantized.npy")
    if a.enc:
        wav = read_wav(a.input, a.sr) if a.input else synth.sweep_clip(a.sr * 2, a.sr).numpy().reshape(-1)
        x = torch.from_numpy(np.clip(wav, -1, 1)).view(1, 1, -1).to(dev)
        idx, _ = encode_stream(model, x, a.num_quantizers, a.num_frames, a.hop_size, timer=timer)
        wire.save_indices_npy(qpath, idx)
    if a.dec:
        idx = wire.load_indices_npy(qpath)
        wav, _ = decode_stream(model, idx, a.num_quantizers, a.num_frames, timer=timer)
        write_wav(os.path.join(a.outdir, f"{a.name}_output.wav"), wav[0, 0].cpu().numpy(), a.sr)
    print(timer.summary())
    return timer.report()


if __name__ == "__main__":
    main()
