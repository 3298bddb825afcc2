"""Streaming driver with the protocol of the reference's `test_onnx.py`, on the
MI355X kernels instead of ONNXRuntime sessions:

  encoder pass : waveform -> hop-sized chunks (hop = 320 * num_frames) -> encoder(+caches) -> RVQ
                 -> indices int16 `[n, B, T]` saved as `{name}_quantized.npy`        (test_onnx.py:50-100)
  decoder pass : indices -> `num_frames` at a time -> Dequantizer -> decoder(+caches) -> waveform
                                                                                     (test_onnx.py:103-139)
  Timer        : per-stage wall time and RTF = audio seconds / wall seconds (↑)      (test_onnx.py:20-47)

    python -m hilcodec_amd.stream_driver -n hil_speech -q 8 -f 1 --enc --dec --input in.wav --outdir out/

`--input` is a 16-bit PCM mono WAV (read with the stdlib `wave` module) or a float32 `.npy`; weights come
from `--checkpoint` (the reference's `NNNNN.pth`, key 'model') or, without it, from the deterministic synthetic
generator used by the tests (there are no trained encoder/decoder weights in the reference tree)."""
from __future__ import annotations

import argparse
import os
import time
import wave
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import synth, wire


class Timer:
    """`test_onnx.py:20-47`."""

    def __init__(self, sr: int):
        self.sr = sr
        self.enc_time = 0.0
        self.dec_time = 0.0
        self.start_time = time.perf_counter()
        self.wav_len = 0

    def tic(self):
        torch.cuda.synchronize()
        self.start_time = time.perf_counter()

    def encoder_time(self):
        torch.cuda.synchronize()
        et = time.perf_counter()
        self.enc_time += et - self.start_time
        self.start_time = et

    def decoder_time(self):
        torch.cuda.synchronize()
        et = time.perf_counter()
        self.dec_time += et - self.start_time
        self.start_time = et

    def report(self) -> dict:
        wav_time = self.wav_len / self.sr
        out = {"wav_seconds": wav_time}
        if self.enc_time > 0:
            out["encoder_seconds"] = self.enc_time
            out["encoder_rtf"] = wav_time / self.enc_time
        if self.dec_time > 0:
            out["decoder_seconds"] = self.dec_time
            out["decoder_rtf"] = wav_time / self.dec_time
        return out

    def print(self):
        r = self.report()
        print(f"\rwav length: {r['wav_seconds']:.1f} s")
        if "encoder_rtf" in r:
            print(f"encoder: {self.enc_time:.1f} s / rtf: {r['encoder_rtf']:.4f} (↑)")
        if "decoder_rtf" in r:
            print(f"decoder: {self.dec_time:.1f} s / rtf: {r['decoder_rtf']:.4f} (↑)")


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
                  cache_enc: Optional[Sequence[Tensor]] = None, timer: Optional[Timer] = None
                  ) -> Tuple[Tensor, List[Tensor]]:
    """wav `[B,1,L]` on the GPU -> (indices int16 `[n,B,L//320]`, final encoder caches).  The tail that does
    not fill a chunk is dropped, like `length = len(wav) // hop_size * hop_size` (test_onnx.py:53)."""
    hop = hop_size * num_frames
    length = wav.shape[-1] // hop * hop
    wav = wav[:, :, :length].contiguous()
    cache = list(cache_enc) if cache_enc is not None else model.encoder.initialize_cache(wav)
    chunks = []
    if timer:
        timer.wav_len = length
        timer.tic()
    for i in range(0, length, hop):
        x, cache = model.encoder(wav[:, :, i:i + hop].contiguous(), *cache)
        chunks.append(model.quantizer(x, num_quantizers))            # [n,B,F]
    if timer:
        timer.encoder_time()
    idx = torch.cat(chunks, dim=2) if chunks else torch.zeros(num_quantizers, wav.shape[0], 0, dtype=torch.int64)
    return idx.to(torch.int16), cache


@torch.no_grad()
def decode_stream(model, indices: Tensor, num_quantizers: int, num_frames: int = 1,
                  cache_dec: Optional[Sequence[Tensor]] = None, timer: Optional[Timer] = None
                  ) -> Tuple[Tensor, List[Tensor]]:
    """indices `[n,B,T]` (int16 or int64) -> (wav `[B,1,320*T]`, final decoder caches)."""
    dev = next(model.parameters()).device
    indices = indices.to(dev)
    cache = list(cache_dec) if cache_dec is not None else model.decoder.initialize_cache(
        torch.zeros(indices.shape[1], 1, device=dev))
    outs = []
    if timer:
        timer.tic()
    for i in range(0, indices.shape[2], num_frames):
        q = model.dequantizer(indices[:num_quantizers, :, i:i + num_frames].contiguous(), num_quantizers)
        w, cache = model.decoder(q, *cache)
        outs.append(w)
    if timer:
        timer.decoder_time()
    wav = torch.cat(outs, dim=2)
    if timer:
        timer.wav_len = wav.shape[-1]
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
    timer = Timer(a.sr)
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
    timer.print()
    return timer.report()


if __name__ == "__main__":
    main()
