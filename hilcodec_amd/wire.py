"""Wire / on-disk formats of the reference's streaming driver (`test_onnx.py`) — SURVEY.md §8(f) row 1.

* code indices: `int16 [n, B, T]` `.npy` (`test_onnx.py:96-100,105`), and a 10-bit-per-index packing
  (1024-entry codebooks = 0.75 kbps per codebook at 75 frames/s, `configs/hilcodec_music.yaml:31`);
* cache templates: `.npz` with `e_in{i}` / `d_in{i}` arrays (`test_onnx.py:71,119`, notebook cell 5);
* trained codebooks: the `embed [1024,128]` fp32 initializer of the reference's `onnx/*_deq{i}.onnx`
  (a single Gather node), read with a minimal protobuf walk — the `onnx` package is not needed.
Host-side helpers only: no arithmetic of the hot path lives here."""
from __future__ import annotations

import struct
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor


# ---------------------------------------------------------------- indices
def save_indices_npy(path: str, indices: Tensor) -> None:
    """indices `[n,B,T]` (any integer dtype, values < 32768) -> int16 .npy, the reference's format."""
    a = indices.detach().cpu().numpy()
    if a.min() < 0 or a.max() > 32767:
        raise ValueError("index out of int16 range")
    np.save(path, a.astype(np.int16))


def load_indices_npy(path: str, device=None) -> Tensor:
    a = np.load(path)
    if a.dtype != np.int16 or a.ndim != 3:
        raise ValueError(f"expected int16 [n,B,T], got {a.dtype} {a.shape}")
    t = torch.from_numpy(a.astype(np.int64))
    return t.to(device) if device is not None else t


def pack_indices_10bit(indices: Tensor) -> bytes:
    """`[n,B,T]` indices in [0,1024) -> header (3 x uint32 LE) + ceil(n*B*T*10/8) bytes, MSB-first."""
    a = indices.detach().cpu().numpy().astype(np.int64)
    if a.ndim != 3 or a.min() < 0 or a.max() >= 1024:
        raise ValueError("indices must be [n,B,T] with values in [0,1024)")
    flat = a.reshape(-1)
    bits = ((flat[:, None] >> np.arange(9, -1, -1)) & 1).astype(np.uint8).reshape(-1)
    return struct.pack("<III", *a.shape) + np.packbits(bits).tobytes()


def unpack_indices_10bit(blob: bytes) -> Tensor:
    n, B, T = struct.unpack("<III", blob[:12])
    count = n * B * T
    bits = np.unpackbits(np.frombuffer(blob[12:], dtype=np.uint8))[: count * 10].reshape(count, 10)
    vals = (bits.astype(np.int64) << np.arange(9, -1, -1)).sum(axis=1)
    return torch.from_numpy(vals.reshape(n, B, T))


# ---------------------------------------------------------------- caches
def save_cache_npz(path: str, caches: Sequence[Tensor], prefix: str) -> None:
    """prefix 'e_in' (encoder, 22 tensors) or 'd_in' (decoder, 30)."""
    np.savez(path, **{f"{prefix}{i}": c.detach().cpu().numpy() for i, c in enumerate(caches)})


def load_cache_npz(path: str, prefix: str, device=None, batch: int = None) -> List[Tensor]:
    z = np.load(path)
    out = []
    i = 0
    while f"{prefix}{i}" in z:
        t = torch.from_numpy(z[f"{prefix}{i}"].astype(np.float32))
        if batch is not None and t.shape[0] != batch:
            t = t.expand(batch, *t.shape[1:]).contiguous()      # templates are saved with batch 1
        out.append(t.to(device) if device is not None else t)
        i += 1
    if not out:
        raise KeyError(f"no '{prefix}0' in {path}")
    return out


# ---------------------------------------------------------------- ONNX initializer reader
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    val = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf: bytes):
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def read_onnx_initializers(path: str) -> Dict[str, np.ndarray]:
    """All fp32 initializers of an ONNX file: ModelProto.graph(7).initializer(5) -> TensorProto
    {dims(1), data_type(2)=1, float_data(4) | raw_data(9), name(8)}."""
    model = open(path, "rb").read()
    out: Dict[str, np.ndarray] = {}
    for num, wt, graph in _fields(model):
        if num != 7 or wt != 2:
            continue
        for gnum, gwt, tensor in _fields(graph):
            if gnum != 5 or gwt != 2:
                continue
            dims: List[int] = []
            dtype, name, raw, floats = None, "", None, None
            for tnum, twt, val in _fields(tensor):
                if tnum == 1 and twt == 0:
                    dims.append(val)
                elif tnum == 1 and twt == 2:               # packed dims
                    p = 0
                    while p < len(val):
                        d, p = _varint(val, p)
                        dims.append(d)
                elif tnum == 2:
                    dtype = val
                elif tnum == 8:
                    name = val.decode()
                elif tnum == 9:
                    raw = val
                elif tnum == 4 and twt == 2:
                    floats = np.frombuffer(val, dtype="<f4")
            if dtype != 1:
                continue
            data = np.frombuffer(raw, dtype="<f4") if raw is not None else floats
            if data is not None:
                out[name] = np.array(data, dtype=np.float32).reshape(dims)
    return out


def read_onnx_codebook(path: str) -> Tensor:
    """The `[K, C]` embed table of one `*_deq{i}.onnx` / `*_vq{i}.onnx` file."""
    inits = read_onnx_initializers(path)
    if "embed" in inits and inits["embed"].ndim == 2:        # the buffer's name in both exported graphs
        return torch.from_numpy(inits["embed"])
    cands = [v for v in inits.values() if v.ndim == 2]
    if len(cands) != 1:
        raise ValueError(f"{path}: no 'embed' initializer and {len(cands)} 2-D fp32 candidates")
    return torch.from_numpy(cands[0])
