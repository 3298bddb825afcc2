"""CPU: wire / on-disk formats of the reference's streaming driver (test_onnx.py) and the ONNX codebook
reader — SURVEY.md §8(f) row 1/2."""
import os

import numpy as np
import pytest
import torch

from hilcodec_amd import synth, wire

REF_ONNX = "/root/reference/onnx"


def test_index_npy_roundtrip(tmp_path):
    idx = torch.randint(0, 1024, (8, 3, 17), generator=torch.Generator().manual_seed(11))
    p = str(tmp_path / "q.npy")
    wire.save_indices_npy(p, idx)
    raw = np.load(p)
    assert raw.dtype == np.int16 and raw.shape == (8, 3, 17)          # test_onnx.py:96-100
    assert torch.equal(wire.load_indices_npy(p), idx)
    bad = idx.clone()
    bad[0, 0, 0] = -1                                                  # explicit out-of-range entries, both ends
    with pytest.raises(ValueError):
        wire.save_indices_npy(p, bad)
    bad[0, 0, 0] = 1 << 15
    with pytest.raises(ValueError):
        wire.save_indices_npy(p, bad)


def test_10bit_packing():
    idx = torch.randint(0, 1024, (12, 2, 75), generator=torch.Generator().manual_seed(12))
    blob = wire.pack_indices_10bit(idx)
    assert len(blob) == 12 + (12 * 2 * 75 * 10 + 7) // 8               # 0.75 kbps per codebook at 75 frames/s
    assert torch.equal(wire.unpack_indices_10bit(blob), idx)
    edge = torch.tensor([[[0, 1023, 512, 1]]])
    assert torch.equal(wire.unpack_indices_10bit(wire.pack_indices_10bit(edge)), edge)
    with pytest.raises(ValueError):
        wire.pack_indices_10bit(torch.tensor([[[1024]]]))


def test_cache_npz_roundtrip(tmp_path):
    from oracle import hilcodec_oracle as O
    enc, dec = O.stream_init_cache(synth.model_kwargs("hil_speech"), 1)
    enc = [c + i for i, c in enumerate(enc)]
    p = str(tmp_path / "cache_enc.npz")
    wire.save_cache_npz(p, enc, "e_in")
    back = wire.load_cache_npz(p, "e_in")
    assert len(back) == 22 and all(torch.equal(a, b) for a, b in zip(enc, back))
    b4 = wire.load_cache_npz(p, "e_in", batch=4)
    assert b4[5].shape == (4, 128, 2) and torch.equal(b4[5][3], enc[5][0])
    with pytest.raises(KeyError):
        wire.load_cache_npz(p, "d_in")


@pytest.mark.skipif(not os.path.isdir(REF_ONNX), reason="reference checkout not present")
def test_against_reference_shipped_files():
    from oracle import hilcodec_oracle as O
    enc_shapes, dec_shapes = O.stream_cache_shapes(synth.model_kwargs("hil_speech"))
    e = wire.load_cache_npz(os.path.join(REF_ONNX, "hil_speech_cache_enc.npz"), "e_in")
    d = wire.load_cache_npz(os.path.join(REF_ONNX, "hil_speech_cache_dec.npz"), "d_in")
    assert [tuple(c.shape) for c in e] == [(1, c, l) for c, l in enc_shapes] and not any(c.any() for c in e)
    assert [tuple(c.shape) for c in d] == [(1, c, l) for c, l in dec_shapes]
    q = wire.load_indices_npy(os.path.join(REF_ONNX, "hil_speech_quantized.npy"))
    assert q.shape == (8, 1, 2296) and int(q.min()) >= 0 and int(q.max()) <= 1023
    for name, n in (("hil_speech", 8), ("hil_music", 12)):
        for i in (0, n - 1):
            cb = wire.read_onnx_codebook(os.path.join(REF_ONNX, f"{name}_deq{i}.onnx"))
            vq = wire.read_onnx_codebook(os.path.join(REF_ONNX, f"{name}_vq{i}.onnx"))
            assert cb.shape == (1024, 128) and torch.isfinite(cb).all() and torch.equal(cb, vq)


def test_trained_dequantizer_kat_oracle(golden):
    """The oracle's Dequantizer against the reference's Dequantizer on the reference's SHIPPED trained
    codebooks + shipped indices (only the referenced code vectors are stored in the fixture)."""
    from oracle import hilcodec_oracle as O
    g = golden("trained_deq")
    idx = torch.from_numpy(g["indices"].astype(np.int64))              # [8,1,F]
    p = {}
    for i in range(8):
        cb = torch.zeros(1024, 128)
        cb[idx[i, 0]] = torch.from_numpy(g["rows"][i])
        p[f"vq.{i}.embed"] = cb
    assert torch.equal(O.stream_dequantize(p, idx, 8), torch.from_numpy(g["q"]))
    assert tuple(g["quantized_shape"]) == (8, 1, 2296)
