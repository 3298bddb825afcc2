"""HIP-graph replay of one streaming hop (encoder -> RVQ -> dequantiser -> decoder with all 52 caches).

A hop for 1024 streams is ~60 kernel launches of 30-400 us each: the GPU work is ~7.9 ms, the host needs another
~0.3 ms to issue it and the gaps between short kernels are visible.  The hop is shape-static, so it is captured
once into a graph whose inputs (the hop's samples, the caches) and outputs live at fixed addresses; a replay is one
launch.  The caches written by the hop are copied back onto the input caches inside the graph (one multi-tensor
copy), so consecutive replays chain exactly like the eager loop of `test_onnx.py:75-93,123-135`.

The captured kernels are the same launches the eager path issues (same C-ABI calls on the capture stream), so a
replayed hop is bit-identical to an eager hop (tests/test_gpu_streaming.py)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor


class GraphedHop:
    """model: `hilcodec_amd.models.hilcodec.streaming.HILCodec` (eval, reparameterisations removed).
    `step(x)` consumes `[B,1,hop]` samples (copied into the static input) and returns (indices `[n,B,T]`, wav `[B,1,hop]`)
    as views of static buffers that the next `step` overwrites."""

    def __init__(self, model, batch: int, hop: int, n: int, device: torch.device, warmup: int = 3):
        self.model, self.n = model, n
        self.device = device
        self.x = torch.zeros(batch, 1, hop, device=device)
        ce, cd = model.initialize_cache(self.x)
        self.cache_enc: List[Tensor] = [c.contiguous() for c in ce]
        self.cache_dec: List[Tensor] = [c.contiguous() for c in cd]
        self.idx: Optional[Tensor] = None
        self.wav: Optional[Tensor] = None
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # builds every lazily cached table (folded weights, codebooks, scheduler words)
                self._hop()
            for c in self.cache_enc + self.cache_dec:
                c.zero_()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.idx, self.wav = self._hop()

    def _hop(self) -> Tuple[Tensor, Tensor]:
        m = self.model
        z, ce = m.encoder(self.x, *self.cache_enc)
        idx = m.quantizer(z, self.n)
        q = m.dequantizer(idx, self.n)
        wav, cd = m.decoder(q, *self.cache_dec)
        torch._foreach_copy_(self.cache_enc + self.cache_dec, list(ce) + list(cd))
        return idx, wav

    def reset(self, cache_enc: Optional[Sequence[Tensor]] = None, cache_dec: Optional[Sequence[Tensor]] = None) -> None:
        """zero history, or resume from caches saved earlier (`wire.save_cache` / `e_in*`, `d_in*`)"""
        with torch.no_grad():
            for i, c in enumerate(self.cache_enc):
                c.zero_() if cache_enc is None else c.copy_(cache_enc[i])
            for i, c in enumerate(self.cache_dec):
                c.zero_() if cache_dec is None else c.copy_(cache_dec[i])

    def step(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        self.x.copy_(x)
        self.graph.replay()
        return self.idx, self.wav
