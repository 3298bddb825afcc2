"""HIP-graph replay of one streaming hop (encoder -> RVQ -> dequantiser -> decoder with all 52 caches) over a
persistent, ping-pong state block in HBM.

A hop for 1024 streams is ~60 kernel launches of 30-400 us each; the hop is shape-static, so it is captured once into
a graph whose inputs (the hop's samples, the caches) and outputs live at fixed addresses, and a replay is one launch.

State: every cache exists twice, as views into two contiguous HBM blocks A and B (22 + 30 caches per stream, 313 MB
per block at 1024 streams).  The reference returns the new caches as fresh tensors (`streaming.py:482-517,619-648`);
here the kernels of an even hop read A and write B, those of an odd hop read B and write A (`cache_out=` of the
streaming Encoder / Decoder), so a hop moves no cache bytes beyond what its kernels read and write — no allocation, no
copy-back.  Two graphs are captured (A->B and B->A) and replayed alternately.

The captured kernels are the same launches the eager path issues (same custom ops on the capture stream), so a
replayed hop is bit-identical to an eager hop (tests/test_gpu_streaming.py).  The graph is not a pure chain: the
log-magnitude spectra of the un-fused SpecBlocks (n_fft 256 / 512 / 1024 at one hop: small launches that depend on the
waveform only) sit on a second branch beside the first encoder stages (`engine._early_spectra`)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import engine, ops


def _spectra_on(encoder, stream: Optional[torch.cuda.Stream]):
    """while a hop is warmed up / captured on this thread: the STFT front halves of the un-fused SpecBlocks go to `stream`
    (engine._early_spectra), a branch of the graph beside the first encoder stages.  Context-local (engine._SIDE_STREAM),
    nothing is written into the model."""
    return engine.spectra_side_stream(stream)


class StateBlock:
    """The 22 + 30 caches of `batch` streams as views into ONE contiguous fp32 buffer (16-B aligned slices)."""

    def __init__(self, model, batch: int, device: torch.device):
        probe = torch.zeros(batch, 1, 1, device=device)
        ce, cd = model.initialize_cache(probe)
        shapes = [tuple(c.shape) for c in list(ce) + list(cd)]
        offs, total = [], 0
        for s in shapes:
            offs.append(total)
            n = 1
            for d in s:
                n *= d
            total += (n + 3) // 4 * 4
        self.buffer = torch.zeros(total, device=device, dtype=torch.float32)
        views = []
        for s, o in zip(shapes, offs):
            n = 1
            for d in s:
                n *= d
            views.append(self.buffer[o:o + n].view(s))
        self.enc: List[Tensor] = views[:len(ce)]
        self.dec: List[Tensor] = views[len(ce):]

    def zero_(self) -> None:
        self.buffer.zero_()

    def load_(self, cache_enc: Optional[Sequence[Tensor]], cache_dec: Optional[Sequence[Tensor]]) -> None:
        for dst, src in ((self.enc, cache_enc), (self.dec, cache_dec)):
            for i, c in enumerate(dst):
                c.zero_() if src is None else c.copy_(src[i])

    @property
    def nbytes(self) -> int:
        return self.buffer.numel() * 4


class GraphedHop:
    """model: `hilcodec_amd.models.hilcodec.streaming.HILCodec` (eval, reparameterisations removed).
    `step(x)` consumes `[B,1,hop]` samples (copied into the static input) and returns (indices `[n,B,T]`, wav `[B,1,hop]`)
    as views of static buffers that a later `step` overwrites (each parity has its own pair).

    `groups` > 1: the streams are split into that many contiguous groups, each with its own state blocks, and the graph
    runs the groups' chains (encoder -> RVQ -> dequantiser -> decoder) side by side on separate HIP streams.  Streams are
    independent, so this is the same arithmetic on the same data — outputs bit-identical to `groups=1`, NO added latency
    (unlike PipelinedHop) — but every launch of a hop covers the chip only 1.3-4 times at 1024 streams, and two or more
    independent chains fill each other's partly-filled last rounds."""

    def __init__(self, model, batch: int, hop: int, n: int, device: torch.device, warmup: int = 2, groups: int = 1):
        self.model, self.n = model, n
        self.device = device
        groups = max(1, min(int(groups), batch))
        self.bounds = [(batch * g // groups, batch * (g + 1) // groups) for g in range(groups)]
        self.x = torch.zeros(batch, 1, hop, device=device)
        # per group: a ping-pong pair of state blocks (a cache tensor is [streams, C, pad]: a group's slice must be contiguous)
        self.gstate = [(StateBlock(model, hi - lo, device), StateBlock(model, hi - lo, device)) for lo, hi in self.bounds]
        self.parity = 0                       # the block holding the CURRENT caches (input of the next hop)
        # the STFT side branch (engine._early_spectra) only for a single chain: with several chains the launches of the other
        # groups already fill the idle CUs, and a fork of a forked stream inside one capture crashes hipStreamEndCapture (ROCm 7.2)
        self.spec_side = [torch.cuda.Stream(device) if groups == 1 else None for _ in self.bounds]
        self.chain = [None] + [torch.cuda.Stream(device) for _ in self.bounds[1:]]      # group 0 runs on the capture stream
        self.sched = [ops.SchedWorkspace(device) for _ in self.bounds]    # ticket words: one workspace per concurrent chain
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):          # builds every lazily cached table (folded weights, codebooks)
                self._hop(0)
                self._hop(1)
            self._zero()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graphs, self.outs = [], []
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                out = self._hop(p)
            self.graphs.append(g)
            self.outs.append(out)
        self._zero()

    @property
    def state(self):
        """(block A, block B) of a single-group schedule (the form tests and callers of earlier rounds use)"""
        if len(self.gstate) != 1:
            raise RuntimeError("GraphedHop.state: the streams are split into groups — use .gstate[g]")
        return self.gstate[0]

    def _zero(self) -> None:
        for a, b in self.gstate:
            a.zero_()
            b.zero_()

    def _chain(self, g: int, p: int) -> Tuple[Tensor, Tensor]:
        m = self.model
        lo, hi = self.bounds[g]
        src, dst = self.gstate[g][p], self.gstate[g][p ^ 1]
        x = self.x[lo:hi]
        with ops.sched_workspace(self.sched[g]):
            with _spectra_on(m.encoder, self.spec_side[g]):
                z, _ = m.encoder(x, *src.enc, cache_out=dst.enc)
            idx = m.quantizer(z, self.n)
            q = m.dequantizer(idx, self.n)
            wav, _ = m.decoder(q, *src.dec, cache_out=dst.dec)
        return idx, wav

    def _hop(self, p: int) -> Tuple[Tensor, Tensor]:
        if len(self.bounds) == 1:
            return self._chain(0, p)
        main = torch.cuda.current_stream(self.device)
        outs = [None] * len(self.bounds)
        for g in range(1, len(self.bounds)):              # fork
            self.chain[g].wait_stream(main)
            with torch.cuda.stream(self.chain[g]):
                outs[g] = self._chain(g, p)
        outs[0] = self._chain(0, p)
        for g in range(1, len(self.bounds)):              # join
            main.wait_stream(self.chain[g])
        return torch.cat([o[0] for o in outs], dim=1), torch.cat([o[1] for o in outs], dim=0)

    def _current(self, which: str) -> List[Tensor]:
        per_group = [getattr(blocks[self.parity], which) for blocks in self.gstate]
        if len(per_group) == 1:
            return per_group[0]                                        # views of the state block itself
        return [torch.cat(cs, dim=0) for cs in zip(*per_group)]      # groups are contiguous stream ranges: copies, full batch

    @property
    def cache_enc(self) -> List[Tensor]:
        """the CURRENT encoder caches of all streams, in the reference's order (`streaming.py:458-470`) — what
        `wire.save_cache` / `reset(...)` take; with groups > 1 concatenated over the groups (copies)"""
        return self._current("enc")

    @property
    def cache_dec(self) -> List[Tensor]:
        return self._current("dec")

    def reset(self, cache_enc: Optional[Sequence[Tensor]] = None, cache_dec: Optional[Sequence[Tensor]] = None) -> None:
        """zero history, or resume from caches saved earlier (`wire.save_cache` / `e_in*`, `d_in*`)"""
        with torch.no_grad():
            self.parity = 0
            for (lo, hi), (a, _b) in zip(self.bounds, self.gstate):
                a.load_(None if cache_enc is None else [c[lo:hi] for c in cache_enc],
                        None if cache_dec is None else [c[lo:hi] for c in cache_dec])

    def step(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        self.x.copy_(x)
        self.graphs[self.parity].replay()
        out = self.outs[self.parity]
        self.parity ^= 1
        return out


class PipelinedHop:
    """Throughput schedule for a node that runs BOTH halves of the codec on the same streams (transcoding, evaluation,
    the benchmark): a two-stage software pipeline over hops.  One graph replay runs, side by side on two HIP streams,
    the encoder + RVQ of hop i and the dequantiser + decoder of hop i-1 — two independent chains (the only edge between
    them, the indices of hop i-1, was produced by the previous replay), so the tails of one chain's small launches are
    filled with workgroups of the other instead of idle CUs.  The arithmetic is the GraphedHop's — same kernels' products in the
    same order; the decoder is captured with `ExecOptions.decoder_stage_narrow = False` (its narrow stages as up-sampling launch +
    chain instead of one launch: another launch structure, the same bits) —
    outputs are bit-identical, the decoded audio just arrives one replay later (`step` returns the indices of the hop it
    was given and the wav of the previous one; `flush` decodes the last hop).  Cost: one hop (320 samples, 13.3 ms) of
    extra latency on the decoded output — a schedule for aggregate throughput, not for the lowest-latency single call,
    which stays `GraphedHop`.

    State blocks as in GraphedHop; the encoder and decoder halves of a block flip on opposite parities (the decoder is
    one hop behind), the indices travel through two fixed `[n,B,T]` buffers."""

    def __init__(self, model, batch: int, hop: int, n: int, device: torch.device, warmup: int = 2, groups: int = 1):
        """`groups` > 1: additionally split the streams into contiguous groups, each with its own encoder and decoder chain
        (2 * groups HIP streams inside the graph), as in GraphedHop."""
        self.model, self.n, self.device = model, n, device
        groups = max(1, min(int(groups), batch))
        self.bounds = [(batch * g // groups, batch * (g + 1) // groups) for g in range(groups)]
        self.x = torch.zeros(batch, 1, hop, device=device)
        self.gstate = [(StateBlock(model, hi - lo, device), StateBlock(model, hi - lo, device)) for lo, hi in self.bounds]
        self.parity = 0                       # encoder parity: the block holding the encoder caches of the next hop
        self.pending = False                  # a hop is encoded but not decoded yet
        # chains: encoder of group 0 on the capture stream, every other chain on its own stream (all forked from the capture
        # stream; a fork of a fork crashes hipStreamEndCapture on ROCm 7.2, so the STFT side branch exists for groups == 1 only)
        self.enc_stream = [None] + [torch.cuda.Stream(device) for _ in self.bounds[1:]]
        self.dec_stream = [torch.cuda.Stream(device) for _ in self.bounds]
        self.spec_side = torch.cuda.Stream(device) if groups == 1 else None
        # concurrent chains never share ticket words
        self.sched_enc = [ops.SchedWorkspace(device) for _ in self.bounds]
        self.sched_dec = [ops.SchedWorkspace(device) for _ in self.bounds]
        warm = torch.cuda.Stream(device)
        warm.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(warm), torch.no_grad():
            idx = torch.cat([self._encode(g, 0) for g in range(groups)], dim=1)
            self.idx = (torch.zeros_like(idx), torch.zeros_like(idx))
            for _ in range(warmup):
                for p in (0, 1):
                    self._both(p)
            self._zero()
        torch.cuda.current_stream(device).wait_stream(warm)
        torch.cuda.synchronize(device)
        self.graphs, self.wavs = [], []
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                wav = self._both(p)
            self.graphs.append(g)
            self.wavs.append(wav)
        self._zero()

    @property
    def state(self):
        if len(self.gstate) != 1:
            raise RuntimeError("PipelinedHop.state: the streams are split into groups — use .gstate[g]")
        return self.gstate[0]

    def _zero(self) -> None:
        for a, b in self.gstate:
            a.zero_()
            b.zero_()

    def _encode(self, g: int, p: int) -> Tensor:
        m = self.model
        lo, hi = self.bounds[g]
        st = self.gstate[g]
        with ops.sched_workspace(self.sched_enc[g]), _spectra_on(m.encoder, self.spec_side):
            z, _ = m.encoder(self.x[lo:hi], *st[p].enc, cache_out=st[p ^ 1].enc)
        return m.quantizer(z, self.n)

    def _decode(self, g: int, p: int) -> Tensor:
        """decode the hop whose encoder ran with parity p^1 (its indices sit in idx[p^1]); decoder parity = p^1"""
        m = self.model
        lo, hi = self.bounds[g]
        st = self.gstate[g]
        # (rounds 4-5 captured this chain with `engine.exec_overrides(decoder_stage_narrow=False)`: beside the encoder's chain the narrow decoder
        # stages were faster as up-sampling launch + chain.  Round 6, with EVERY stage of a hop one launch and the closing conv inside the last:
        # 4.64 ms with that override, 4.49 without (tools/ab_pipelined_overrides.py) — the model's own options stand.  The plain replay
        # (`GraphedHop`, no added latency) is at 4.46 ms: the two-chain schedule no longer buys anything at 1 024 streams.)
        with ops.sched_workspace(self.sched_dec[g]):
            wav, _ = m.decoder(m.dequantizer(self.idx[p ^ 1][:, lo:hi].contiguous(), self.n), *st[p ^ 1].dec,
                               cache_out=st[p].dec)
        return wav

    def _both(self, p: int) -> Tensor:
        main = torch.cuda.current_stream(self.device)
        G = len(self.bounds)
        wavs, idxs = [None] * G, [None] * G
        for g in range(G):                                # fork: every decoder chain, and the encoder chains of groups > 0
            self.dec_stream[g].wait_stream(main)
            with torch.cuda.stream(self.dec_stream[g]):
                wavs[g] = self._decode(g, p)
            if g > 0:
                self.enc_stream[g].wait_stream(main)
                with torch.cuda.stream(self.enc_stream[g]):
                    idxs[g] = self._encode(g, p)
        idxs[0] = self._encode(0, p)
        for g in range(G):                                # join
            main.wait_stream(self.dec_stream[g])
            if g > 0:
                main.wait_stream(self.enc_stream[g])
        self.idx[p].copy_(idxs[0] if G == 1 else torch.cat(idxs, dim=1))
        return wavs[0] if G == 1 else torch.cat(wavs, dim=0)

    @property
    def cache_enc(self) -> List[Tensor]:
        """CURRENT encoder caches of all streams, concatenated over the groups.  Only after `flush()`: while a hop is pending the
        decoder is one hop behind the encoder, and a (cache_enc, cache_dec) pair taken then would resume with the decoder out of
        step — refused instead of silently wrong."""
        self._no_pending("cache_enc")
        per_group = [blocks[self.parity].enc for blocks in self.gstate]
        return per_group[0] if len(per_group) == 1 else [torch.cat(cs, dim=0) for cs in zip(*per_group)]

    @property
    def cache_enc_unsynced(self) -> List[Tensor]:
        """The encoder caches as of the LAST ENCODED hop, also while a hop is pending (then the decoder's caches are one hop older: this
        list alone is a valid encoder snapshot, a (cache_enc_unsynced, cache_dec) pair is not a resumable state — `flush()` first for
        that).  For callers that checkpoint the encoder side only; rounds 2-4's `cache_enc` behaved like this."""
        per_group = [blocks[self.parity].enc for blocks in self.gstate]
        return per_group[0] if len(per_group) == 1 else [torch.cat(cs, dim=0) for cs in zip(*per_group)]

    @property
    def cache_dec(self) -> List[Tensor]:
        """CURRENT decoder caches (only after `flush()`, see `cache_enc`: then both cache lists describe the same instant)"""
        self._no_pending("cache_dec")
        per_group = [blocks[self.parity].dec for blocks in self.gstate]
        return per_group[0] if len(per_group) == 1 else [torch.cat(cs, dim=0) for cs in zip(*per_group)]

    def _no_pending(self, what: str) -> None:
        if self.pending:
            raise RuntimeError(f"PipelinedHop.{what}: a hop is encoded but not decoded yet — call flush() first (the decoder's caches "
                               "are one hop behind the encoder's until then)")

    def reset(self, cache_enc: Optional[Sequence[Tensor]] = None, cache_dec: Optional[Sequence[Tensor]] = None) -> None:
        """zero history, or resume from caches exported earlier (`cache_enc` / `cache_dec`, which can only be read after a `flush()`)"""
        with torch.no_grad():
            self.parity, self.pending = 0, False
            for (lo, hi), (a, _b) in zip(self.bounds, self.gstate):
                a.load_(None if cache_enc is None else [c[lo:hi] for c in cache_enc],
                        None if cache_dec is None else [c[lo:hi] for c in cache_dec])

    def step(self, x: Tensor) -> Tuple[Tensor, Optional[Tensor]]:
        """x `[B,1,hop]` -> (indices of THIS hop `[n,B,T]`, wav `[B,1,hop]` of the PREVIOUS hop or None on the first
        call); both are views of static buffers that the next-but-one `step` overwrites."""
        p = self.parity
        self.x.copy_(x)
        if self.pending:
            self.graphs[p].replay()
            wav = self.wavs[p]
        else:                                              # first hop: nothing to decode yet
            with torch.no_grad():
                self.idx[p].copy_(torch.cat([self._encode(g, p) for g in range(len(self.bounds))], dim=1))
            wav = None
        self.pending = True
        self.parity ^= 1
        return self.idx[p], wav

    def flush(self) -> Optional[Tensor]:
        """decode the last encoded hop (end of the streams)"""
        if not self.pending:
            return None
        with torch.no_grad():
            wav = torch.cat([self._decode(g, self.parity) for g in range(len(self.bounds))], dim=0)
        self.pending = False
        return wav
