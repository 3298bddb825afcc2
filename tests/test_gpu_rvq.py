"""GPU parity of the RVQ kernels: bit-exact indices against the reference's golden vectors and the
oracle, except where the fp64 best-vs-second gap shows a sub-rounding near-tie (counted, bounded)."""
import numpy as np
import pytest
import torch

from hilcodec_amd import synth

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def make_codebooks(seed, nq, K=1024, D=128):
    return {f"quantizer.layers.{i}.embed": torch.from_numpy(
        synth.normalish(synth.key_seed(seed, f"rvq{i}"), K * D) * np.float32(0.3 * 0.95 ** i)).view(K, D)
        for i in range(nq)}


def check_indices(O, sd, z, idx_gpu, idx_ref, eps=1e-4):
    """bit-exact, or the first differing stage of a frame must be a near-tie in fp64 (< eps)."""
    same = idx_gpu == idx_ref
    if same.all():
        return 0
    gaps = O.rvq_gaps_fp64(sd, z, idx_ref)
    n_bad = 0
    B, n, Tn = idx_ref.shape
    for b, t in {(b, t) for b, s, t in (~same).nonzero().tolist()}:
        s0 = int((~same[b, :, t]).nonzero()[0])
        assert gaps[b, s0, t] < eps, f"genuine RVQ mismatch at b={b} stage={s0} t={t}: fp64 gap {gaps[b, s0, t]:.3e}"
        n_bad += 1
    return n_bad


def test_rvq_golden(golden):
    from hilcodec_amd import fold, ops
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    g = golden("rvq")
    nq, K, D = 12, 1024, 128
    z = torch.from_numpy(synth.normalish(int(g["z_seed"]), 2 * D * 75)).view(2, D, 75)
    z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
    sd = make_codebooks(int(g["codebook_seed"]), nq)
    cb, cbt, norms = [t.to(dev) for t in fold.codebook_tables([sd[f"quantizer.layers.{i}.embed"] for i in range(nq)])]
    idx, q, loss = ops.rvq_encode(z.to(dev), cb, cbt, norms, nq, want_loss=True)
    gi = T(g["indices"]).long()
    near = check_indices(O, sd, z, idx.cpu(), gi)
    assert near <= 1
    if near == 0:
        assert torch.equal(q.cpu(), T(g["q"]))                       # same gathers, same sum order -> bit-exact
        assert abs(float(loss) - float(g["loss"])) <= 2e-6 * float(g["loss"])
    idx5, q5, loss5 = ops.rvq_encode(z.to(dev), cb, cbt, norms, 5, want_loss=True)
    assert check_indices(O, sd, z, idx5.cpu(), T(g["indices_n5"]).long()) <= 1
    assert idx5.shape == (2, 5, 75)
    # streaming layout [B,T,C] -> [n,B,T]
    ids, _, _ = ops.rvq_encode(z.transpose(1, 2).contiguous().to(dev), cb, cbt, norms, nq, channel_last=True,
                               stage_major=True, want_q=False)
    assert torch.equal(ids.permute(1, 0, 2), idx)
    # Dequantizer: bit-exact against the reference's q (sequential sum of gathers)
    dq = ops.rvq_decode(T(g["indices"]).long().permute(1, 0, 2).contiguous().to(dev), cb, nq, channel_last=True,
                        stage_major=True)
    assert torch.equal(dq.cpu().transpose(1, 2), T(g["q"]))
    dq2 = ops.rvq_decode(T(g["indices"]).long().to(dev), cb, nq, channel_last=False, stage_major=False)
    assert torch.equal(dq2.cpu(), T(g["q"]))
    for bad in (0, 13):
        with pytest.raises(AssertionError):
            ops.rvq_encode(z.to(dev), cb, cbt, norms, bad)


def test_rvq_ties_pick_lowest_index():
    """Duplicate code vectors: the first (lowest) index must win, like torch CPU min/max(dim)."""
    from hilcodec_amd import fold, ops
    dev = torch.device("cuda:0")
    K, D = 1024, 128
    e = torch.from_numpy(synth.normalish(1, K * D) * np.float32(0.3)).view(K, D)
    e[700] = e[5]; e[300] = e[5]; e[1023] = e[17]
    z = torch.stack([e[5] * 1.0, e[17] * 1.0, e[300]], dim=0).t().reshape(1, D, 3).contiguous()
    cb, cbt, norms = [t.to(dev) for t in fold.codebook_tables([e])]
    idx, q, _ = ops.rvq_encode(z.to(dev), cb, cbt, norms, 1)
    assert idx.cpu().flatten().tolist() == [5, 17, 5]


@pytest.mark.parametrize("B,Tn,nq", [(1, 1, 8), (3, 75, 8), (2, 333, 12), (16, 75, 2)])
def test_rvq_vs_oracle(B, Tn, nq):
    from hilcodec_amd import fold, ops
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    D = 128
    z = torch.from_numpy(synth.normalish(B * 1000 + Tn, B * D * Tn)).view(B, D, Tn)
    z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
    sd = make_codebooks(31 + nq, nq)
    cb, cbt, norms = [t.to(dev) for t in fold.codebook_tables([sd[f"quantizer.layers.{i}.embed"] for i in range(nq)])]
    q_o, _, loss_o, idx_o = O.rvq_forward(sd, z, None, nq)
    idx, q, loss = ops.rvq_encode(z.to(dev), cb, cbt, norms, nq, want_loss=True)
    near = check_indices(O, sd, z, idx.cpu(), idx_o)
    assert near <= max(1, B * Tn // 2000)
    if near == 0:
        assert torch.equal(q.cpu(), q_o)
        assert abs(float(loss) - float(loss_o)) <= 2e-6 * float(loss_o)


@pytest.mark.parametrize("B,Tn,nq,mixed", [(256, 75, 8, False), (120, 75, 12, True), (9, 1001, 8, False), (256, 75, 3, True)])
def test_rvq_large_batches_on_the_matrix_pipe_equal_small_ones(B, Tn, nq, mixed):
    """From 8 192 frames on hilc_rvq_encode scores on the matrix pipe (32 frames per workgroup, v_mfma_f32_32x32x2_f32 = the VALU form's
    fmaf chain over the channels in the same order): indices, quantised sum and loss equal, bit for bit, the same clips encoded in
    chunks small enough for the VALU form — both layouts, per-clip stage counts, frame counts that are no multiple of 32 — and
    the oracle on a sample of the clips."""
    from hilcodec_amd import fold, ops
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    D = 128
    assert B * Tn >= 8192
    sd = make_codebooks(23, nq)
    z = torch.from_numpy(synth.normalish(4242, B * D * Tn)).view(B, D, Tn)
    z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
    cb, cbt, norms = [t.to(dev) for t in fold.codebook_tables([sd[f"quantizer.layers.{i}.embed"] for i in range(nq)])]
    ns = [1 + (7 * b) % nq for b in range(B)] if mixed else nq
    zd = z.to(dev)
    idx, q, loss = ops.rvq_encode(zd, cb, cbt, norms, ns, want_loss=True)
    chunk = max(1, 4096 // Tn)
    parts = [ops.rvq_encode(zd[b:b + chunk].contiguous(), cb, cbt, norms, ns[b:b + chunk] if mixed else nq, want_loss=True)
             for b in range(0, B, chunk)]
    assert torch.equal(idx, torch.cat([p[0] for p in parts])) and torch.equal(q, torch.cat([p[1] for p in parts]))
    tot = sum(float(p[2]) * min(chunk, B - i * chunk) for i, p in enumerate(parts)) / B
    assert abs(float(loss) - tot) <= 2e-6 * tot
    # streaming layout [B,T,C] -> [n,B,T]
    ids, qs, _ = ops.rvq_encode(z.transpose(1, 2).contiguous().to(dev), cb, cbt, norms, ns, channel_last=True, stage_major=True)
    assert torch.equal(ids.permute(1, 0, 2), idx) and torch.equal(qs.transpose(1, 2), q)
    for b in (0, B // 2, B - 1):
        n = ns[b] if mixed else nq
        q_o, _, _, idx_o = O.rvq_forward(sd, z[b:b + 1], n, nq)
        assert check_indices(O, sd, z[b:b + 1], idx[b:b + 1, :n].cpu(), idx_o) == 0
        assert torch.equal(q[b:b + 1].cpu(), q_o)


def test_rvq_valu_switch_keeps_large_batches_off_the_matrix_pipe_with_equal_indices():
    """`flags = HILC_RVQ_VALU_ONLY` (ABI 15; `ops.rvq_encode(valu_only=True)`, the quantiser modules' `rvq_valu_only` attribute) keeps
    batches of 8 192 frames and more on the VALU form with 16 frames per workgroup — the caller's switch should an fp32 MFMA ever
    stop being a sequential fmaf chain.  Same process, same inputs: indices, q and loss equal the matrix-pipe form's bit for bit
    (uniform and per-clip stage counts), and the module attribute reaches the launch."""
    from hilcodec_amd import fold, ops
    from hilcodec_amd.models.hilcodec.vector_quantize import ResidualVQ
    dev = torch.device("cuda:0")
    nq, D, B, Tn = 8, 128, 120, 75
    sd = make_codebooks(23, nq)
    z = torch.from_numpy(synth.normalish(4242, B * D * Tn)).view(B, D, Tn)
    z = (torch.nn.functional.normalize(z, dim=1) * D ** 0.5).to(dev)
    cb, cbt, norms = [t.to(dev) for t in fold.codebook_tables([sd[f"quantizer.layers.{i}.embed"] for i in range(nq)])]
    ns = [1 + (5 * b) % nq for b in range(B)]
    for n in (nq, ns):
        a = ops.rvq_encode(z, cb, cbt, norms, n, want_loss=True)
        b = ops.rvq_encode(z, cb, cbt, norms, n, want_loss=True, valu_only=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and float(a[2]).hex() == float(b[2]).hex()
    # an unknown flag bit is refused
    import ctypes
    from hilcodec_amd import _lib
    one = ctypes.c_void_p(z.data_ptr())
    assert _lib.lib.hilc_rvq_encode(one, one, one, one, one, None, None, 1, 128, 4, 1024, 8, 8, 0, 0, 4, None) == -4
    rvq = ResidualVQ(nq, dim=D, codebook_size=1024).to(dev).eval()
    for i, l in enumerate(rvq.layers):
        l.embed.copy_(sd[f"quantizer.layers.{i}.embed"])
        l.initted = True
    q0, _, l0, i0 = rvq(z, None, return_indices=True)
    rvq.rvq_valu_only = True
    q1, _, l1, i1 = rvq(z, None, return_indices=True)
    assert torch.equal(i0, i1) and torch.equal(q0, q1) and float(l0).hex() == float(l1).hex()
    assert torch.equal(i0, ops.rvq_encode(z, cb, cbt, norms, nq)[0])


def test_rvq_modules_reference_api():
    from hilcodec_amd.models.hilcodec.vector_quantize import ResidualVQ
    from hilcodec_amd.modules.vector_quantize import ResidualVQ as LegacyRVQ
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    nq, D = 4, 128
    sd = make_codebooks(5, nq)
    z = torch.from_numpy(synth.normalish(8, 2 * D * 50)).view(2, D, 50)
    new = ResidualVQ(num_quantizers=nq, dropout=True, dropout_index=[2, 4], dim=D, codebook_size=1024,
                     kmeans_init=True).eval()
    with pytest.raises(RuntimeError):
        new(z.to(dev))                                  # un-initialised k-means codebooks: loud, not silent
    new.load_state_dict({f"layers.{i}.embed": sd[f"quantizer.layers.{i}.embed"] for i in range(nq)}, strict=False)
    for l in new.layers:
        l.initted = True
    q, nr, loss, idx = new(z.to(dev), 3, return_indices=True)
    q_o, nr_o, loss_o, idx_o = O.rvq_forward(sd, z, 3, nq)
    assert torch.equal(idx.cpu(), idx_o) and torch.equal(q.cpu(), q_o)
    assert nr.dtype == np.int64 and (nr == nr_o).all() and loss.dim() == 0
    with pytest.raises(AssertionError):
        new(z.to(dev), 5)
    old = LegacyRVQ(num_quantizers=nq, dim=D, codebook_size=1024).eval()
    for i, l in enumerate(old.layers):
        l._codebook.embed.copy_(sd[f"quantizer.layers.{i}.embed"])
    q2, nr2, loss2 = old(z.to(dev))
    q_o2, _, loss_o2, _ = O.rvq_forward(sd, z, None, nq, variant="legacy")
    assert torch.equal(q2.cpu(), q_o2)


def test_trained_dequantizer_kat(golden):
    """Dequantizer kernel on the reference's shipped trained codebooks + indices (int16 wire dtype)."""
    from hilcodec_amd.models.hilcodec.streaming import Dequantizer
    dev = torch.device("cuda:0")
    g = golden("trained_deq")
    idx = torch.from_numpy(g["indices"])                                # int16 [8,1,F]
    deq = Dequantizer(num_quantizers=8, dim=128, codebook_size=1024).eval()
    for i in range(8):
        cb = torch.zeros(1024, 128)
        cb[idx[i, 0].long()] = torch.from_numpy(g["rows"][i])
        deq.layers[i].embed.copy_(cb)
    q = deq(idx.to(dev), 8)
    assert torch.equal(q.cpu(), torch.from_numpy(g["q"]))
    q4 = deq(idx.to(dev), 4)
    ref4 = sum(torch.from_numpy(g["rows"][i]) for i in range(4)).unsqueeze(0)
    assert torch.allclose(q4.cpu(), ref4, atol=1e-6)


def test_rvq_mixed_n_per_clip():
    """one batch, a different bitrate per clip.  Each clip must equal the reference's (oracle's)
    result for a uniform call with that clip's n; unused index rows hold -1; the dequantiser honours the same n."""
    from hilcodec_amd.models.hilcodec.vector_quantize import ResidualVQ
    from hilcodec_amd.models.hilcodec.streaming import ResidualVQ as SRVQ, Dequantizer
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    nq, D, B, Tn = 8, 128, 7, 75           # 75 frames per clip: 16-frame workgroups straddle clips
    ns = [2, 8, 4, 1, 8, 2, 5]
    sd = make_codebooks(17, nq)
    z = torch.from_numpy(synth.normalish(99, B * D * Tn)).view(B, D, Tn)
    z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
    rvq = ResidualVQ(num_quantizers=nq, dim=D, codebook_size=1024).eval()
    rvq.load_state_dict({f"layers.{i}.embed": sd[f"quantizer.layers.{i}.embed"] for i in range(nq)}, strict=False)
    q, _, loss, idx = rvq(z.to(dev), ns, return_indices=True)
    assert idx.shape == (B, 8, Tn)
    err = 0.0
    for b, n in enumerate(ns):
        q_o, _, loss_o, idx_o = O.rvq_forward(sd, z[b:b + 1], n, nq)
        assert torch.equal(idx[b:b + 1, :n].cpu(), idx_o), f"clip {b} n={n}"
        assert (idx[b, n:] == -1).all()
        assert torch.equal(q[b:b + 1].cpu(), q_o)
        err += float(loss_o) * D * Tn
    assert abs(float(loss) - err / (B * D * Tn)) <= 2e-6 * float(loss)
    # streaming layout [n,B,T] + dequantiser
    zs = z.transpose(1, 2).contiguous().to(dev)
    srvq, deq = SRVQ(num_quantizers=nq, dim=D, codebook_size=1024).eval(), Dequantizer(num_quantizers=nq, dim=D, codebook_size=1024).eval()
    for m in (srvq, deq):
        m.load_state_dict({f"layers.{i}.embed": sd[f"quantizer.layers.{i}.embed"] for i in range(nq)}, strict=False)
    sidx = srvq(zs, torch.tensor(ns))
    assert torch.equal(sidx.permute(1, 0, 2), idx)
    qd = deq(sidx, ns)
    assert torch.equal(qd.transpose(1, 2), q)
    with pytest.raises(AssertionError):
        rvq(z.to(dev), [2, 9, 4, 1, 8, 2, 5])
    with pytest.raises(RuntimeError):
        rvq(z.to(dev), [2, 4])


def _train_module(seed, nq, decay, threshold, init, dev):
    from hilcodec_amd.models.hilcodec.vector_quantize import ResidualVQ
    rvq = ResidualVQ(num_quantizers=nq, dropout=False, channel_last=False, dim=128, codebook_size=1024,
                     kmeans_init=False, decay=decay, ema_num_threshold=threshold, ema_num_initial=init).to(dev).train()
    for i in range(nq):
        e = torch.from_numpy(synth.normalish(synth.key_seed(seed, f"rvq{i}"), 1024 * 128) * np.float32(0.3 * 0.95 ** i)).view(1024, 128)
        rvq.layers[i].embed.copy_(e)
        rvq.layers[i].ema_embed.copy_(e * init)
    return rvq


def test_rvq_training_branch_golden(golden):
    """two training steps (code search, EMA cluster statistics, codebook update) against the
    reference's ResidualVQ in train mode (tests/golden/rvq_train.npz).  Indices bit-exact; the statistics are a
    sum over frames whose order differs from the CPU GEMM's, so tables agree to fp32 rounding."""
    g = golden("rvq_train")
    dev = torch.device("cuda:0")
    nq, D, B, Tn = 4, 128, 4, 75
    rvq = _train_module(int(g["codebook_seed"]), nq, float(g["decay"]), 0.0, float(g["ema_num_initial"]), dev)
    for step in range(2):
        z = torch.from_numpy(synth.normalish(int(g[f"z_seed{step}"]), B * D * Tn)).view(B, D, Tn)
        z = (torch.nn.functional.normalize(z, dim=1) * D ** 0.5).to(dev).requires_grad_(True)
        q, nr, loss, idx = rvq(z, None, return_indices=True)
        assert torch.equal(idx.cpu(), T(g[f"indices{step}"]).long()), f"step {step}"
        assert abs(float(loss.detach()) - float(g[f"loss{step}"])) <= 2e-6 * float(loss.detach())
        assert (q[:, :, ::5].detach().cpu() - T(g[f"q_probe{step}"])).abs().max() <= 1e-6
        assert nr.dtype == np.int64 and not nr.any()
        loss.backward()                                     # commitment loss reaches the encoder output
        assert z.grad is not None and torch.isfinite(z.grad).all() and float(z.grad.abs().sum()) > 0
    for i in range(nq):
        assert torch.equal(rvq.layers[i].ema_num.cpu(), T(g["ema_num"][i]))     # counts are exact integers
        assert (rvq.layers[i].embed[::16].cpu() - T(g["embed_rows"][i])).abs().max() <= 2e-6
        assert (rvq.layers[i].ema_embed[::16].cpu() - T(g["ema_embed_rows"][i])).abs().max() <= 2e-6
        assert abs(float(rvq.layers[i].embed.double().sum()) - float(g["embed_sum"][i])) < 1e-3


def test_rvq_training_stats_and_expiry_vs_oracle():
    """hilc_rvq_ema_stats / hilc_rvq_ema_update against the oracle's training step on a larger batch, n < Nq,
    then the dead-code expiry: the oracle says WHICH codes expire; the replacements must be batch residuals."""
    from hilcodec_amd import fold, ops
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    nq, D, B, Tn, decay, init, thr = 3, 128, 16, 75, 0.9, 0.5, 0.46
    rvq = _train_module(123, nq, decay, thr, init, dev)
    st = {}
    for i in range(nq):
        st[f"layers.{i}.embed"] = rvq.layers[i].embed.cpu().clone()
        st[f"layers.{i}.ema_embed"] = rvq.layers[i].ema_embed.cpu().clone()
        st[f"layers.{i}.ema_num"] = rvq.layers[i].ema_num.cpu().clone()
    z = torch.from_numpy(synth.normalish(31, B * D * Tn)).view(B, D, Tn)
    z = torch.nn.functional.normalize(z, dim=1) * D ** 0.5
    # raw statistics first
    cb, cbt, norms = [t.to(dev) for t in fold.codebook_tables([st[f"layers.{i}.embed"] for i in range(nq)])]
    idx, _, _ = ops.rvq_encode(z.to(dev), cb, cbt, norms, 2)
    bucket = ops.rvq_ema_stats(z.to(dev), cb, idx, 2).cpu()
    res = z.transpose(1, 2).reshape(-1, D)
    for s in range(2):
        ind = idx[:, s].reshape(-1).cpu()
        onehot = torch.nn.functional.one_hot(ind, 1024).float()
        assert torch.equal(bucket[s, :1024], onehot.sum(0))
        assert (bucket[s, 1024:].view(1024, D) - onehot.t() @ res).abs().max() <= 2e-5
        res = res - st[f"layers.{s}.embed"][ind]
    # three module steps with expiry on (decay 0.9: unused codes fall to 0.5 * 0.9^k < 0.46 at the 1st step)
    for step in range(2):
        before = [rvq.layers[i].embed.cpu().clone() for i in range(nq)]
        q, nr, loss, idx = rvq(z.to(dev), 2, return_indices=True)
        qo, losso, idxo, expired = O.rvq_train_step(st, z, 2, nq, decay, ema_num_threshold=thr)
        assert torch.equal(idx.cpu(), idxo)
        assert nr[2] == 0 and [int(v) for v in nr[:2]] == [int(expired[i].sum()) for i in range(2)]
        assert torch.equal(rvq.layers[2].embed.cpu(), before[2])          # stage beyond n: untouched
        resid = z.transpose(1, 2).reshape(-1, D)
        for i in range(2):
            keep = ~expired[i]
            assert (rvq.layers[i].embed.cpu()[keep] - st[f"layers.{i}.embed"][keep]).abs().max() <= 2e-6
            assert torch.equal(rvq.layers[i].ema_num.cpu(), st[f"layers.{i}.ema_num"])   # reference never resets ema_num
            new = rvq.layers[i].embed.cpu()[expired[i]]
            if len(new):
                d = torch.cdist(new.double(), resid.double()).min(dim=1).values
                assert d.max() < 1e-4, "replacement vectors must be residuals of this batch"
                assert torch.allclose(rvq.layers[i].ema_embed.cpu()[expired[i]], new * init)
            # carry the module's (random) replacements into the oracle state for the next step
            st[f"layers.{i}.embed"] = rvq.layers[i].embed.cpu().clone()
            st[f"layers.{i}.ema_embed"] = rvq.layers[i].ema_embed.cpu().clone()
            resid = resid - before[i][idx[:, i].reshape(-1).cpu()]
    rvq.eval()
    q_eval, _, _, idx_eval = rvq(z.to(dev), None, return_indices=True)      # updated tables are what eval now uses
    sd = {f"quantizer.layers.{i}.embed": rvq.layers[i].embed.cpu() for i in range(nq)}
    _, _, _, idx_o = O.rvq_forward(sd, z, None, nq)
    assert check_indices(O, sd, z, idx_eval.cpu(), idx_o) <= 1


def test_legacy_rvq_contracts_and_training_vs_oracle():
    """Row a9 (`modules/vector_quantize.py`): VectorQuantize's 3-tuple and kwargs, channel_last honoured (not
    transposed behind the caller's back), and the train-mode EMA update against the pinned oracle restatement."""
    from hilcodec_amd.modules.vector_quantize import ResidualVQ, VectorQuantize
    from oracle import hilcodec_oracle as O
    dev = torch.device("cuda:0")
    D, K, nq, decay, init = 128, 1024, 3, 0.9, 0.5
    embeds = [torch.from_numpy(synth.normalish(820 + i, K * D) * np.float32(0.3)).view(K, D) for i in range(nq)]
    sd = {f"quantizer.layers.{i}.embed": e for i, e in enumerate(embeds)}
    z = torch.from_numpy(synth.normalish(75, 2 * D * 40)).view(2, D, 40)
    # single layer: (quantize, num_replace, commit_loss)
    vq = VectorQuantize(dim=D, codebook_size=K, commitment=0.25).eval().to(dev)
    vq._codebook.embed.copy_(embeds[0])
    q, nr, cl = vq(z.to(dev))
    qo, _, _, _ = O.rvq_forward(sd, z, 1, nq, variant="legacy")
    assert torch.equal(q.cpu(), qo) and nr == 0 and cl is None
    q, nr, cl = vq(z.to(dev), calculate_commitment_loss=True)
    assert abs(float(cl) - 0.25 * float(torch.nn.functional.mse_loss(qo, z))) < 1e-6
    # channel_last=True takes [B,T,C] and returns [B,T,C]
    vq_cl = VectorQuantize(dim=D, codebook_size=K, channel_last=True).eval().to(dev)
    vq_cl._codebook.embed.copy_(embeds[0])
    q_cl, _, _ = vq_cl(z.transpose(1, 2).contiguous().to(dev))
    assert q_cl.shape == (2, 40, D) and torch.equal(q_cl.transpose(1, 2).cpu(), qo)
    rvq_cl = ResidualVQ(num_quantizers=nq, dim=D, codebook_size=K, channel_last=True).eval().to(dev)
    for l, e in zip(rvq_cl.layers, embeds):
        l._codebook.embed.copy_(e)
    q3, nr3, loss3 = rvq_cl(z.transpose(1, 2).contiguous().to(dev), 2)
    q3o, _, loss3o, _ = O.rvq_forward(sd, z, 2, nq, variant="legacy")
    assert torch.equal(q3.transpose(1, 2).cpu(), q3o) and abs(float(loss3) - float(loss3o)) < 1e-6 * float(loss3o)
    # training: three EMA steps, Laplace-smoothed division (no threshold)
    rvq = ResidualVQ(num_quantizers=nq, dim=D, codebook_size=K, decay=decay, ema_num_initial=init).train().to(dev)
    st = {}
    for i, (l, e) in enumerate(zip(rvq.layers, embeds)):
        l._codebook.embed.copy_(e)
        l._codebook.ema_embed.copy_(e * init)
        st[f"layers.{i}.embed"], st[f"layers.{i}.ema_embed"], st[f"layers.{i}.ema_num"] = e.clone(), e * init, torch.ones(K) * init
    for step in range(3):
        zs = torch.from_numpy(synth.normalish(75 + step, 2 * D * 40)).view(2, D, 40)
        q, nr, loss = rvq(zs.to(dev), 2)
        qo, losso, _ = O.legacy_rvq_train_step(st, zs, 2, nq, decay)
        assert (q.cpu() - qo).abs().max() < 1e-6 and abs(float(loss) - float(losso)) < 1e-6 * float(losso)
        assert nr.dtype == np.int64 and not nr.any()
        for i in range(nq):
            cbk = rvq.layers[i]._codebook
            assert torch.equal(cbk.ema_num.cpu(), st[f"layers.{i}.ema_num"])
            assert (cbk.embed.cpu() - st[f"layers.{i}.embed"]).abs().max() < 1e-5
    assert torch.equal(rvq.layers[2]._codebook.embed.cpu(), embeds[2])        # n = 2: the third table is untouched
    # expiry with a threshold: codes below it are replaced by batch vectors and their EMA state is re-initialised
    rvq_t = ResidualVQ(num_quantizers=1, dim=D, codebook_size=K, decay=0.5, ema_num_threshold=0.3,
                       ema_num_initial=init).train().to(dev)
    rvq_t.layers[0]._codebook.embed.copy_(embeds[0])
    rvq_t.layers[0]._codebook.ema_embed.copy_(embeds[0] * init)
    _, nr, _ = rvq_t(z.to(dev))
    cbk = rvq_t.layers[0]._codebook
    assert nr[0] > 0 and int((cbk.ema_num == init).sum()) >= nr[0]
    flat = z.transpose(1, 2).reshape(-1, D)
    replaced = (cbk.ema_num.cpu() == init).nonzero().squeeze(1)[:8]
    for r in replaced.tolist():
        assert (flat == cbk.embed[r].cpu()).all(dim=1).any()                   # a vector of this very batch
