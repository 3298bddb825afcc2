#!/usr/bin/env python
"""Random-shape check of the stage launches of a STREAMING hop against the launches they replace, bit for bit over two hops with
random caches: chains vs block by block, encoder stages vs blocks + hilc_dws_conv_stream, decoder stages vs hilc_up_conv_stream + blocks,
(round 6) the wide stages, the first stage with first conv + SpecBlock vs hilc_spec_block_conv_pre + stage, the last stage with the closing conv
vs stage + hilc_conv_post — outputs and every cache.  Stream counts 1 ... 1100 (ragged against the runs of whole streams), hop lengths around the tile widths.
   python tools/fuzz_stream_launches.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
g = torch.Generator().manual_seed(321)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)


def block(C, j):
    w1, w2 = rnd(C, C) / C ** 0.5, rnd(C, C) / C ** 0.5
    d1, b1, d2, b2 = rnd(C, 5) * 0.5, rnd(C) * 0.2, rnd(C, 5) * 0.5, rnd(C) * 0.2
    pre, post = (1.0 + j / 3.0) ** -0.5, 0.4 + 0.1 * j
    return dict(pre=pre, post=post, single=(ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2),
                chain=(ops.resblock_chain_pack(w1), d1, b1, ops.resblock_chain_pack(w2), d2, b2, pre, post))


def blocks_ref(x, bls, caches):
    for j, b in enumerate(bls):
        x, caches[j] = ops.resblock(x, *b["single"], b["pre"], b["post"], hist=caches[j])
    return x


def same_caches(a, b):
    return all(torch.equal(p[0], q[0]) and torch.equal(p[1], q[1]) for p, q in zip(a, b))


bad = 0
for case in range(N):
    kind = rng.choice(["chain", "enc", "dec", "enc0", "post"])
    B = rng.choice([1, 2, 3, 7, 33, 100, 257, 1024, 1100])
    try:
        if kind == "chain":
            C = rng.choice([64, 96, 128, 192, 256, 512, 768]); n = rng.choice([2, 3]) if C in (96, 192, 768) else 2
            T = rng.choice([4, 8, 16, 32]) if C >= 512 else (rng.choice([4, 8, 40, 44, 80, 120]) if C >= 256 else rng.choice([4, 8, 40, 120, 160, 164, 320, 324, 640]))
            if not ops.resblock_chain_supported(C, T, n, B):
                continue
            bls = [block(C, j) for j in range(n)]
            ca = [[rnd(B, C, 4) * 0.7, rnd(B, C, 4) * 0.7] for _ in range(n)]
            cb = [[c.clone() for c in p] for p in ca]
            ok = True
            for h in range(2):
                x = rnd(B, C, T)
                y, flat = ops.resblock_chain(x, [b["chain"] for b in bls], ca)
                ca = [flat[2 * j:2 * j + 2] for j in range(n)]
                ref = blocks_ref(x, bls, cb)
                ok = ok and torch.equal(y, ref) and same_caches(ca, cb)
        elif kind == "enc":
            C, r = rng.choice([(64, 2), (128, 4), (256, 5), (512, 8)]); n = rng.choice([1, 2])
            T = rng.choice([8, 16, 32]) if C == 512 else (rng.choice([40, 80, 120, 200]) if C == 256 else rng.choice([4, 8, 40, 120, 160, 164, 320, 324, 640]))
            if T % r or not ops.encoder_stage_supported(C, T, n, r, B):
                continue
            bls = [block(C, j) for j in range(n)]
            wd, dw, db = rnd(C, 2 * C) / C ** 0.5, rnd(2 * C, 2 * r) * 0.4, rnd(2 * C) * 0.2
            down = (ops.resblock_chain_pack(wd[:, :C].contiguous()), ops.resblock_chain_pack(wd[:, C:].contiguous()), dw, db, 0.7746, r)
            ca = [[rnd(B, C, 4) * 0.7, rnd(B, C, 4) * 0.7] for _ in range(n)]
            cb = [[c.clone() for c in p] for p in ca]
            da = rnd(B, 2 * C, r) * 0.6
            db_ = da.clone()
            ok = True
            for h in range(2):
                x = rnd(B, C, T)
                res = rnd(B, 2 * C, T // r) if rng.random() < 0.5 else None
                y, flat, da = ops.encoder_stage(x, [b["chain"] for b in bls], down, hist=ca, down_hist=da, res=res)
                ca = [flat[2 * j:2 * j + 2] for j in range(n)]
                y2 = blocks_ref(x, bls, cb)
                ref, db_ = ops.dws_conv_stream(y2, wd, dw, db, db_, res=res, stride=r, in_scale=0.7746, in_elu=True)
                ok = ok and torch.equal(y, ref) and torch.equal(da, db_) and same_caches(ca, cb)
        elif kind == "enc0":
            # round 6: first conv + stage-0 SpecBlock + C = 64 stage in one launch vs hilc_spec_block_conv_pre(hist) + hilc_encoder_stage
            from hilcodec_amd import fold, synth
            C, r, n = 64, 2, rng.choice([1, 2]); T = rng.choice([128, 132, 320, 324, 640, 960]); H = rng.choice([63, 64, 1023])
            if not ops.encoder_stage0_supported(T, n, r, 64, 1, 5, B, True):
                continue
            bt = fold.stft_basis_layout(synth.stft_basis(64)).to(dev)
            wt = fold.pointwise_layout(rnd(C, 33, 1) / 33 ** 0.5).to(dev)
            dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt, 64)
            spec = (dft_p, nyq, pw_p, rnd(C) * 0.1 if rng.random() < 0.7 else None, rnd(64, 5) * 0.5, rnd(64) * 0.1 if rng.random() < 0.7 else None,
                    1 / 0.1122080159, -4.0, 2.8, True, 0.37)
            bls = [block(C, j) for j in range(n)]
            wd, dw, db = rnd(C, 2 * C) / C ** 0.5, rnd(2 * C, 2 * r) * 0.4, rnd(2 * C) * 0.2
            down = (ops.resblock_chain_pack(wd[:, :C].contiguous()), ops.resblock_chain_pack(wd[:, C:].contiguous()), dw, db, 0.7746, r)
            ca = [[rnd(B, C, 4) * 0.7, rnd(B, C, 4) * 0.7] for _ in range(n)]
            cb = [[c.clone() for c in p] for p in ca]
            da = rnd(B, 2 * C, r) * 0.6
            db_ = da.clone()
            ok = True
            for h in range(2):
                wav = rnd(B, 1, T) * 0.1
                hist = rnd(B, 1, H) * 0.1 if rng.random() < 0.8 else None
                res = rnd(B, 2 * C, T // r) if rng.random() < 0.5 else None
                y, flat, da = ops.encoder_stage0(wav, spec, [b["chain"] for b in bls], down, res=res, hist=ca, down_hist=da, wav_hist=hist)
                ca = [flat[2 * j:2 * j + 2] for j in range(n)]
                x0 = ops.spec_block_conv_pre(wav, dft_p, nyq, pw_p, spec[3], spec[4], spec[5], spec[6], 64, 1, -4.0, 2.8, True, 0.37, hist=hist)
                y2, f2, db_ = ops.encoder_stage(x0, [b["chain"] for b in bls], down, hist=cb, down_hist=db_, res=res)
                cb = [f2[2 * j:2 * j + 2] for j in range(n)]
                ok = ok and torch.equal(y, y2) and torch.equal(da, db_) and same_caches(ca, cb)
        elif kind == "post":
            # round 6: the last decoder stage + closing conv in one launch vs hilc_decoder_stage + hilc_conv_post with the conv's cache
            C, r, n = 96, 2, 3; Tin = rng.choice([2, 6, 40, 80, 160, 162, 320]); T = Tin * r
            if T % 4 or not ops.decoder_stage_post_supported(C, T, n, r, 5) or not ops.decoder_stage_supported(C, T, n, r, B):
                continue
            bls = [block(C, j) for j in range(n)]
            tw, wu, bu = rnd(2 * C, 2 * r) * 0.3, rnd(2 * C, C) / (2 * C) ** 0.5, rnd(C) * 0.1
            up = (tw, ops.resblock_chain_pack(wu[:C].contiguous()), ops.resblock_chain_pack(wu[C:].contiguous()), bu, 0.7071, r)
            post = (rnd(C, 5) * 0.2, rnd(1) * 0.1 if rng.random() < 0.7 else None, 0.5, 0.1122, rng.random() < 0.8)
            ca = [[rnd(B, C, 4) * 0.7, rnd(B, C, 4) * 0.7] for _ in range(n)]
            cb = [[c.clone() for c in p] for p in ca]
            ua = rnd(B, 2 * C, 1) * 0.6
            ub = ua.clone()
            pa = rnd(B, C, 4) * 0.6
            pb = pa.clone()
            ok = True
            for h in range(2):
                xin = rnd(B, 2 * C, Tin)
                wav, flat, ua, pa = ops.decoder_stage_post(xin, up, [b["chain"] for b in bls], post, ca, ua, pa)
                ca = [flat[2 * j:2 * j + 2] for j in range(n)]
                y, f2, ub = ops.decoder_stage(xin, up, [b["chain"] for b in bls], cb, ub)
                cb = [f2[2 * j:2 * j + 2] for j in range(n)]
                wav2, pb = ops.conv_post(y, post[0], post[1], in_scale=0.5, in_elu=True, out_scale=0.1122, do_tanh=post[4], hist=pb, want_hist=True)
                ok = ok and torch.equal(wav, wav2) and torch.equal(ua, ub) and torch.equal(pa, pb) and same_caches(ca, cb)
        else:
            C, r, nmax = rng.choice([(96, 2, 3), (192, 4, 3), (384, 5, 3), (768, 8, 3)]); n = rng.randint(1, nmax)
            Tin = rng.choice([1, 2, 4]) if C == 768 else (rng.choice([4, 8, 12, 16, 24]) if C == 384 else rng.choice([1, 2, 4, 8, 30, 40, 41, 80, 160]))
            T = Tin * r
            if T % 4 or not ops.decoder_stage_supported(C, T, n, r, B):
                continue
            bls = [block(C, j) for j in range(n)]
            tw, wu, bu = rnd(2 * C, 2 * r) * 0.3, rnd(2 * C, C) / (2 * C) ** 0.5, rnd(C) * 0.1
            taps = ops.up_conv_taps(tw, r)
            up = (tw if taps is None else taps, ops.resblock_chain_pack(wu[:C].contiguous()), ops.resblock_chain_pack(wu[C:].contiguous()), bu, 0.7071, r)
            ca = [[rnd(B, C, 4) * 0.7, rnd(B, C, 4) * 0.7] for _ in range(n)]
            cb = [[c.clone() for c in p] for p in ca]
            ua = rnd(B, 2 * C, 1) * 0.6
            ub = ua.clone()
            ok = True
            for h in range(2):
                xin = rnd(B, 2 * C, Tin)
                y, flat, ua = ops.decoder_stage(xin, up, [b["chain"] for b in bls], ca, ua)
                ca = [flat[2 * j:2 * j + 2] for j in range(n)]
                y2, ub = ops.up_conv(xin, tw, wu, bu, r, in_scale=0.7071, in_elu=True, hist=ub, want_hist=True)
                ref = blocks_ref(y2, bls, cb)
                ok = ok and torch.equal(y, ref) and torch.equal(ua, ub) and same_caches(ca, cb)
        if not ok:
            bad += 1
        print(f"{case:4d} {kind:5s} C={C} T={T} B={B} n={n} {'ok' if ok else 'MISMATCH'}", flush=True)
    except Exception as e:
        bad += 1
        print(f"{case:4d} {kind:5s} C={C} ERROR {type(e).__name__}: {str(e)[:200]}", flush=True)
print("mismatches / errors:", bad)
sys.exit(1 if bad else 0)
