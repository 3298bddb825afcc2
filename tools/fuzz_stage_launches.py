#!/usr/bin/env python
"""Random-shape check of the round-4 / round-5 launches of the OFFLINE model against the launches they replace (bit for bit): wide one-launch blocks vs
two hilc_dws_conv, chains vs block by block, encoder stages vs blocks + down-sampling layer, decoder stages vs up-sampling layer + blocks, the last
decoder stage with the closing conv vs stage + hilc_conv_post, the first encoder stage with its input phase vs hilc_spec_block_conv_pre + stage.
Shapes: random clip counts and lengths around the tile widths (32 / 64 / 128 columns), clips shorter than a tile, single clips.
   python tools/fuzz_stage_launches.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
g = torch.Generator().manual_seed(123)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)


def block(C, j):
    w1, w2 = rnd(C, C) / C ** 0.5, rnd(C, C) / C ** 0.5
    d1, b1, d2, b2 = rnd(C, 5) * 0.5, rnd(C) * 0.2, rnd(C, 5) * 0.5, rnd(C) * 0.2
    pre, post = (1.0 + j / 3.0) ** -0.5, 0.4 + 0.1 * j
    return dict(w1=w1, w2=w2, d1=d1, b1=b1, d2=d2, b2=b2, pre=pre, post=post,
                single=(ops.resblock_pack(w1), d1, b1, ops.resblock_pack(w2), d2, b2),
                chain=(ops.resblock_chain_pack(w1, False), d1, b1, ops.resblock_chain_pack(w2, False), d2, b2, pre, post))


def pick_T(mult):
    base = rng.choice([4, 8, 28, 32, 36, 60, 64, 68, 124, 128, 132, 252, 300, 600, 1160, 3000])
    t = base + 4 * rng.randrange(0, 3)
    return max(mult, (t // mult) * mult)


bad = 0
for case in range(N):
    kind = rng.choice(["wide", "chain", "enc", "dec", "post", "spec0"])
    try:
        if kind == "wide":
            C = rng.choice([256, 384, 512, 768]); T = pick_T(4); B = rng.choice([1, 2, 3, 7, 40])
            bl = block(C, 0); x = rnd(B, C, T)
            y = ops.resblock(x, *bl["single"], bl["pre"], bl["post"])
            h = ops.dws_conv(x, bl["w1"], bl["d1"], bl["b1"], in_scale=bl["pre"], in_elu=True, out_elu=True)
            ref = ops.dws_conv(h, bl["w2"], bl["d2"], bl["b2"], res=x, out_scale=bl["post"])
        elif kind == "chain":
            C = rng.choice([64, 96, 128, 192, 256, 384, 512]); n = rng.choice([2, 3]) if C in (96, 192, 384) else 2
            T = pick_T(4); B = rng.choice([1, 2, 5, 33])
            bls = [block(C, j) for j in range(n)]; x = rnd(B, C, T)
            assert ops.resblock_chain_supported(C, T, n, B, streaming=False)
            y = ops.resblock_chain(x, [b["chain"] for b in bls])
            ref = x
            for b in bls:
                ref = ops.resblock(ref, *b["single"], b["pre"], b["post"])
        elif kind == "enc":
            C, r = rng.choice([(64, 2), (128, 4), (256, 5), (512, 8)]); n = rng.choice([1, 2])
            T = pick_T(4 * r if r != 8 else 8); B = rng.choice([1, 2, 5, 33])
            if T % 4 or T % r:
                continue
            bls = [block(C, j) for j in range(n)]; x = rnd(B, C, T)
            wd, dw, db = rnd(C, 2 * C) / C ** 0.5, rnd(2 * C, 2 * r) * 0.4, rnd(2 * C) * 0.2
            down = (ops.resblock_chain_pack(wd[:, :C].contiguous(), False), ops.resblock_chain_pack(wd[:, C:].contiguous(), False), dw, db, 0.7746, r)
            res = rnd(B, 2 * C, T // r) if rng.random() < 0.5 else None
            assert ops.encoder_stage_supported(C, T, n, r, B, streaming=False)
            y = ops.encoder_stage(x, [b["chain"] for b in bls], down, res=res)
            ref = x
            for b in bls:
                ref = ops.resblock(ref, *b["single"], b["pre"], b["post"])
            ref = ops.dws_conv(ref, wd, dw, db, stride=r, in_scale=0.7746, in_elu=True)
            if res is not None:
                ref = ref + res
        elif kind == "post":
            C, r, n = 96, 2, 3
            Tin = max(2, pick_T(4) // r); B = rng.choice([1, 2, 5, 33, 300])
            if (Tin * r) % 4:
                Tin *= 2
            T = Tin * r
            bls = [block(C, j) for j in range(n)]
            tw, wu, bu = rnd(2 * C, 2 * r) * 0.3, rnd(2 * C, C) / (2 * C) ** 0.5, rnd(C) * 0.1
            up = (tw, ops.resblock_chain_pack(wu[:C].contiguous(), False), ops.resblock_chain_pack(wu[C:].contiguous(), False), bu, 0.7071, r)
            pw, pb = rnd(C, 5) * 0.2, (rnd(1) * 0.1 if rng.random() < 0.7 else None)
            tanh = rng.random() < 0.7
            xin = rnd(B, 2 * C, Tin)
            assert ops.decoder_stage_post_supported(C, T, n, r, 5)
            y = ops.decoder_stage_post(xin, up, [b["chain"] for b in bls], (pw, pb, 0.5, 0.1122, tanh))
            ref = ops.conv_post(ops.decoder_stage(xin, up, [b["chain"] for b in bls]), pw, pb, in_scale=0.5, in_elu=True, out_scale=0.1122, do_tanh=tanh)
        elif kind == "spec0":
            from hilcodec_amd import fold, synth
            C, r = 64, 2; n = rng.choice([1, 2])
            T = pick_T(4); B = rng.choice([1, 2, 5, 33, 300])
            bls = [block(C, j) for j in range(n)]
            bt = fold.stft_basis_layout(synth.stft_basis(64)).to(dev)
            wt = rnd(33, 64) / 33 ** 0.5
            dft_p, nyq, pw_p = ops.spec_block_tables(bt, wt.contiguous(), 64)
            sb = rnd(64) * 0.1 if rng.random() < 0.7 else None
            pre_w, pre_b = rnd(64, 5) * 0.5, (rnd(64) * 0.1 if rng.random() < 0.7 else None)
            wd, dw, db = rnd(C, 2 * C) / C ** 0.5, rnd(2 * C, 2 * r) * 0.4, rnd(2 * C) * 0.2
            down = (ops.resblock_chain_pack(wd[:, :C].contiguous(), False), ops.resblock_chain_pack(wd[:, C:].contiguous(), False), dw, db, 0.7746, r)
            res = rnd(B, 2 * C, T // r) if rng.random() < 0.5 else None
            wav = rnd(B, 1, T) * 0.1
            norm = rng.choice([0, 1])
            spec = (dft_p, nyq, pw_p, sb, pre_w, pre_b, 8.9, -4.0, 2.8, norm, 0.37)
            assert ops.encoder_stage0_supported(T, n, r, 64, 1, 5)
            y = ops.encoder_stage0(wav, spec, [b["chain"] for b in bls], down, res=res)
            x0 = ops.spec_block_conv_pre(wav, dft_p, nyq, pw_p, sb, pre_w, pre_b, 8.9, 64, 1, -4.0, 2.8, norm, 0.37)
            ref = ops.encoder_stage(x0, [b["chain"] for b in bls], down, res=res)
        else:
            C, r, nmax = rng.choice([(96, 2, 3), (192, 4, 3), (384, 5, 3), (768, 8, 1)]); n = rng.randint(1, nmax)
            Tin = max(1, pick_T(4) // r); B = rng.choice([1, 2, 5, 33])
            if (Tin * r) % 4:
                Tin *= 4
            bls = [block(C, j) for j in range(n)]
            tw, wu, bu = rnd(2 * C, 2 * r) * 0.3, rnd(2 * C, C) / (2 * C) ** 0.5, rnd(C) * 0.1
            taps = ops.up_conv_taps(tw, r)
            up = (tw if taps is None else taps, ops.resblock_chain_pack(wu[:C].contiguous(), False), ops.resblock_chain_pack(wu[C:].contiguous(), False), bu, 0.7071, r)
            xin = rnd(B, 2 * C, Tin); T = Tin * r
            assert ops.decoder_stage_supported(C, T, n, r, B, streaming=False)
            y = ops.decoder_stage(xin, up, [b["chain"] for b in bls])
            ref = ops.up_conv(xin, tw, wu, bu, r, in_scale=0.7071, in_elu=True)
            for b in bls:
                ref = ops.resblock(ref, *b["single"], b["pre"], b["post"])
        ok = torch.equal(y, ref)
        if not ok:
            bad += 1
        print(f"{case:4d} {kind:5s} C={C} T={T} B={B} {'ok' if ok else 'MISMATCH ' + str(float((y - ref).abs().max()))}", flush=True)
    except Exception as e:
        bad += 1
        print(f"{case:4d} {kind:5s} C={C} ERROR {type(e).__name__}: {str(e)[:200]}", flush=True)
print("mismatches / errors:", bad)
sys.exit(1 if bad else 0)
