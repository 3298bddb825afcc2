#!/usr/bin/env python
"""Per-workgroup lifetimes of the offline fused block (carry form: workgroup w walks tiles [w*N/G, (w+1)*N/G)): do the workgroups
that share a CU finish together?  Uses the -DHILC_DEBUG_STAMPS build of tools/res_phase_times.py."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, "gpurun_out", "libhilcodec_amd_stamps.so")
assert os.path.isfile(DBG), "run tools/res_phase_times.py first (it builds the stamped library)"
os.environ["HILC_LIB"] = DBG
import torch
from hilcodec_amd import ops
from hilcodec_amd._lib import lib
dev = torch.device("cuda:0")
B = 256
for C, T, G in [(64, 24000, 768), (96, 24000, 512), (128, 12000, 512), (192, 12000, 256)]:
    x = torch.randn(B, C, T, device=dev)
    w1 = ops.resblock_pack(torch.randn(C, C, device=dev) / C ** 0.5); w2 = ops.resblock_pack(torch.randn(C, C, device=dev) / C ** 0.5)
    d1 = torch.randn(C, 5, device=dev); b1 = torch.randn(C, device=dev); d2 = torch.randn(C, 5, device=dev); b2 = torch.randn(C, device=dev)
    tiles = (T + 127) // 128
    N = B * tiles
    ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5); torch.cuda.synchronize()
    buf = torch.zeros(N, 8, dtype=torch.int64, device=dev)
    lib.hilc_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5); torch.cuda.synchronize()
    lib.hilc_debug_set_stamp_buffer(None)
    s = buf.cpu().double()
    live = s[:, 0] > 0
    t0 = s[live, 0].min()
    end = torch.zeros(G, dtype=torch.float64); start = torch.zeros(G, dtype=torch.float64)
    # the kernel's run map (resblock.hip: classes = workgroups per CU, shares HILC_RES_SHARE2_0 / HILC_RES_SHARE3_*; equal runs
    # with HILC_EQUAL_RUNS=1 for a library built with -DHILC_RES_SHARE2_0=0.5 -DHILC_RES_SHARE3_0=0.3333 -DHILC_RES_SHARE3_1=0.3333)
    cls = G // 256
    share = {1: [1.0], 2: [0.64, 0.36], 3: [0.44, 0.31, 0.25]}[cls]
    if os.environ.get("HILC_EQUAL_RUNS") == "1":
        share = [1.0 / cls] * cls
    cum = [0]
    for v in share:
        cum.append(int(sum(share[:len(cum)]) * 65536.0 + 0.5))
    cum[-1] = 65536
    P = G // cls

    def run_of(w):
        if cls == 1:
            return w * N // G, (w + 1) * N // G
        u, c = w % P, w // P
        s0 = u * N // P
        ln = (u + 1) * N // P - s0
        return s0 + ((ln * cum[c]) >> 16), s0 + ((ln * cum[c + 1]) >> 16)
    for w in range(G):
        a, b = run_of(w)
        blk = s[a:b]
        ok = blk[:, 7] > 0
        end[w] = (blk[ok, 7].max() - t0) if ok.any() else float("nan")
        start[w] = (blk[ok, 0].min() - t0) if ok.any() else float("nan")
    dur = end - start                       # s_memtime is per XCD: only differences inside one workgroup mean anything
    total = torch.nan_to_num(dur, nan=0.0).max()
    def q(v):
        v = v[~torch.isnan(v)] / total
        return "/".join(f"{z:.3f}" for z in torch.quantile(v, torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95], dtype=torch.float64)).tolist())
    print(f"C={C}: {G} workgroups, {N // G} tiles on average; longest workgroup {total / 1e3:.0f} k ticks; lifetime / longest, 5/25/50/75/95 %: all {q(dur)}; "
          + "; ".join(f"blocks [{i * 256},{min(G, (i + 1) * 256)}) {q(dur[i * 256:(i + 1) * 256])}" for i in range((G + 255) // 256)))
