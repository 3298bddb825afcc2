#!/usr/bin/env python
"""s_memtime stamps of the fp32 GEMM core on one depthwise-separable layer (debug aid): per workgroup
start -> first slice staged -> K loop done -> epilogue done.  Builds its own copy of the library (-DHILC_DEBUG_STAMPS)."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, "gpurun_out", "libhilcodec_amd_stamps.so")
if not os.path.isfile(DBG):
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                    "-DHILC_DEBUG_STAMPS", "-o", DBG] + sorted(glob.glob(os.path.join(ROOT, "hilcodec_amd", "csrc", "*.hip"))), check=True)
os.environ["HILC_LIB"] = DBG
import torch
from hilcodec_amd import ops
from hilcodec_amd._lib import lib
dev = torch.device("cuda:0")
B = 256
setter = lib.hilc_debug_set_lin_stamp_buffer
setter.argtypes = [ctypes.c_void_p]
for K, M, T in [(768, 768, 600), (384, 384, 3000), (256, 256, 3000)]:
    x = torch.randn(B, K, T, device=dev); wt = torch.randn(K, M, device=dev) / K ** 0.5
    dw = torch.randn(M, 5, device=dev) * 0.4; db = torch.randn(M, device=dev) * 0.2
    fn = lambda: ops.dws_conv(x, wt, dw, db, in_scale=0.9, in_elu=True, out_elu=True)
    fn(); torch.cuda.synchronize()
    tiles = B * ((T + 123) // 124)
    nwg = (tiles + 7) // 8 * 8 * ((M // 32 + 3) // 4)
    buf = torch.zeros(nwg + 64, 8, dtype=torch.int64, device=dev)
    setter(ctypes.c_void_p(buf.data_ptr()))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    setter(None)
    ms = e0.elapsed_time(e1)
    live = buf[:, 0] > 0
    d = (buf[live, 1:4] - buf[live, 0:3]).double()
    q = lambda v: "/".join(f"{t:.0f}" for t in torch.quantile(v[:500000], torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64, device=dev)).tolist())
    tot = (buf[live, 3] - buf[live, 0]).double()
    import numpy as np
    hb = buf[live].cpu().numpy()
    hw = hb[:, 4] & 0xffffffff; xcc = (hb[:, 4] >> 32) & 0xf
    key = xcc * 100000 + ((hw >> 13) & 7) * 1000 + ((hw >> 12) & 1) * 100 + ((hw >> 8) & 15)
    occ = []
    for kk in np.unique(key):
        m = key == kk
        occ.append((hb[m, 3] - hb[m, 0]).sum() / float(hb[m, 3].max() - hb[m, 0].min()))
    print(f"   {len(occ)} CUs seen; workgroups in flight per CU (sum of lifetimes / span of that CU's stamps): "
          f"median {np.median(occ):.2f}, min {np.min(occ):.2f}, max {np.max(occ):.2f}")
    mf = (K // 2) * 4 * 64
    print(f"K{K} M{M} T{T}: {int(live.sum())} workgroups, kernel {ms:.3f} ms; ticks 10/50/90 %: prologue {q(d[:, 0])}, K loop {q(d[:, 1])} "
          f"(MFMA-only {mf}), epilogue {q(d[:, 2])}, total {q(tot)}; sum of workgroup lifetimes / kernel time = "
          f"{tot.sum().item() / (ms * 1e-3 * 2.3e9) / 256:.2f} per CU at 2.3 GHz")
