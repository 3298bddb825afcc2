#!/usr/bin/env python
"""Per-phase s_memtime stamps of hilc_resblock (debug aid): median cycles per phase over all workgroups."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the stamp hook is compiled out of the product library (it would be process-global state): build a debug copy
DBG = os.path.join(ROOT, "gpurun_out", "libhilcodec_amd_stamps.so")
if not os.path.isfile(DBG):
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                    "-DHILC_DEBUG_STAMPS", "-o", DBG] + sorted(glob.glob(os.path.join(ROOT, "hilcodec_amd", "csrc", "*.hip"))),
                   check=True)
os.environ["HILC_LIB"] = DBG
import torch
from hilcodec_amd import ops
from hilcodec_amd._lib import lib
dev = torch.device("cuda:0")
B = 256
for C, T in [(64, 24000), (96, 24000), (128, 12000), (192, 12000)]:
    x = torch.randn(B, C, T, device=dev)
    w1 = torch.randn(C, C, device=dev) / C ** 0.5; w2 = torch.randn(C, C, device=dev) / C ** 0.5
    d1 = torch.randn(C, 5, device=dev); b1 = torch.randn(C, device=dev)
    d2 = torch.randn(C, 5, device=dev); b2 = torch.randn(C, device=dev)
    nblk = B * ((T + 119) // 120)
    ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5); torch.cuda.synchronize()
    buf = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
    lib.hilc_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5); torch.cuda.synchronize()
    lib.hilc_debug_set_stamp_buffer(None)
    d = (buf[:, 1:] - buf[:, :-1]).double()
    med = d.median(dim=0).values.tolist()
    tot = (buf[:, 7] - buf[:, 0]).double().median().item()
    # residency check: when does each workgroup of the persistent grid start its first / finish its last tile?
    grid = 256 * {64: 3, 96: 2, 128: 2, 192: 1}[C]
    grid = min(grid, nblk)
    first = buf[:grid, 0].double()
    ntile_wg = (nblk + grid - 1) // grid
    last_idx = torch.arange(grid, device=dev) + (ntile_wg - 1) * grid
    last_idx = torch.where(last_idx < nblk, last_idx, last_idx - grid)
    last = buf[last_idx, 7].double()
    life = (last - first)
    q = torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0], device=dev, dtype=torch.float64)
    print(f"   grid {grid}: workgroup lifetime quantiles (0,10,25,50,75,90,100 %) {[int(v) for v in torch.quantile(life, q).tolist()]} ticks; "
          f"{int((life > 1.2 * life.median()).sum())} workgroups more than 20 % over the median")
    names = ["P0 load+ELU", "G1", "P2 acc->lds", "P3 dw+ELU", "G2", "P5 acc->lds", "P6 dw+store"]
    mf = (C // 2) * (C // 32) * 64
    print(f"C={C}: total {tot:.0f} ticks; ideal MFMA per GEMM {mf} cyc; " + ", ".join(f"{n}={v:.0f}" for n, v in zip(names, med)))
