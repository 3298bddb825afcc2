#!/usr/bin/env python
"""Per-phase s_memtime stamps of hilc_resblock (debug aid): median cycles per phase over all workgroups."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the stamp hook is compiled out of the product library (it would be process-global state): build a debug copy
DBG = os.path.join(ROOT, "gpurun_out", "libhilcodec_amd_stamps.so")
EXTRA = [f for f in os.environ.get("HILC_STAMP_FLAGS", "").split() if f]
if os.environ.get("HILC_STAMP_LIB"):          # a stamped library built beforehand (__graft_entry__.compile_library(..., defines=("HILC_DEBUG_STAMPS",)))
    DBG = os.path.abspath(os.environ["HILC_STAMP_LIB"])
elif not os.path.isfile(DBG) or EXTRA:
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                    "-ffp-contract=off", "-DHILC_DEBUG_STAMPS", *EXTRA, "-o", DBG] + sorted(glob.glob(os.path.join(ROOT, "hilcodec_amd", "csrc", "*.hip"))),
                   check=True)
os.environ["HILC_LIB"] = DBG
import torch
from hilcodec_amd import ops
from hilcodec_amd._lib import lib
dev = torch.device("cuda:0")
B = 256
STREAMING = os.environ.get("HILC_STAMP_STREAM") in ("1", "2")      # the hop shapes of 1024 streams instead of the offline layers
SHAPES = [(64, 320), (96, 320), (128, 160), (192, 160)] if STREAMING else [(64, 24000), (96, 24000), (128, 12000), (192, 12000)]
if os.environ.get("HILC_STAMP_STREAM") == "2":              # ... the wide blocks of a hop (narrow-tile shapes)
    SHAPES = [(256, 40), (384, 40), (512, 8), (768, 8)]
if STREAMING:
    B = 1024
for C, T in SHAPES:
    x = torch.randn(B, C, T, device=dev)
    hist = [torch.randn(B, C, 4, device=dev), torch.randn(B, C, 4, device=dev)] if STREAMING else None
    w1 = torch.randn(C, C, device=dev) / C ** 0.5; w2 = torch.randn(C, C, device=dev) / C ** 0.5
    d1 = torch.randn(C, 5, device=dev); b1 = torch.randn(C, device=dev)
    d2 = torch.randn(C, 5, device=dev); b2 = torch.randn(C, device=dev)
    TO = (120 if STREAMING else 128) if C <= 192 else (56 if C < 512 else 32)
    nblk = (B * T + TO - 1) // TO if STREAMING else B * ((T + TO - 1) // TO)
    w1, w2 = ops.resblock_pack(w1), ops.resblock_pack(w2)
    ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5, hist=hist); torch.cuda.synchronize()
    buf = torch.zeros(nblk, 8, dtype=torch.int64, device=dev)
    lib.hilc_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5, hist=hist); torch.cuda.synchronize()
    lib.hilc_debug_set_stamp_buffer(None)
    d = (buf[:, 1:] - buf[:, :-1]).double()
    med = d.median(dim=0).values.tolist()
    tot = (buf[:, 7] - buf[:, 0]).double().median().item()
    live = buf[:, 0] > 0
    tt = (buf[live, 7] - buf[live, 0]).double()
    qs = torch.quantile(tt[:200000], torch.tensor([0.05, 0.25, 0.5, 0.75, 0.95, 0.99], dtype=torch.float64, device=dev)).tolist()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5, hist=hist); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"C={C}: {int(live.sum())} tiles stamped, mean {tt.mean().item():.0f}, quantiles 5/25/50/75/95/99 % = "
          + "/".join(f"{q:.0f}" for q in qs) + f"; kernel {ms:.3f} ms -> sum(ticks)/ms = {tt.sum().item() / ms / 1e6:.2f} G tile-ticks per second"
          f" = {tt.sum().item() / ms / 1e6 / 2.4 / 256:.2f} tiles in flight per CU if a tick is a 2.4 GHz cycle")
    names = ["P0 ELU(regs)", "G1", "P2 acc->lds", "P3 dw+ELU", "G2", "P5 acc->lds", "P6 dw+store+prefetch"]
    if os.environ.get("HILC_STAMP_NAMES") == "wave":     # resblock_wave_kernel's stamps
        names = ["P0 ELU->LDS", "G1", "reorder+halo out+barrier", "E1 dw1+ELU->LDS", "G2", "reorder+halo out+barrier", "E2 dw2->LDS->rows+x->HBM, next x"]
    mf = (C // 2) * (C // 32) * 64 if C <= 192 else (C // 2) * (C // 32) * 64 * (64 if C < 512 else 32) // 128
    print(f"C={C}: total {tot:.0f} ticks; ideal MFMA per GEMM {mf} cyc; " + ", ".join(f"{n}={v:.0f}" for n, v in zip(names, med)))
