#!/usr/bin/env python
"""Sustained shader clock / power while one kernel shape runs in a loop (rocm-smi sampled from a thread).
usage: clock_probe.py resblock:<C> | gemm:<K>x<M>x<T> | dws:<K>x<M>x<T>"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hilcodec_amd import ops
dev = torch.device("cuda:0")
kind, shape = sys.argv[1].split(":")
B = 256
if kind == "resblock":
    C = int(shape); T = 12000 if C >= 128 else 24000
    x = torch.randn(B, C, T, device=dev)
    w1 = torch.randn(C, C, device=dev) / C ** 0.5; w2 = torch.randn(C, C, device=dev) / C ** 0.5
    d1 = torch.randn(C, 5, device=dev); b1 = torch.randn(C, device=dev); d2 = torch.randn(C, 5, device=dev); b2 = torch.randn(C, device=dev)
    fn = lambda: ops.resblock(x, w1, d1, b1, w2, d2, b2, 0.9, 0.5)
    flops = 4.0 * B * T * C * C
else:
    K, M, T = (int(v) for v in shape.split("x"))
    x = torch.randn(B, K, T, device=dev); wt = torch.randn(K, M, device=dev) / K ** 0.5
    y = torch.empty(B, M, T, device=dev)
    dw = torch.randn(M, 5, device=dev); db = torch.randn(M, device=dev)
    fn = (lambda: ops.pw_conv(x, wt)) if kind == "gemm" else (lambda: ops.dws_conv(x, wt, dw, db, in_scale=0.9, in_elu=True))
    flops = 2.0 * B * T * K * M
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            s = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
            samples.append(" | ".join(s))
        except Exception as e:
            samples.append(repr(e))
        time.sleep(0.5)
fn(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(20): fn()
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{sys.argv[1]}: {flops * n / dt / 1e12:.1f} TF sustained over {dt:.1f} s")
for s in samples[2:8]: print("   ", s)
