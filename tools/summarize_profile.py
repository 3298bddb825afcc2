#!/usr/bin/env python
"""Condense rocprofv3 output (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into the small
summaries committed under profiles/.  HBM traffic follows MI355X_MICROARCH.md §HBM: bytes =
FETCH_SIZE*1024*2 (gfx950 reports half of a wide coalesced read) + WRITE_SIZE*1024; separate passes."""
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

d, out_prefix, steps_total = sys.argv[1], sys.argv[2], int(sys.argv[3])


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("hilc::", "")
    m = re.search(r"(\w+_kernel)(<[^(]*>)?\(", name)          # kernel name + its full template argument list
    if m:
        return (m.group(1) + (m.group(2) or "")).replace(", ", ",")
    return name[:60]


stats = collections.OrderedDict()
for r in csv.DictReader(open(f"{d}/stats_kernel_stats.csv")):
    stats[short(r["Name"])] = dict(calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6,
                                   avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"]))
traffic = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, col in (("fetch", 0), ("write", 1)):
    for r in csv.DictReader(open(f"{d}/{which}_counter_collection.csv")):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        k = short(r["Kernel_Name"])
        traffic[k][col] += float(r["Counter_Value"])
        if col == 0:
            traffic[k][2] += 1
with open(out_prefix + "_kernel_stats.csv", "w") as f:
    f.write("kernel,calls,total_ms,avg_us,percent,hbm_read_GB_per_step,hbm_write_GB_per_step\n")
    for k, s in stats.items():
        fe, wr, n = traffic.get(k, [0, 0, 0])
        f.write(f"{k},{s['calls']},{s['total_ms']:.3f},{s['avg_us']:.1f},{s['pct']:.2f},"
                f"{fe * 1024 * 2 / 1e9 / steps_total:.3f},{wr * 1024 / 1e9 / steps_total:.3f}\n")
mf = [k for k in stats if (k.startswith(("gemm_kernel", "gemm_lin_kernel", "gemm_lin_wr_kernel")) and "Stft" not in k)
      or k.startswith(("resblock", "spec_block"))]
tot_ms = sum(stats[k]["total_ms"] for k in mf)
calls = sum(stats[k]["calls"] for k in mf)
rd = sum(traffic[k][0] for k in mf) * 1024 * 2
wr = sum(traffic[k][1] for k in mf) * 1024
def _build_id():
    """what was profiled: hash of the kernel sources (= hilcodec_amd._lib.source_hash) and the commit, so that bench.py can
    tell a stale traffic figure from a current one"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "hilcodec_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "include", "hilcodec_amd.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        sha = ""
    return h.hexdigest()[:16], (sha or os.environ.get("HILC_GIT_SHA", "unknown"))


csrc_sha16, git_sha = _build_id()
print(json.dumps({"csrc_sha16": csrc_sha16, "git_sha": git_sha, "mfma_family_kernels": mf, "calls": calls, "total_ms": tot_ms, "avg_launch_us": tot_ms / calls * 1e3,
                  "hbm_bytes_per_launch": (rd + wr) / calls, "hbm_read_GB_per_step": rd / 1e9 / steps_total,
                  "hbm_write_GB_per_step": wr / 1e9 / steps_total}))
