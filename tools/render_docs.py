#!/usr/bin/env python
"""DESIGN.md / README.md from tools/templates/*.in + the measurements of one `tools/profile_run.sh` directory: every @KEY@ of a template is a number
of that run (bench lines, layer table, PMC summary), so the documents quote what `profiles/<tag>_*` holds and nothing else.
   python tools/render_docs.py gpurun_out/r06_a"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = sys.argv[1]


def line(name):
    with open(os.path.join(D, name)) as f:
        return json.loads([l for l in f if l.startswith("{")][-1])


b = line("bench.json")
mus, sg, sp = line("bench_hil_music.json"), line("bench_streaming_graph.json"), line("bench_streaming_pipelined.json")
summ = json.load(open(os.path.join(D, "summary_mfma_family.json")))
frac = lambda d: d["roofline"].get("whole_path_frac") or d["value"] * 34.219e9 / 157.3e12
by = b.get("cpu_baseline", {}).get("by_threads", {})
gb = float(summ.get("hbm_read_GB_per_step", float("nan"))) + float(summ.get("hbm_write_GB_per_step", float("nan")))
rows = []
for l in open(os.path.join(D, "layer_table.txt")):
    m = re.match(r"\s*\d+\s+(\w+)\s+(.*?)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+) (TF|GB/s)\s*$", l)
    if m:
        rows.append((m.group(1), m.group(2).strip(), float(m.group(3)), float(m.group(5)), m.group(6)))
tab = ["| launch | ms | rate | fraction |", "|---|---|---|---|"]
for kind, shape, ms, rate, unit in rows:
    if ms < 0.3:
        continue
    tab.append(f"| {kind} {shape} | {ms:.2f} | {rate:.1f} {unit} | {rate / 157.3:.2f} |" if unit == "TF" else f"| {kind} {shape} | {ms:.2f} | {rate:.0f} {unit} | — |")
small = sum(r[2] for r in rows if r[2] < 0.3)
tab.append(f"| {sum(1 for r in rows if r[2] < 0.3)} launches below 0.3 ms (75-frame layers, RVQ, L2Norm, …) | {small:.2f} | | |")
K = {
    "OFF_MS": f"{b['ms_per_step']:.2f}", "OFF_XRT": f"{b['value']:,.0f}".replace(",", " "), "OFF_FRAC": f"{frac(b):.3f}", "OFF_FRAC3": f"{frac(b):.3f}"[2:],
    "MUS_MS": f"{mus['ms_per_step']:.2f}", "STR_MS": f"{sg['ms_per_step']:.2f}", "STR_XRT": f"{sg['value']:,.0f}".replace(",", " "), "STR_FRAC": f"{frac(sg):.3f}",
    "PIPE_MS": f"{sp['ms_per_step']:.2f}", "CPU_BEST": f"{b.get('cpu_baseline', {}).get('value', float('nan')):.1f}",
    "CPU_1": f"{by.get('1', {}).get('value', float('nan')):.2f}", "HBM_GB": f"{gb:.1f}", "HBM_FRAC": f"{gb * 1e9 / (b['ms_per_step'] * 1e-3) / 8e12:.3f}",
    "OFFLINE_TABLE": "\n".join(tab),
}
for name in ("DESIGN.md", "README.md"):
    t = open(os.path.join(ROOT, "tools", "templates", name + ".in")).read()
    for k, v in K.items():
        t = t.replace("@" + k + "@", v)
    left = re.findall(r"@[A-Z_0-9]+@", t)
    assert not left, left
    open(os.path.join(ROOT, name), "w").write(t)
    print(name, len(t.splitlines()), "lines")
