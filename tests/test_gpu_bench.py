"""GPU: the bench.py contract on a small workload — one JSON line per invocation with the fields the driver and the judge
read (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config.workload / roofline / cpu_baseline), for the headline mode, the streaming schedules, the launch-structure
A/B switch and the multi-GPU plumbing at world size 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, launcher=()):
    out = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # exactly one JSON line
    return json.loads(lines[0])


def check_common(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == pytest.approx(157.3) and r["unit"] == "TFLOP/s"


def test_headline_line_small():
    d = run_bench("--batch", "8", "--steps", "2", "--warmup", "1", "--cpu-clips", "2", "--no-clock-probe", "--other-configs")
    check_common(d, 2, 1)
    assert d["ranks"]["process_group_initialized"] is False and d["ranks"]["backend"].startswith("none")
    # the short configs[2] / configs[3] lines ride in the same JSON line (the default invocation runs them at full size)
    oc = d["other_configs"]
    assert set(oc) == {"configs[2]", "configs[3] graph", "configs[3] graph, 2 stream groups", "configs[3] pipelined graph",
                       "configs[3] pipelined graph, 2 stream groups"}
    assert oc["configs[3] graph"]["index_checksum"] == oc["configs[3] graph, 2 stream groups"]["index_checksum"]
    assert "hil_music" in oc["configs[2]"]["workload"] and all(v["value"] > 0 and v["dtype"] == "f32" for v in oc.values())
    # the traffic figure names the build it was measured on
    if d["roofline"]["traffic"] is not None:
        assert set(d["roofline"]["traffic_build"]) == {"csrc_sha16", "git_sha", "stale"}
    cb = d["cpu_baseline"]["by_threads"]
    assert "1" in cb and all(len(v["passes_s"]) == 3 for k, v in cb.items() if k.isdigit())
    # honest CPU figure: a sweep over thread counts plus independent single-threaded processes, best reported
    assert len(cb) >= 4 and d["cpu_baseline"]["value"] >= cb["1"]["value"] and d["cpu_baseline"]["best_setting"] in cb
    assert d["cpu_baseline"]["cores"] == cb[d["cpu_baseline"]["best_setting"]]["cores"]
    assert d["dtype"] == "f32" and "EXPERIMENTAL" not in d["metric"]
    r = d["roofline"]
    assert r["achieved"] > 0 and 0 < r["frac"] < 1 and r["achieved"] / r["peak"] == pytest.approx(r["frac"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    p = d["parity_census"]
    assert p["index_mismatch_frames"] == 0 and p["genuine_mismatches"] == 0 and p["dwav_max_end_to_end"] < 1e-4


@pytest.mark.parametrize("extra", [(), ("--graph",), ("--graph", "--pipeline")])
def test_streaming_lines_small(extra):
    d = run_bench("--mode", "streaming", "--batch", "16", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
                  "--no-clock-probe", *extra)
    check_common(d, 4, 2)
    assert "streaming" in d["config"]["workload"]
    if "--pipeline" in extra:
        assert "pipelined" in d["config"]["workload"]


def test_exec_opt_switch_changes_no_index():
    """`--exec-opt` (A/B of a launch-structure flag) prints the same index checksum as the default structure"""
    base = run_bench("--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-clock-probe")
    d = run_bench("--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-clock-probe",
                  "--exec-opt", "stage_launches=0", "--exec-opt", "wide_blocks=0")
    check_common(d, 2, 1)
    assert d["index_checksum"] == base["index_checksum"] and d["dtype"] == "f32"


def test_rccl_path_executes_at_world_size_one():
    """The multi-GPU plumbing on the one GPU there is: launched the way the driver launches N > 1 (torch.distributed.run, one
    rank per GPU), with --force-dist the process group is REALLY created over RCCL (`init_process_group("nccl", device_id=...)`),
    both barriers and the counters all_gather run through it (/root/reference/train.py:51-61 is the reference's
    NCCL setup)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = run_bench("--gpus", "1", "--force-dist", "--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                  "--no-clock-probe",
                  launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", str(port)))
    check_common(d, 2, 1)
    r = d["ranks"]
    assert r["process_group_initialized"] is True and r["backend"].startswith("rccl") and r["rccl_ranks"] == 1
    assert len(r["wall_s_per_rank"]) == 1 and r["collectives_in_timed_region"] == 0
    plain = run_bench("--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-clock-probe")
    assert plain["index_checksum"] == d["index_checksum"]            # the group changes nothing about the results


def test_emulated_rank_line_and_shard_checksums():
    """BASELINE configs[4] (hil_music, 2048 clips over 8 GPUs) on one GPU: `--emulate-rank r --emulate-world 8` runs rank r's
    shard; here at a small per-GPU batch, the 8 per-rank index checksums (in-process, same sharding code) add up to the
    un-sharded job's."""
    import torch
    import bench
    from hilcodec_amd import distributed as D
    d = run_bench("--emulate-rank", "7", "--emulate-world", "8", "--batch", "2", "--samples", "2400", "--steps", "1", "--warmup", "1",
                  "--no-cpu-baseline", "--no-clock-probe")
    assert d["config"]["shard"] == [14, 16] and d["config"]["global_batch"] == 16 and "EMULATED rank 7 of 8" in d["config"]["workload"]
    assert "hil_music" in d["config"]["workload"] and d["n_gpus"] == 1
    dev = torch.device("cuda:0")
    total, world, T = 16, 8, 2400
    step, _, ctx = bench.offline_workload("hil_music", total, 0, T, dev)
    with torch.no_grad():
        idx_all, _ = step(0)
    model = ctx["model"]
    from hilcodec_amd import synth
    sums = []
    for r in range(world):
        lo, hi = D.shard_range(total, r, world)
        x = synth.synth_clips(hi - lo, T, seed=1234, first=lo).to(dev)
        with torch.no_grad():
            _, _, _, idx = model.quantizer(model.encoder(x), None, return_indices=True)
        assert torch.equal(idx, idx_all[lo:hi])                     # a shard computes exactly its clips of the global batch
        sums.append(int(idx.sum()))
    assert sum(sums) == int(idx_all.sum()) and sums[7] == d["index_checksum"]
