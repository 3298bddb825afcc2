#!/bin/bash
# SQ counters of every kernel of the OFFLINE STEP as it runs in the product (stage launches), PMC passes only, no trace domains.
#   bash tools/pmc_stage.sh <tag> [extra bench.py flags]
# Writes gpurun_out/<tag>/stage_sq_counters.txt: one block per kernel, counters averaged per launch, plus derived ratios
# (share of wave-cycles that issue / wait / are parked, VALU per MFMA, MFMA-busy share of the kernel's SIMD-cycles).
TAG=${1:-pmc_stage}; shift
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
: > $O/raw.csv
for CNT in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $O/p$i -o p$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock-probe \
    --no-launch-timing --no-other-configs "$@" > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cat "$f" >> $O/raw.csv
  rm -rf $O/p$i
done
python - $O/raw.csv <<'PY' > $O/stage_sq_counters.txt
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
rows = csv.reader(open(sys.argv[1])); hdr = None
for r in rows:
    if r and r[0] == "Correlation_Id" or (hdr is None and "Kernel_Name" in r):
        hdr = r; continue
    if hdr is None or len(r) != len(hdr): continue
    d = dict(zip(hdr, r))
    k = d["Kernel_Name"].replace("(anonymous namespace)::", "").replace("hilc::", "")
    m = re.search(r"(\w+_kernel)(<[^(]*>)?\(", k)
    k = (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else k[:60]
    agg[k][d["Counter_Name"]] += float(d["Counter_Value"]); n[(k, d["Counter_Name"])] += 1
def per(k, c):
    return agg[k].get(c, 0.0) / (n[(k, c)] or 1)
print("# SQ counters per launch, one warm-up + one timed offline step (bench.py --steps 1 --warmup 1), averaged over the launches of a kernel.")
print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles per SE/XCD (see guide).")
for k in sorted(agg, key=lambda k: -per(k, "SQ_WAVE_CYCLES") * (n[(k, "SQ_WAVE_CYCLES")] or 1)):
    wc = per(k, "SQ_WAVE_CYCLES")
    if wc <= 0: continue
    print(f"\n{k}   ({n[(k, 'SQ_WAVE_CYCLES')]} launches)")
    for c in sorted(agg[k]):
        print(f"   {c:30s} {per(k, c):18.0f}")
    mf = per(k, "SQ_INSTS_MFMA"); va = per(k, "SQ_INSTS_VALU") - mf
    print(f"   -> of the wave-cycles: issuing {per(k, 'SQ_ACTIVE_INST_ANY') / wc:.3f}, issue-stalled (WAIT_INST_ANY) {per(k, 'SQ_WAIT_INST_ANY') / wc:.3f}, parked (WAIT_ANY: waitcnt / barrier) {per(k, 'SQ_WAIT_ANY') / wc:.3f}")
    print(f"   -> VALU (non-MFMA) per MFMA {va / mf if mf else float('nan'):.2f}; LDS instr per MFMA {per(k, 'SQ_INSTS_LDS') / mf if mf else float('nan'):.2f}; "
          f"LDS bank-conflict share of LDS-active {per(k, 'SQ_LDS_BANK_CONFLICT') / max(per(k, 'SQ_LDS_IDX_ACTIVE'), 1):.3f}")
    tr = per(k, "SQ_INSTS_VALU_TRANS_F32")
    if tr > 0 and mf:
        print(f"   -> transcendental VALU (v_exp_f32 / v_log_f32) per MFMA {tr / mf:.3f} = {tr / max(va, 1):.3f} of the non-MFMA VALU instructions; "
              f"wave-cycles waiting on LDS (WAIT_INST_LDS) {per(k, 'SQ_WAIT_INST_LDS') / wc:.3f}; mean LDS / VMEM instructions in flight per wave "
              f"{per(k, 'SQ_INST_LEVEL_LDS') / wc:.2f} / {per(k, 'SQ_INST_LEVEL_VMEM') / wc:.2f}")
    bc = per(k, "SQ_BUSY_CYCLES")
    if bc > 0:
        print(f"   -> MFMA busy / SQ busy cycles {per(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / bc:.3f} (raw ratio of the two counters; both summed over the same units)")
PY
cat $O/stage_sq_counters.txt | head -150
