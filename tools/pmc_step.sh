#!/bin/bash
# Instruction mix of every kernel of the offline step (PMC pass only): VALU / MFMA / SALU / LDS / VMEM instructions per launch.
#   bash tools/pmc_step.sh <tag>
TAG=${1:-pmc_step}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p -o p -- \
  python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-clock-probe --no-launch-timing --no-other-configs > $O/p.log 2>&1
f=$(find $O/p -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' > $O/instruction_mix.txt
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("hilc::", "")
    m = re.search(r"(\w+_kernel)(<[^(]*>)?\(", k)
    k = (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else k[:50]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print(f"{'kernel':70s} {'launches':>8s} {'MFMA/l':>10s} {'VALU-MFMA/l':>12s} {'per MFMA':>9s} {'SALU/l':>10s} {'LDS/l':>10s} {'VMEM_RD/l':>10s} {'VMEM_WR/l':>10s}")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0)):
    c = agg[k]; l = n[(k, "SQ_INSTS_VALU")] or 1
    mf = c.get("SQ_INSTS_MFMA", 0) / l; va = c.get("SQ_INSTS_VALU", 0) / l - mf
    print(f"{k[:70]:70s} {l:8d} {mf / 1e6:9.2f}M {va / 1e6:11.2f}M {va / mf if mf else float('nan'):9.2f} {c.get('SQ_INSTS_SALU', 0) / l / 1e6:9.2f}M "
          f"{c.get('SQ_INSTS_LDS', 0) / l / 1e6:9.2f}M {c.get('SQ_INSTS_VMEM_RD', 0) / l / 1e6:9.2f}M {c.get('SQ_INSTS_VMEM_WR', 0) / l / 1e6:9.2f}M")
PY
cat $O/instruction_mix.txt
rm -rf $O/p
