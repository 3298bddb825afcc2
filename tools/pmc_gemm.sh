#!/bin/bash
# SQ counters of the GEMM core on one layer shape (PMC passes only, no trace domains).  bash tools/pmc_gemm.sh <tag> [shape]
TAG=${1:-pmc}
SHAPE=${2:-768x768x600}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/p$i -o p$i -- python $R/tools/gemm_bench.py --only $SHAPE --reps 3 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm_kernel" not in k and "gemm_lin_kernel" not in k and "gemm_lin_wr_kernel" not in k: continue
    k = k.replace("(anonymous namespace)::", "").replace("hilc::", "")[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in agg[k].items(): print(f"   {c:34s} {v / n[(k, c)]:16.0f}  per launch ({n[(k, c)]} launches)")
PY
  rm -rf $O/p$i
done
