#!/bin/bash
# Launch table: for every scenario of tools/launch_table.py — shipped configs, odd shapes, the non-default ExecOptions — the
# entry points of one pass (ops.timed_launches() records) and the KERNELS that ran (rocprofv3 --kernel-trace of the same pass), and at the end
# every kernel instantiation of the built library that NO scenario reached.
#   bash tools/launch_table.sh [tag]      -> gpurun_out/<tag>/launch_table.txt
TAG=${1:-launch_table}; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/launch_table.txt
: > $O/seen.txt
python $R/tools/launch_table.py --list > $O/scenarios.txt
i=0
while IFS= read -r S; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv -d $O/t$i -o t -- python $R/tools/launch_table.py "$S" > $O/s$i.log 2> $O/s$i.err
  f=$(find $O/t$i -name "*kernel_trace.csv" | head -1)
  python - "$f" "$O/s$i.log" "$O/seen.txt" <<'PY' >> $O/launch_table.txt
import collections, csv, re, sys
rows = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in csv.DictReader(open(sys.argv[1]))))
mark = max(i for i, r in enumerate(rows) if "hist_out_kernel" in r[1] and r[2] == 256)      # the marker between warm-up and the counted pass
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("hilc::", "").replace("void ", "")
    m = re.search(r"(\w+_kernel)(<.*>)?\(", k)
    return (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else k.split("(")[0]
cnt = collections.Counter(short(r[1]) for r in rows[mark + 1:])
print(open(sys.argv[2]).read().rstrip())
print("   kernels of the pass:")
for k, n in cnt.most_common():
    print(f"      {n:3d} x {k}")
print()
with open(sys.argv[3], "a") as f:
    for k in cnt:
        f.write(k + "\n")
    for r in rows[:mark]:
        f.write(short(r[1]) + "\n")          # (warm-up: one-off packing kernels count as reached too)
PY
  rm -rf $O/t$i
done < $O/scenarios.txt
cd $R
# every kernel of the library vs the ones some scenario reached
python - $O/seen.txt <<'PY' >> $O/launch_table.txt
import re, subprocess, sys, glob, os
seen = set(open(sys.argv[1]).read().split("\n"))
names = []
for obj in sorted(glob.glob("hilcodec_amd/lib/obj/*.o")):
    out = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True).stdout
    # host-side stubs of the kernels: __device_stub__ symbols name every __global__ instantiation of the object
    for line in out.splitlines():
        sym = line.split()[-1]
        if "__device_stub__" in sym:
            names.append((os.path.basename(obj), sym))
dem = subprocess.run(["c++filt"] + [n for _, n in names], capture_output=True, text=True).stdout.splitlines()
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("hilc::", "").replace("void ", "").replace("__device_stub__", "")
    m = re.search(r"(\w+_kernel)(<.*>)?\(", k)
    return (m.group(1) + (m.group(2) or "")).replace(", ", ",") if m else k.split("(")[0]
allk = {}
for (obj, _), d in zip(names, dem):
    allk.setdefault(short(d), obj)
miss = sorted(k for k in allk if k not in seen)
print(f"## {len(allk)} kernel instantiations in the library, {len(allk) - len(miss)} reached by the scenarios above, {len(miss)} not")
print("## (MB = the row-tile height 1..4 that pick_mb chooses per shape: other shapes reach the other heights; the rest are the entry points'")
print("##  forms for arguments no shipped model passes — no ELU, misaligned views, a stage without its end phase, the training-side RVQ kernels —")
print("##  each pinned by tests/test_gpu_ops.py / test_gpu_rvq.py, several as the bit-equality witness of the launch that replaced them):")
import collections
fam = collections.Counter(re.sub(r"^(gemm_lin_kernel|gemm_kernel)<(\d+),", r"\1<MB,", k) for k in miss)
for k, n in sorted(fam.items()):
    print(f"   {n:2d} x {k}   [{allk.get(k, '')}]")
PY
cat $O/launch_table.txt | tail -60
