#!/bin/bash
# Register / LDS / scratch use of every kernel of one HIP source (device-only compile, no GPU needed).
#   bash tools/kernel_regs.sh hilcodec_amd/csrc/resblock_chain.hip [extra hipcc flags] | grep 'resblock_kernel<384'
# columns: agpr vgpr lds_bytes scratch_bytes vgpr_spills  demangled-name
SRC=$1; shift
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only "$@" -c $SRC -o $T/dev.bundle || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/dev.bundle --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co > $T/notes.txt
python3 - $T/notes.txt <<'EOF'
import re, subprocess, sys
t = open(sys.argv[1]).read()
rows = []
for blk in t.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    rows.append((int(blk.split()[0]), int(g("vgpr_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")),
                 int(g("vgpr_spill_count")), g("name")))
names = subprocess.run(["c++filt"] + [r[5] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{r[0]:4d} {r[1]:4d} {r[2]:7d} {r[3]:5d} {r[4]:4d}  {n[:110]}")
EOF
rm -rf $T
