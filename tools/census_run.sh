#!/bin/bash
# Full-size parity census on the GPU box, production ELU (exp2 form) vs the expm1f A/B build, both models.
# Writes gpurun_out/census/*.json (copy into profiles/ to keep).  The A/B library is built by
#   python -c "import __graft_entry__ as g; g.compile_library('hilcodec_amd/lib/libhilcodec_amd_expm1.so', defines=('HILC_ELU_EXPM1',))"
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/census
for m in hil_speech:64 hil_music:32; do
  name=${m%%:*}; n=${m##*:}
  python tests/census.py --model $name --clips $n > gpurun_out/census/${name}_exp2.json 2> gpurun_out/census/${name}_exp2.err
  if [ -f hilcodec_amd/lib/libhilcodec_amd_expm1.so ]; then
    python tests/census.py --model $name --clips $n --lib hilcodec_amd/lib/libhilcodec_amd_expm1.so \
      > gpurun_out/census/${name}_expm1.json 2> gpurun_out/census/${name}_expm1.err
  fi
done
tail -n +1 gpurun_out/census/*.json
