#!/bin/bash
# code objects that hold ONLY lg_gemm_asm, one per ablation set: tools/lg_variants.sh <outdir> <generator.py> name=ablate ...
# (names containing 'wrong' are destructive: timing only, tools/lg_bench.py does not check them)
out=$1; gen=$2; shift 2
mkdir -p "$out"
L=/opt/rocm/lib/llvm/bin
for spec in "$@"; do
  name=${spec%%=*}; ab=${spec#*=}
  GRAPE_LG_ABLATE=$ab python3 "$gen" "$out/$name.s" > /dev/null || exit 1
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$out/$name.s" -o "$out/$name.o" || exit 1
  $L/ld.lld -shared "$out/$name.o" -o "$out/$name.co" || exit 1
  rm -f "$out/$name.o"
done
