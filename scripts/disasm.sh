#!/bin/bash
# Disassemble the gfx950 code object of one HIP object file of the product.
#   scripts/disasm.sh glx_aggregate [outdir]    -> <outdir>/glx_aggregate.s (+ .symbols: demangled kernel names)
set -e
name=${1:-glx_aggregate}
out=${2:-/tmp/isa}
root=$(cd "$(dirname "$0")/.." && pwd)
llvm=/opt/rocm/lib/llvm/bin
mkdir -p "$out"
$llvm/llvm-objcopy --dump-section .hip_fatbin="$out/$name.fat" "$root/graph-learn_amd/lib/obj/$name.o"
$llvm/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 \
  --input="$out/$name.fat" --output="$out/$name.co"
$llvm/llvm-objdump -d "$out/$name.co" | c++filt > "$out/$name.s"
grep -n '^[0-9a-f]* <' "$out/$name.s" > "$out/$name.symbols"
echo "$out/$name.s: $(wc -l < "$out/$name.s") lines, $(wc -l < "$out/$name.symbols") symbols"
