#!/bin/bash
# Build-time guard for the kernels that issue LDS-DMA from inline asm (conv64.hip, prop_dma.hip): the asm statements
# write M0 without telling the compiler (hipcc refuses "m0" as a clobber: reserved register).  That is only sound while
# nothing else in those translation units' device code touches M0 -- so disassemble and fail the build if any
# instruction other than our own `s_mov_b32 m0, sN` + the DMA that follows reads or writes it.
set -e
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# the flags the product objects are built with (csrc/Makefile passes its FLAGS; lab builds pass theirs incl. their -D switches)
M0_FLAGS=${M0_FLAGS:---offload-arch=gfx950 -O3 -std=c++17}
for src in "$@"; do
  asm=$(mktemp /tmp/m0check.XXXXXX.s)
  $HIPCC $M0_FLAGS --cuda-device-only -S "$src" -o "$asm"
  bad=$(grep -nw "m0" "$asm" | grep -v "^\S*\s*;" | grep -v "s_mov_b32 m0, s[0-9]*$" || true)
  n=$(grep -c "s_mov_b32 m0, s" "$asm" || true)
  rm -f "$asm"
  if [ -n "$bad" ]; then echo "check_m0: $src uses M0 outside the LDS-DMA statements:"; echo "$bad"; exit 1; fi
  echo "check_m0: $src ok ($n LDS-DMA sites, no other use of M0)"
done
