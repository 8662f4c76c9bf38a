#!/bin/bash
# usage: tools/kernel_regs.sh build/obj/recursion_x_34.o [name-filter]   -- VGPRs / scratch / spills of every gfx950 kernel in a hipcc object
set -e
obj=$(realpath "$1"); filt=${2:-.}
tmp=$(mktemp -d /tmp/kregs.XXXXXX)
cp "$obj" "$tmp/o.o"
(cd "$tmp" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o >/dev/null 2>&1)
f=$(ls "$tmp" | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/$f" | grep -E "^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.vgpr_spill_count" | paste - - - - | grep "$filt" | awk '{print "scratch", $4, "vgprs", $6, "spills", $8, $2}'
