#!/bin/bash
# disassembly of ONE kernel of the built library: tools/disasm_kernel.sh 'k_solve_wg<6, 2, false>' [lib] > out.s   (build container, no GPU)
LIB=${2:-motion-planning-for-autonomous-driving-with-mpc_amd/csrc/libmpcgpu.so}
LLVM=/opt/rocm/lib/llvm/bin
D=$(mktemp -d); ln -s "$(realpath $LIB)" $D/lib.so
(cd $D && $LLVM/llvm-objdump --offloading lib.so > /dev/null)
CO=$(ls $D/*gfx950* | head -1)
SYM=$($LLVM/llvm-readelf -sW $CO | awk '$4=="FUNC"{print $8}' | while read s; do n=$(c++filt "$s" | sed -e 's/(anonymous namespace):://g' -e 's/^void //' -e 's/(.*$//'); if [ "$n" == "$1" ]; then echo $s; break; fi; done)
[ -z "$SYM" ] && { echo "kernel not found: $1" >&2; exit 1; }
$LLVM/llvm-objdump -d --disassemble-symbols=$SYM $CO
rm -rf $D
