#!/bin/bash
# A/B of library variants on ONE box: tools/ab_bench.sh REPS ARGS... -- for every libX.so under tools/ubench/_exp/ab/, REPS alternating samples
# of tools/ab_time.py ARGS (fresh process each, the variant copied over the built library)
REPS=${1:-3}; shift
LIB=motion-planning-for-autonomous-driving-with-mpc_amd/csrc/libmpcgpu.so
cp $LIB /tmp/lib_keep.so
for r in $(seq $REPS); do
  for v in tools/ubench/_exp/ab/lib*.so; do
    cp $v $LIB
    timeout 200 python tools/ab_time.py "$@" $(basename $v .so) 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/lib_keep.so $LIB
