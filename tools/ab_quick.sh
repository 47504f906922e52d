#!/bin/bash
# A/B of the library variants under tools/ubench/_exp/ab/ on one box: headline, B = 256, stamps of a straggler round; REPS alternating rounds
REPS=${1:-2}
LIB=motion-planning-for-autonomous-driving-with-mpc_amd/csrc/libmpcgpu.so
cp $LIB /tmp/lib_keep.so
for r in $(seq $REPS); do
  for v in tools/ubench/_exp/ab/lib*.so; do
    cp $v $LIB; n=$(basename $v .so)
    timeout 200 python tools/ab_time.py 4096 zamlf_n30_nx6 $n 2>&1 | grep -v amdgpu.ids
    timeout 200 python tools/ab_time.py 256 zamlf_n30_nx6 $n 2>&1 | grep -v amdgpu.ids
  done
done
for v in tools/ubench/_exp/ab/lib*.so; do cp $v $LIB; echo $(basename $v .so); timeout 200 python tools/res_timing.py 256 1 -1 2>&1 | grep "k_solve_wg timing" | tail -1; done
cp /tmp/lib_keep.so $LIB
