# paired-store experiment, the quick probe: for every tools/ubench/_exp/ab/libPAIR<bits>.so (the library built with -DMPC_EXP_PAIR=<bits>, see
# mpc_stage_math.h) three comparisons of kernel variants against the pipeline on a fresh handle each (tools/pair_probe.py).  GPU box.
LIB=motion-planning-for-autonomous-driving-with-mpc_amd/csrc/libmpcgpu.so
cp $LIB /tmp/lib_keep.so
mkdir -p gpurun_out/pairst
for f in tools/ubench/_exp/ab/libPAIR*.so; do
  v=$(basename $f .so | sed s/lib//)
  cp $f $LIB
  python tools/pair_probe.py 2>&1 | grep -v amdgpu.ids | grep "solve 0 rows\|refs identical" | cut -c1-200 > gpurun_out/pairst/probe_$v.txt
  echo "== $v"; cat gpurun_out/pairst/probe_$v.txt
done
cp /tmp/lib_keep.so $LIB
