cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LIB=motion-planning-for-autonomous-driving-with-mpc_amd/csrc/libmpcgpu.so
cp $LIB /tmp/lib_keep.so
mkdir -p gpurun_out/ksp
for v in STOP3 STOP4 STOP1 STOP2; do
  cp tools/ubench/_exp/ab/lib$v.so $LIB
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ksp/$v -o t --output-format csv -- python tools/ab_time.py 4096 zamlf_n30_nx6 $v > gpurun_out/ksp/$v.log 2>&1
  grep "k_start" gpurun_out/ksp/$v/t_kernel_stats.csv | awk -F'","' -v v=$v '{printf "%s k_start calls %s avg %.1f us min %.1f\n", v, $2, $4/1000, $6/1000}'
done
cp /tmp/lib_keep.so $LIB
