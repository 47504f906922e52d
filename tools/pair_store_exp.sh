# the paired-store experiment (profiles/r05_store_pairing.txt): the library built with -DMPC_EXP_PAIR=1 (tools/ubench/_exp/ab/libPAIRST.so) in place of
# the default one: result hashes of fixed batches on every path against the default build's, the whole GPU suite five times in sequence, 100 repeats
LIB=motion-planning-for-autonomous-driving-with-mpc_amd/csrc/libmpcgpu.so
cp $LIB /tmp/lib_keep.so
mkdir -p gpurun_out/pairst
timeout 300 python tools/bits_dump.py > gpurun_out/pairst/bits_default.txt 2>&1
cp tools/ubench/_exp/ab/libPAIRST.so $LIB
timeout 300 python tools/bits_dump.py > gpurun_out/pairst/bits_pair.txt 2>&1
diff gpurun_out/pairst/bits_default.txt gpurun_out/pairst/bits_pair.txt > gpurun_out/pairst/bits_diff.txt && echo "result hashes: identical to the default build" || { echo "result hashes DIFFER"; head -20 gpurun_out/pairst/bits_diff.txt; }
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1; done
timeout 800 python tools/repeat_check.py 100 2>&1 | grep -v amdgpu.ids
for i in 1 2 3; do timeout 200 python tools/ab_time.py 4096 zamlf_n30_nx6 pair; done 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_keep.so $LIB
for i in 1 2 3; do timeout 200 python tools/ab_time.py 4096 zamlf_n30_nx6 default; done 2>&1 | grep -v amdgpu.ids
