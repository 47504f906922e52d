cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ks
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/ks/head -o head --output-format csv -- python bench.py --headline-only --no-cpu-baseline --no-traffic > gpurun_out/ks/head.log 2>&1
head -6 gpurun_out/ks/head/head_kernel_stats.csv | cut -c1-60,200-330
tail -1 gpurun_out/ks/head.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
