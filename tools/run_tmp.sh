cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout -k 10 600 python -u -m pytest tests/test_gpu_large_pool.py -x -v -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3i/large.txt 2>&1; rc=$?
tail -40 gpurun_out/r3i/large.txt | cut -c1-300
exit $rc
