cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3j/gpu_tests.txt 2>&1; rc=$?
tail -15 gpurun_out/r3j/gpu_tests.txt | cut -c1-300
exit $rc
