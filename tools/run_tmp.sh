set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
lib=baby-sglang_amd/libbsgl_attn.so
cp $lib /tmp/tree.so
cp variants/libbsgl_sp1.so $lib
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_caller_shapes.py tests/test_gpu_bench_sizes.py -m gpu -x -q --timeout 100 --timeout-method thread -k "extend or prefill or chunk" > gpurun_out/r4c/t.txt 2>&1
rc=$?
tail -5 gpurun_out/r4c/t.txt
cp /tmp/tree.so $lib
[ $rc = 0 ] || exit $rc
bash tools/run_ab.sh r4c 2 1 "" sp1 sp2 nosplit
