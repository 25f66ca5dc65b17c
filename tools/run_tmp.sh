set -u
cd $GRAFT_REPO_ROOT
V=$1
mkdir -p gpurun_out/r4d
lib=baby-sglang_amd/libbsgl_attn.so
cp $lib /tmp/tree.so
cp variants/libbsgl_$V.so $lib
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_caller_shapes.py tests/test_gpu_bench_sizes.py -m gpu -x -q --timeout 100 --timeout-method thread -k "extend or prefill or chunk" > gpurun_out/r4d/t.txt 2>&1
rc=$?
tail -5 gpurun_out/r4d/t.txt
cp /tmp/tree.so $lib
[ $rc = 0 ] || exit $rc
shift
bash tools/run_ab.sh r4d 2 1 "" tree "$@"
