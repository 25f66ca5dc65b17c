set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3y
lib=baby-sglang_amd/libbsgl_attn.so
cp $lib /tmp/tree.so
cp variants/libbsgl_ppms.so $lib
BSGL_PREFILL_BLK=128 timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_caller_shapes.py -m gpu -x -q --timeout 100 --timeout-method thread -k "extend or prefill or chunk" > gpurun_out/r3y/t128.txt 2>&1
rc=$?
tail -15 gpurun_out/r3y/t128.txt
cp /tmp/tree.so $lib
[ $rc = 0 ] || exit $rc
bash tools/run_ab.sh r3y 1 1 "--knob BSGL_PREFILL_BLK=128" ppms ppms_c20 pp
