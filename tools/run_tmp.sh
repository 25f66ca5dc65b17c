set -e
mkdir -p gpurun_out/r3u
BSGL_PREFILL_BLK=128 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_caller_shapes.py -m gpu -x -q --timeout 120 --timeout-method thread -k "extend or prefill or chunk" > gpurun_out/r3u/t128.txt 2>&1
tail -3 gpurun_out/r3u/t128.txt
