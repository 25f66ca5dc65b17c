cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3n
timeout -k 10 300 python -u tools/golden_distance.py > gpurun_out/r3n/golden_distance.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r3n/golden_distance.txt
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "extend or prefill" --timeout 200 --timeout-method thread 2>&1 | tail -5 | cut -c1-400
