cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout -k 10 600 python -u tools/bench_serving.py > gpurun_out/r3m/bench_serving.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r3m/bench_serving.txt | tail -8
timeout -k 10 600 python -u tools/bench_serving.py --model 0.5b > gpurun_out/r3m/bench_serving05.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r3m/bench_serving05.txt | tail -4
timeout -k 10 600 python -u tools/bench_serving.py --requests 64 --prompt 256 --shared 128 --new 256 > gpurun_out/r3m/bench_serving_decode.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r3m/bench_serving_decode.txt | tail -4
