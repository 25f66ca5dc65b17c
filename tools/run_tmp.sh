cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
# parity of the 8-wave form: the extend tests with the block size forced to 128
BSGL_PREFILL_BLK=128 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_sizes.py tests/test_gpu_large_pool.py -x -q -m gpu -k "prefill or extend or pool" --timeout 200 --timeout-method thread > gpurun_out/r3p/tests128.txt 2>&1; rc=$?
tail -5 gpurun_out/r3p/tests128.txt | cut -c1-300
[ $rc = 0 ] || exit $rc
for kn in "" "--knob BSGL_PREFILL_BLK=128"; do
  for r in 1 2; do
  echo "== $kn"; timeout -k 10 150 python -u tools/bench_prefill.py --warmup 30 --reps 40 $kn 2>&1 | grep -v amdgpu.ids | cut -c1-150
  done
done
echo "== profile 128"; timeout -k 10 150 python -u tools/bench_prefill.py --warmup 30 --reps 40 --knob BSGL_PREFILL_BLK=128 --profile 2>&1 | grep -v amdgpu.ids | cut -c1-200
for s in "--hq 12 --hkv 2 --batch 64" "--batch 16 --ext 4096 --prefix 0" "--dtype f16"; do
  for kn in "" "--knob BSGL_PREFILL_BLK=128"; do echo "== $s $kn"; timeout -k 10 150 python -u tools/bench_prefill.py --warmup 30 --reps 40 $s $kn 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
done
