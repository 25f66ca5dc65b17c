set -e
L=baby-sglang_amd/libbsgl_attn.so
cp $L /tmp/new_lib.so
for rep in 1 2; do
for which in old new; do
  if [ $which = old ]; then cp tools/old_lib.so $L; else cp /tmp/new_lib.so $L; fi
  echo "== $which"
  timeout -k 10 120 python tools/bench_prefill.py --reps 10 2>&1 | grep -v amdgpu.ids
  timeout -k 10 120 python tools/bench_prefill.py --reps 10 --batch 32 --ext 2048 --prefix 0 2>&1 | grep -v amdgpu.ids
done
done
cp /tmp/new_lib.so $L
timeout -k 10 120 python tools/bench_prefill.py --reps 10 --profile 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "prefill or extend or kv_heads_of_64 or random" 2>&1 | tail -3
