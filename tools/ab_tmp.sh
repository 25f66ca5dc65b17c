set -e
L=baby-sglang_amd/libbsgl_attn.so
cp $L /tmp/new_lib.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "kv_heads_of_64 or random or decode or golden or fused or balanced or full_size" 2>&1 | tail -3
for rep in 1 2; do
for which in old new; do
  if [ $which = old ]; then cp tools/old_lib.so $L; else cp /tmp/new_lib.so $L; fi
  echo "== $which"
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-extras --cpu-seconds 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"
  timeout -k 10 200 python tools/bench_decode_shapes.py --cases 0.5b:64:2048,1.5b:32:2048,1.5b:64:2048,7b:64:2048,1.5b:64:8192 2>&1 | grep -v amdgpu.ids
done
done
cp /tmp/new_lib.so $L
