/*
 * The C ABI from plain C: one decode step of paged attention on an MI355X without Python or
 * PyTorch -- what a host written in another language binds (include/bsgl_attn.h).
 * Three requests of 70 / 513 / 9 tokens, Qwen2.5-1.5B head shapes (12 q heads, 2 kv heads, head_dim
 * 128), bf16, scattered slots; the result is checked against a float loop over the same rounded
 * inputs (softmax(q K^T / sqrt(D)) V, kv head of q head h = h / 6: attn_backend.py:112-149).
 *
 *   gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/decode_from_c.c \
 *       -Lbaby-sglang_amd -lbsgl_attn -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,'$ORIGIN/../baby-sglang_amd' -Wl,-rpath,/opt/rocm/lib -o examples/decode_from_c
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bsgl_attn.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static uint16_t to_bf16(float f) {  /* round to nearest even */
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float from_bf16(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float rnd(uint32_t* s) {  /* uniform in [-1, 1) */
  *s = *s * 1664525u + 1013904223u;
  return (float)(*s >> 8) / 8388608.0f - 1.0f;
}

int main(void) {
  enum { B = 3, HQ = 12, HKV = 2, D = 128, G = HQ / HKV, MAX_CTX = 1024, N_SLOTS = 700 };
  const int seq[B] = {70, 513, 9};
  uint32_t seed = 12345u;
  const size_t slab_elems = (size_t)N_SLOTS * HKV * D;
  uint16_t* k_slab = malloc(slab_elems * 2), *v_slab = malloc(slab_elems * 2);
  uint16_t* q = malloc((size_t)B * HQ * D * 2), *out = malloc((size_t)B * HQ * D * 2);
  int32_t* table = calloc((size_t)B * MAX_CTX, 4);
  if (!k_slab || !v_slab || !q || !out || !table) return 2;
  for (size_t i = 0; i < slab_elems; ++i) { k_slab[i] = to_bf16(rnd(&seed)); v_slab[i] = to_bf16(rnd(&seed)); }
  for (size_t i = 0; i < (size_t)B * HQ * D; ++i) q[i] = to_bf16(2.0f * rnd(&seed));
  /* slots 1 .. 592 dealt out in a scattered order (slot 0 is the reference's never-allocated pad row) */
  int next = 0;
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < seq[b]; ++t, ++next) table[(size_t)b * MAX_CTX + t] = 1 + (next * 389) % 593;
  int64_t req_idx[B] = {0, 1, 2}, seq_lens[B] = {70, 513, 9};

  void *d_k, *d_v, *d_q, *d_out, *d_table, *d_req, *d_seq, *d_ws = NULL;
  CHECK_HIP(hipMalloc(&d_k, slab_elems * 2));
  CHECK_HIP(hipMalloc(&d_v, slab_elems * 2));
  CHECK_HIP(hipMalloc(&d_q, (size_t)B * HQ * D * 2));
  CHECK_HIP(hipMalloc(&d_out, (size_t)B * HQ * D * 2));
  CHECK_HIP(hipMalloc(&d_table, (size_t)B * MAX_CTX * 4));
  CHECK_HIP(hipMalloc(&d_req, sizeof req_idx));
  CHECK_HIP(hipMalloc(&d_seq, sizeof seq_lens));
  CHECK_HIP(hipMemcpy(d_k, k_slab, slab_elems * 2, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_v, v_slab, slab_elems * 2, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_q, q, (size_t)B * HQ * D * 2, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_table, table, (size_t)B * MAX_CTX * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_req, req_idx, sizeof req_idx, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_seq, seq_lens, sizeof seq_lens, hipMemcpyHostToDevice));

  /* the caller owns the split-merge workspace: sized by the library, zero-filled once */
  const int32_t n_splits = bsgl_decode_num_splits(B, HKV, MAX_CTX);
  const size_t ws_bytes = bsgl_decode_workspace_bytes(B, HQ, D, n_splits, 0);
  if (ws_bytes) {
    CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemset(d_ws, 0, ws_bytes));
  }
  const int rc = bsgl_paged_decode(d_out, HQ * D, NULL, d_q, HQ * D, NULL, NULL, 0, 0, NULL,
                                   d_k, d_v, HKV * D, N_SLOTS, (const int32_t*)d_table, MAX_CTX, B, MAX_CTX,
                                   (const int64_t*)d_req, (const int64_t*)d_seq, B, HQ, HKV, D, 1.0f / sqrtf((float)D),
                                   BSGL_DTYPE_BF16, BSGL_DTYPE_BF16, n_splits, BSGL_DECODE_AUTO, d_ws, ws_bytes, NULL);
  if (rc != BSGL_OK) { fprintf(stderr, "bsgl_paged_decode: %d (%s)\n", rc, bsgl_last_error()); return 1; }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(out, d_out, (size_t)B * HQ * D * 2, hipMemcpyDeviceToHost));

  double worst = 0.0;
  float* s = malloc(MAX_CTX * sizeof(float));
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < HQ; ++h) {
      const int kvh = h / G;
      float m = -INFINITY;
      for (int t = 0; t < seq[b]; ++t) {
        const size_t row = ((size_t)table[(size_t)b * MAX_CTX + t] * HKV + kvh) * D;
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc += from_bf16(q[((size_t)b * HQ + h) * D + d]) * from_bf16(k_slab[row + d]);
        s[t] = acc / sqrtf((float)D);
        if (s[t] > m) m = s[t];
      }
      double l = 0.0;
      for (int t = 0; t < seq[b]; ++t) { s[t] = expf(s[t] - m); l += s[t]; }
      for (int d = 0; d < D; ++d) {
        double o = 0.0;
        for (int t = 0; t < seq[b]; ++t)
          o += (double)s[t] * from_bf16(v_slab[((size_t)table[(size_t)b * MAX_CTX + t] * HKV + kvh) * D + d]);
        const double err = fabs(o / l - (double)from_bf16(out[((size_t)b * HQ + h) * D + d]));
        if (err > worst) worst = err;
      }
    }
  printf("bsgl ABI %d: %d requests decoded with %d splits, workspace %zu bytes, max |gpu - host| = %.3e\n",
         bsgl_abi_version(), B, (int)n_splits, ws_bytes, worst);
  return worst < 4e-3 ? 0 : 1;  /* bf16 output: half a unit in the last place of values below 1 */
}
