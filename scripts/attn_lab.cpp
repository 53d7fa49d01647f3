// Attention kernel lab (no Python): times dk_attention_bf16 variants on the BASELINE shapes and checks
// them against variant 0 (validated against the CPU oracle by tests/test_gpu_ops.py).
//   scripts/build_lab.sh builds build_lab/attn_lab;  usage: attn_lab [iters] [shape] [mode]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dk_hip.h"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t rng_state = 777;
static float nrand() {  // ~N(0,1): sum of 4 uniforms
  float s = 0;
  for (int i = 0; i < 4; ++i) {
    rng_state = rng_state * 1664525u + 1013904223u;
    s += ((rng_state >> 8) & 0xffffff) / 16777216.0f - 0.5f;
  }
  return s * 1.7320508f;
}

struct Shape { int B, H, S, D; const char* name; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;
  const int only_mode = argc > 3 ? atoi(argv[3]) : -999;
  std::vector<Shape> shapes = {{1, 24, 4352, 128, "flux joint B1"}, {2, 24, 4685, 64, "sd3 joint cfg B2"}, {4, 24, 4352, 128, "flux B4"},
                               {1, 24, 4608, 128, "flux-dev joint B1"}, {1, 3, 1000, 128, "small ragged"}};
  std::vector<int> modes = {0, 1};
  if (getenv("LAB_MODES")) {
    modes.clear();
    for (char* t = strtok(strdup(getenv("LAB_MODES")), ","); t; t = strtok(nullptr, ",")) modes.push_back(atoi(t));
  }
  const int NV = (int)modes.size();
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int idx = -1;
  for (const Shape& s : shapes) {
    ++idx;
    if (only_shape >= 0 && idx != only_shape) continue;
    const int h = s.H * s.D;
    const int ld = 3 * h + (getenv("LAB_PAD") ? atoi(getenv("LAB_PAD")) : 0);  // row pitch of the packed q|k|v buffer
    const size_t n = (size_t)s.B * s.S * ld;
    std::vector<uint16_t> hq(n);
    for (size_t i = 0; i < n; ++i) hq[i] = f2bf(nrand());
    void* qkv;
    CK(hipMalloc(&qkv, n * 2));
    CK(hipMemcpy(qkv, hq.data(), n * 2, hipMemcpyHostToDevice));
    const size_t no = (size_t)s.B * s.S * h;
    std::vector<void*> out(NV);
    for (int v = 0; v < NV; ++v) CK(hipMalloc(&out[v], no * 2));
    const float scale = 1.0f / sqrtf((float)s.D);
    auto run = [&](int v) {
      dk_tune_set("attn", only_mode != -999 ? only_mode : modes[v]);
      if (dk_attention_bf16(qkv, (char*)qkv + h * 2, (char*)qkv + 2 * h * 2, out[v], s.B, s.H, s.S, s.D, ld, h, scale, st) != 0) {
        printf("launch failed: %s\n", dk_last_error());
        exit(1);
      }
    };
    if (only_mode != -999) {
      for (int i = 0; i < iters; ++i) run(0);
      CK(hipStreamSynchronize(st));
      printf("ran %s mode %d x %d\n", s.name, only_mode, iters);
      continue;
    }
    std::vector<uint16_t> ref(no), got(no);
    for (int v = 0; v < NV; ++v) {
      CK(hipMemsetAsync(out[v], 0xff, no * 2, st));
      run(v);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(v == 0 ? ref.data() : got.data(), out[v], no * 2, hipMemcpyDeviceToHost));
      if (v > 0) {
        double maxd = 0, sum2 = 0, ref2 = 0;
        for (size_t i = 0; i < no; ++i) {
          const double a = bf2f(ref[i]), b = bf2f(got[i]);
          const double d = fabs(a - b);
          if (!(d <= maxd)) maxd = d;
          sum2 += d * d;
          ref2 += a * a;
        }
        const double rel = sqrt(sum2 / (ref2 + 1e-30));
        printf("  check %-18s mode %d vs mode %d: rel-L2 %.3e max abs %.3e %s\n", s.name, modes[v], modes[0], rel, maxd,
               (rel < 4e-3 && maxd < 0.03) ? "OK" : "MISMATCH");
      }
    }
    std::vector<double> best(NV, 1e30);
    for (int r = 0; r < 5; ++r)
      for (int v = 0; v < NV; ++v) {
        run(v);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run(v);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / iters < best[v]) best[v] = ms / iters;
      }
    const double fl = 4.0 * s.B * s.H * (double)s.S * s.S * s.D;
    printf("%-20s B%d H%d S%d D%d ", s.name, s.B, s.H, s.S, s.D);
    for (int v = 0; v < NV; ++v) printf(" m%d: %7.1f TF (%.3f ms)", modes[v], fl / best[v] / 1e9, best[v]);
    printf("\n");
    fflush(stdout);
    CK(hipFree(qkv));
    for (int v = 0; v < NV; ++v) CK(hipFree(out[v]));
  }
  dk_tune_set("attn", -1);
  return 0;
}
