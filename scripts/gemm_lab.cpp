// Kernel lab for the GEMM / attention variants behind the C ABI (no Python, starts in milliseconds):
// checks every variant against the 128x128 GEMM kernel (itself validated against the CPU oracle by
// tests/test_gpu_ops.py) on random data, then times them interleaved.
//   hipcc -O2 -std=c++17 scripts/gemm_lab.cpp -Iinclude -Ldiffusionkit_amd -ldk_hip -Wl,-rpath,$PWD/diffusionkit_amd -o gpurun_out/gemm_lab
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dk_hip.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
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
static uint32_t rng_state = 12345;
static float frand() {  // uniform [-1, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) & 0xffffff) / 8388608.0f - 1.0f;
}
static void* dev_random(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  static const bool zero = getenv("LAB_ZERO") != nullptr;  // zero operands: shows the DVFS headroom (guide rule 25)
  for (size_t i = 0; i < n; ++i) h[i] = zero ? 0 : f2bf(frand() * scale);
  void* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

struct Shape { int M, N, K; const char* name; int epi; };

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;  // run one shape ...
  const int only_mode = argc > 3 ? atoi(argv[3]) : -999;  // ... with one kernel mode (for rocprofv3 --pmc runs)
  std::vector<Shape> shapes = {
      {4096, 9216, 3072, "flux qkv img", DK_EPI_BIAS},
      {4096, 3072, 3072, "flux o img", DK_EPI_GATE_RES},
      {4096, 12288, 3072, "flux fc1 img", DK_EPI_BIAS_GELU},
      {4096, 3072, 12288, "flux fc2 img", DK_EPI_GATE_RES},
      {4352, 9216, 3072, "flux single qkv", DK_EPI_BIAS},
      {4352, 12288, 3072, "flux single fc1", DK_EPI_BIAS_GELU},
      {4352, 3072, 15360, "flux single l2", DK_EPI_GATE_RES},
      {8192, 4608, 1536, "sd3 qkv", DK_EPI_BIAS},
      {8192, 1536, 6144, "sd3 fc2", DK_EPI_GATE_RES},
      {1178, 4608, 1536, "sd3 qkv txt (ragged M)", DK_EPI_BIAS},
      {4096, 4096, 4096, "square 4096", DK_EPI_BIAS},
      {8192, 8192, 8192, "square 8192", DK_EPI_BIAS},
      {4096, 4096, 64, "fixed-cost K=64", DK_EPI_BIAS},
      {4096, 4096, 1024, "K=1024", DK_EPI_BIAS},
      {4096, 4096, 2048, "K=2048", DK_EPI_BIAS},
      {4096, 4096, 16384, "K=16384", DK_EPI_BIAS},
      {256, 12288, 3072, "flux fc1 txt (M=256)", DK_EPI_BIAS_GELU},
      {256, 12288, 4096, "t5 qkv (M=256)", DK_EPI_BIAS},
      {256, 4096, 10240, "t5 wo (M=256)", DK_EPI_BIAS},
      {1178, 6144, 1536, "sd3 fc1 txt", DK_EPI_BIAS_GELU},
      // round 5: the model's exact launches that the list above only approximates
      {4352, 21504, 3072, "flux single linear1", DK_EPI_BIAS},
      {8192, 1536, 1536, "sd3 o", DK_EPI_GATE_RES},
      {8192, 6144, 1536, "sd3 fc1", DK_EPI_BIAS_GELU},
      {4608, 21504, 3072, "flux-dev linear1", DK_EPI_BIAS},
      {4608, 3072, 15360, "flux-dev l2", DK_EPI_GATE_RES},
      {4096, 2432, 2432, "sd3.5 o (half tile)", DK_EPI_GATE_RES},
  };
  std::vector<int> modes = {128, 5, 6, 3};
  if (getenv("LAB_MODES")) {  // e.g. LAB_MODES=128,1,11,12 (>= 10: ablation builds, not checked)
    modes.clear();
    for (char* t = strtok(strdup(getenv("LAB_MODES")), ","); t; t = strtok(nullptr, ",")) modes.push_back(atoi(t));
  }
  const int NV = (int)modes.size();
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (getenv("LAB_SPLIT")) dk_tune_set("gemm_split", atoi(getenv("LAB_SPLIT")));
  if (getenv("LAB_MF")) dk_tune_set("gemm_mf", atoi(getenv("LAB_MF")));  // gemm256v3.hip tile height: 8 / 7 (224-row tiles), -1 automatic
  if (getenv("LAB_SKEW")) dk_tune_set("gemm_skew", atoi(getenv("LAB_SKEW")));  // gemm256v4.hip: start skew of multi-round launches, 0.25 us steps  // v3 remainder-wave K split: -1 auto, 0 off, 1 force
  void* ws = nullptr;
  const size_t ws_bytes = dk_gemm_workspace_bytes();
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMemset(ws, 0, ws_bytes));
  int shape_idx = -1;
  for (const Shape& s : shapes) {
    ++shape_idx;
    if (only_shape >= 0 && shape_idx != only_shape) continue;
    if (only_mode != -999) {
      void* A = dev_random((size_t)s.M * s.K, 1.0f);
      void* W = dev_random((size_t)s.N * s.K, 0.05f);
      void* Cc;
      CK(hipMalloc(&Cc, (size_t)s.M * s.N * 2));
      dk_gemm_desc d;
      memset(&d, 0, sizeof(d));
      d.A = A; d.W = W; d.C = Cc; d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.K; d.ldc = s.N; d.alpha = 1.0f; d.epilogue = DK_EPI_BIAS;
      d.workspace = ws; d.workspace_bytes = ws_bytes;
      dk_tune_set("gemm", only_mode);
      for (int i = 0; i < iters; ++i) dk_gemm_bf16(&d, st);
      CK(hipStreamSynchronize(st));
      printf("ran %s mode %d x %d\n", s.name, only_mode, iters);
      continue;
    }
    const int pad = getenv("LAB_PAD") ? atoi(getenv("LAB_PAD")) : 0;  // extra elements per A / W row
    void* A = dev_random((size_t)s.M * (s.K + pad), 1.0f);
    void* W = dev_random((size_t)s.N * (s.K + pad), 0.05f);
    void* bias = dev_random(s.N, 0.5f);
    void* gate = dev_random(s.N, 1.0f);
    void* res = dev_random((size_t)s.M * s.N, 1.0f);
    std::vector<void*> C(NV);
    for (int i = 0; i < NV; ++i) CK(hipMalloc(&C[i], (size_t)s.M * s.N * 2));
    dk_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.W = W; d.bias = bias; d.gate = gate; d.res = res;
    d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.K + pad; d.ldw = s.K + pad; d.ldc = s.N; d.ldr = s.N;
    d.gate_seg_len = s.M; d.gate_stride = s.N; d.alpha = 1.0f; d.epilogue = s.epi;
    d.workspace = ws; d.workspace_bytes = ws_bytes;
    std::vector<uint16_t> ref((size_t)s.M * s.N), got((size_t)s.M * s.N);
    std::vector<double> best(NV, 1e30);
    std::vector<bool> skip(NV, false);
    for (int v = 0; v < NV; ++v) {
      dk_tune_set("gemm", modes[v]);
      CK(hipMemsetAsync(C[v], 0xff, (size_t)s.M * s.N * 2, st));
      d.C = C[v];
      if (dk_gemm_bf16(&d, st) != 0) { printf("  mode %d not applicable: %s\n", modes[v], dk_last_error()); skip[v] = true; continue; }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(v == 0 ? ref.data() : got.data(), C[v], (size_t)s.M * s.N * 2, hipMemcpyDeviceToHost));
      if (v > 0 && modes[v] <= 13) {
        size_t bad = 0;
        double maxd = 0;
        for (size_t i = 0; i < ref.size(); ++i) {
          if (ref[i] != got[i]) {
            ++bad;
            double dd = fabs((double)bf2f(ref[i]) - (double)bf2f(got[i]));
            if (!(dd <= maxd)) maxd = dd;
          }
        }
        printf("  check %-24s mode %3d vs 128: %zu / %zu differ, max abs %.4g %s\n", s.name, modes[v], bad, ref.size(), maxd,
               bad == 0 ? "OK" : (maxd < 0.07 && bad < ref.size() / 50 ? "(rounding-level) OK" : "MISMATCH"));
      }
    }
    // interleaved timing rounds
    for (int r = 0; r < 5; ++r)
      for (int v = 0; v < NV; ++v) {
        if (skip[v]) continue;
        dk_tune_set("gemm", modes[v]);
        d.C = C[v];
        dk_gemm_bf16(&d, st);  // warm
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) dk_gemm_bf16(&d, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / iters < best[v]) best[v] = ms / iters;
      }
    if (getenv("LAB_TRACE")) {  // a -DV4_TRACE build of gemm256v4.hip (scripts/build_lab.sh TRACE=1): s_memtime stamps of workgroups 0 and last, wave 0
      dk_tune_set("gemm", 10);
      d.C = C[0];
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ws, 0, 256));
        dk_gemm_bf16(&d, st);
        CK(hipStreamSynchronize(st));
        unsigned long long t[16];
        CK(hipMemcpy(t, ws, sizeof(t), hipMemcpyDeviceToHost));
        for (int b = 0; b < 2; ++b) {
          const unsigned long long* o = t + 8 * b;
          if (o[5] == 0) continue;
          printf("  trace %-22s %s workgroup (shader cycles): entry -> first DMA %5llu | -> first K-tile landed %6llu | K loop %8llu | drain %6llu | tail (incl. vmcnt(0) of its stores) %6llu | total %8llu\n",
                 s.name, b == 0 ? "first" : "last ", o[1] - o[0], o[2] - o[1], o[3] - o[2], o[4] - o[3], o[5] - o[4], o[5] - o[0]);
        }
      }
      CK(hipMemset(ws, 0, 256));
    }
    const double fl = 2.0 * s.M * s.N * s.K;
    printf("%-26s %5dx%5dx%5d ", s.name, s.M, s.N, s.K);
    for (int v = 0; v < NV; ++v) printf(" m%d: %7.1f TF", modes[v], fl / best[v] / 1e9);
    printf("  us:");
    for (int v = 0; v < NV; ++v) printf(" %.1f", best[v] * 1e3);
    printf("   (best of 5 x %d)\n", iters);
    fflush(stdout);
    hipFree(A); hipFree(W); hipFree(bias); hipFree(gate); hipFree(res);
    for (int i = 0; i < NV; ++i) hipFree(C[i]);
  }
  dk_tune_set("gemm", -1);
  return 0;
}
