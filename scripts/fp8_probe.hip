// fp8 probe (lab only): pins on hardware what the fp8 GEMM (diffusionkit_amd/csrc/gemm256f8.hip) assumes about gfx950's
// block-scaled MFMA and its fp8 conversion, and measures the instruction's issue rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/fp8_probe.hip -o build_lab/fp8_probe && build_lab/fp8_probe
//
// 1. v_mfma_scale_f32_16x16x128_f8f6f4 with both operands e4m3.  Test 1 / 2 state the FIRST hypothesis -- operand lane l holds row
//    (l & 15), the 32 K-elements [32 * (l >> 4), +32), and its scale byte scales exactly those -- and FAIL on hardware; section 1b
//    maps which register positions a lane's scale governs.  Result (profiles/r02_fp8_probe.log): lane (r, q) holds
//    K [16 q, 16 q + 16) in registers 0-3 and K [64 + 16 q, +16) in registers 4-7; the scale byte of lane (r, g) scales row r,
//    K [32 g, 32 g + 32).  Test 4 states that and passes.  C/D: lane l holds column l & 15, rows 4 * (l >> 4) + {0..3}.
// 2. v_cvt_pk_fp8_f32: round-to-nearest-even OCP e4m3, what happens above 448.
// 3. rate: 256 workgroups x 8 waves, 8 independent accumulators.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x)                                                            \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

static float e4m3_decode(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) v = NAN;
  else v = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}
// round-to-nearest-even, saturating at 448
static uint8_t e4m3_encode_sat(float x) {
  const uint8_t s = std::signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (std::isnan(a)) return s | 0x7F;
  if (a >= 448.f) return s | 0x7E;
  // search the nearest representable (126 positive codes): fine for a probe
  int best = 0;
  float bd = 1e30f;
  for (int c = 0; c < 0x7F; ++c) {
    const float d = fabsf(e4m3_decode((uint8_t)c) - a);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
  }
  return s | (uint8_t)best;
}

// one wave: D = A (16 x 128) . B^T (16 x 128) with per-lane scale registers; OPSEL selects the byte of the scale registers
template <int OPSEL>
__global__ void mfma_once(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* D) {
  const int l = threadIdx.x, row = l & 15, kb = l >> 4;
  v8i a, b;
  for (int j = 0; j < 8; ++j) {
    // OPSEL 4 = the layout found on hardware: registers 0-3 <- K [16 kb, +16), registers 4-7 <- K [64 + 16 kb, +16)
    const int off = OPSEL == 4 ? (j < 4 ? kb * 16 + 4 * j : 64 + kb * 16 + 4 * (j - 4)) : kb * 32 + 4 * j;
    a[j] = *(const int*)(A + row * 128 + off);
    b[j] = *(const int*)(B + row * 128 + off);
  }
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, OPSEL & 3, sa[l], OPSEL & 3, sb[l]);
  for (int r = 0; r < 4; ++r) D[(4 * kb + r) * 16 + row] = acc[r];  // D[i][j]: i = A row, j = B row
}

__global__ void cvt_sweep(const float* x, uint8_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n + 1) return;
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], w, false);
  out[2 * i] = (uint8_t)(w & 0xFF);
  out[2 * i + 1] = (uint8_t)((w >> 8) & 0xFF);
}

template <int SCALED>
__global__ __launch_bounds__(512, 2) void rate(const v8i* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  v8i a[2], b[4];
  for (int i = 0; i < 2; ++i) a[i] = in[(blockIdx.x * 512 + tid) * 6 + i];
  for (int i = 0; i < 4; ++i) b[i] = in[(blockIdx.x * 512 + tid) * 6 + 2 + i];
  v4f acc[2][4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (SCALED) {
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
          } else {  // the four non-scaled 16x16x32 fp8 MFMAs over the same 32 bytes per lane
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const long av = ((long)a[i][2 * s + 1] << 32) | (unsigned)a[i][2 * s];
              const long bv = ((long)b[j][2 * s + 1] << 32) | (unsigned)b[j][2 * s];
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av, bv, acc[i][j], 0, 0, 0);
            }
          }
        }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  out[blockIdx.x * 512 + tid] = s;
}

int main() {
  srand(1);
  // ---- 1. layout + scale semantics ----
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  for (auto* v : {&A, &B})
    for (auto& x : *v) {
      do x = (uint8_t)(rand() & 0xFF); while ((x & 0x7F) == 0x7F || ((x >> 3) & 15) > 9);  // |value| < 8: no overflow, no NaN
    }
  uint8_t *dA, *dB;
  int *dsa, *dsb;
  float* dD;
  CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
  for (int test = 0; test < 5; ++test) {
    // test 0: all scales 1 (byte 0); 1: random per-lane scales in byte 0; 2: random scales in byte 2, garbage elsewhere, op_sel 2;
    // 3: scale registers vary only with the row (lanes l, l+16, .. share it) -- a cross-check of the (row, block) association
    std::vector<int> sa(64), sb(64);
    std::vector<int> ea(64), eb(64);
    for (int l = 0; l < 64; ++l) {
      // (test 4: random per-lane scales like test 1, checked against the layout section 1b found)
      ea[l] = test == 0 ? 127 : test == 3 ? 120 + (l & 15) : 121 + rand() % 12;
      eb[l] = test == 0 ? 127 : test == 3 ? 127 : 121 + rand() % 12;
      if (test == 2) {
        sa[l] = (rand() & 0xFF) | ((rand() & 0xFF) << 8) | (ea[l] << 16) | ((rand() & 0x7F) << 24);
        sb[l] = (rand() & 0xFF) | ((rand() & 0xFF) << 8) | (eb[l] << 16) | ((rand() & 0x7F) << 24);
      } else {
        sa[l] = ea[l] | 0x11223300;
        sb[l] = eb[l] | 0x44556600;
      }
    }
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    if (test == 2) hipLaunchKernelGGL(mfma_once<2>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    else if (test == 4) hipLaunchKernelGGL(mfma_once<4>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    else hipLaunchKernelGGL(mfma_once<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    CK(hipDeviceSynchronize());
    std::vector<float> D(256);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double worst = 0, ref_max = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double acc = 0;
        for (int k = 0; k < 128; ++k)
          acc += (double)e4m3_decode(A[i * 128 + k]) * e4m3_decode(B[j * 128 + k]) * ldexp(1.0, ea[(k / 32) * 16 + i] - 127) *
                 ldexp(1.0, eb[(k / 32) * 16 + j] - 127);
        worst = fmax(worst, fabs(acc - D[i * 16 + j]));
        ref_max = fmax(ref_max, fabs(acc));
      }
    printf("mfma_scale test %d: max |D - ref| = %.3e (|ref| max %.3e)  %s\n", test, worst, ref_max, worst <= 1e-3 * ref_max ? "PASS" : "FAIL");  // (the MFMA's internal sum is not a full fp32 chain: ~1e-4 relative)
  }
  // ---- 1b. which data positions does lane L's scale govern?  A = ones, B = ones at register positions (lane group qb, byte half h)
  //          only, scale of lane L = 2^3, all others 2^0: D[i][j] = 16 * 8 if L governs (row i, qb, h), else 16
  {
    std::vector<uint8_t> A1(16 * 128, 0x38), B1(16 * 128);  // 0x38 = 1.0 in e4m3
    CK(hipMemcpy(dA, A1.data(), A1.size(), hipMemcpyHostToDevice));
    printf("scale map (lane L -> [row: (q,half) ...]), scale operand of src0 (A):\n");
    for (int side = 0; side < 2; ++side) {
      if (side == 1) printf("scale map, scale operand of src1 (B) -> [col: (q,half) ...]:\n");
      for (int L = 0; L < 64; ++L) {
        std::string line;
        for (int qb = 0; qb < 4; ++qb)
          for (int h = 0; h < 2; ++h) {
            for (int r = 0; r < 16; ++r)
              for (int k = 0; k < 128; ++k) B1[r * 128 + k] = (k / 32 == qb && (k % 32) / 16 == h) ? 0x38 : 0x00;
            // side 0: scaled operand = A (ones), selector pattern in B; side 1: roles swapped
            CK(hipMemcpy(side == 0 ? dB : dA, B1.data(), B1.size(), hipMemcpyHostToDevice));
            CK(hipMemcpy(side == 0 ? dA : dB, A1.data(), A1.size(), hipMemcpyHostToDevice));
            std::vector<int> sa(64, 127), sb(64, 127);
            (side == 0 ? sa : sb)[L] = 130;
            CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
            CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(mfma_once<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            CK(hipDeviceSynchronize());
            std::vector<float> D(256);
            CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
            for (int i = 0; i < 16; ++i) {
              const float v = side == 0 ? D[i * 16 + 0] : D[0 * 16 + i];
              if (v != 16.f) { char buf[64]; snprintf(buf, 64, " %d:(%d,%d)=%g", i, qb, h, v); line += buf; }
            }
          }
        printf("  L%2d ->%s\n", L, line.c_str());
      }
    }
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
  }
  // ---- 2. conversion ----
  {
    std::vector<float> x;
    for (int i = 0; i < 4000; ++i) {
      const float mag = expf(-9.f + 16.5f * (float)i / 4000.f);  // 1.2e-4 .. 1.8e3
      x.push_back((i & 1) ? -mag : mag);
    }
    for (int c = 0; c < 0x7E; ++c) {  // exact ties between neighbouring codes
      x.push_back(0.5f * (e4m3_decode((uint8_t)c) + e4m3_decode((uint8_t)(c + 1))));
      x.push_back(e4m3_decode((uint8_t)c));
    }
    x.push_back(448.f); x.push_back(464.f); x.push_back(465.f); x.push_back(480.f); x.push_back(1e6f); x.push_back(INFINITY);
    if (x.size() & 1) x.push_back(0.f);
    float* dx;
    uint8_t* dq;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dq, x.size()));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_sweep, dim3((x.size() / 2 + 63) / 64), dim3(64), 0, 0, dx, dq, (int)x.size());
    CK(hipDeviceSynchronize());
    std::vector<uint8_t> q(x.size());
    CK(hipMemcpy(q.data(), dq, x.size(), hipMemcpyDeviceToHost));
    int bad_in = 0, bad_out = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      const uint8_t want = e4m3_encode_sat(x[i]);
      if (q[i] != want) {
        if (fabsf(x[i]) <= 448.f) { if (bad_in++ < 8) printf("  cvt(%.6g) = 0x%02x, RNE wants 0x%02x\n", x[i], q[i], want); }
        else { if (bad_out++ < 8) printf("  cvt(%.6g) = 0x%02x (saturating RNE would give 0x%02x)\n", x[i], q[i], want); }
      }
    }
    printf("cvt_pk_fp8_f32: %zu values, %d mismatches inside |x| <= 448 (%s), %d above (saturation %s)\n", x.size(), bad_in,
           bad_in == 0 ? "PASS" : "FAIL", bad_out, bad_out == 0 ? "yes" : "NO");
  }
  // ---- 3. rate ----
  {
    const int nwg = 256, iters = 4000;
    std::vector<uint8_t> in((size_t)nwg * 512 * 6 * 32);
    for (auto& x : in) do x = (uint8_t)(rand() & 0xFF); while ((x & 0x7F) == 0x7F);
    v8i* din;
    float* dout;
    CK(hipMalloc(&din, in.size())); CK(hipMalloc(&dout, (size_t)nwg * 512 * 4));
    CK(hipMemcpy(din, in.data(), in.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int scaled = 1; scaled >= 0; --scaled) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (scaled) hipLaunchKernelGGL(rate<1>, dim3(nwg), dim3(512), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(rate<0>, dim3(nwg), dim3(512), 0, 0, din, dout, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = 2.0 * 16 * 16 * 128 * 32.0 * iters * nwg * 8;
        if (rep) printf("rate %s: %.3f ms, %.0f TFLOP/s (random e4m3 operands)\n", scaled ? "mfma_scale 16x16x128 (scales = 1)" : "4 x mfma 16x16x32 fp8", ms,
                        flop / ms * 1e-9);
      }
    }
  }
  return 0;
}
