// Probe: (1) lane layout of v_mfma_f32_16x16x4_f32, (2) rate of a register-chained GEMM stream
// whose A operand (weights) is streamed from L2 in packed 1-KiB fragments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void layout_kernel(const float* A /*16x4 row-major [i][k]*/, const float* B /*4x16 [k][j]*/, float* D /*16x16 [i][j]*/) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// Stream kernel: each wave owns 32 "frames" (2 frame tiles), input x^T in C-layout regs in[KT][2],
// computes NT output tiles of out^T = W x^T, W packed [nt][kt][lane][4]. Repeats `reps` times.
template <int KT, int NT>
__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ wp, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 in[KT][2];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) { in[kt][ft] = f32x4{(float)(lane + kt), 1.f, (float)ft, 0.5f}; }
  f32x4 sum0 = {0, 0, 0, 0}, sum1 = {0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    const f32x4* w = wp + lane;
    f32x4 frag[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) frag[kt] = w[kt * 64];
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      const f32x4* wn = w + (size_t)((nt + 1 < NT ? nt + 1 : 0) * KT) * 64;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 a = frag[kt];
        frag[kt] = wn[kt * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], in[kt][0][r], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], in[kt][1][r], acc1, 0, 0, 0);
        }
      }
      sum0 += acc0; sum1 += acc1;
    }
  }
  f32x4 s = sum0 + sum1;
  out[(blockIdx.x * blockDim.x + threadIdx.x)] = s[0] + s[1] + s[2] + s[3];
}

int main() {
  // ---- layout ----
  std::vector<float> A(64), B(64), D(256), Dref(256, 0.f);
  for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7 + 3) % 11 - 5); B[i] = (float)((i * 5 + 1) % 13 - 6); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) Dref[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  float *dA, *dB, *dD; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != Dref[i]) ++bad;
  printf("layout check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  // ---- stream rate ----
  constexpr int KT = 10, NT = 120;  // one "layer" worth: 120 n-tiles x K=160
  size_t wfloats = (size_t)NT * KT * 64 * 4;
  std::vector<float> W(wfloats); for (size_t i = 0; i < wfloats; ++i) W[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  f32x4* dW; CK(hipMalloc(&dW, wfloats * 4)); CK(hipMemcpy(dW, W.data(), wfloats * 4, hipMemcpyHostToDevice));
  float* dO; CK(hipMalloc(&dO, 4096 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {256, 512, 1024}) {
    int reps = 8;
    stream_kernel<KT, NT><<<blocks, 256>>>(dW, dO, 1); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); stream_kernel<KT, NT><<<blocks, 256>>>(dW, dO, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4 /*waves*/ * reps * NT * KT * 8 /*mfma*/ * 2048.0;
    printf("stream blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flops / ms * 1e-9);
  }
  return 0;
}
