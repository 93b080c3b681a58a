// Cost of n independent v_exp_f32 (vs n v_mul_f32) placed between runs of 16 fp32 MFMAs, one wave per SIMD: is the transcendental
// unit quarter rate (16 cycles per wave instruction) on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <int N, bool EXP>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* av) {
  f4 acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = f4{0, 0, 0, 0};
  float a = av[threadIdx.x], b = av[256 + threadIdx.x];
  float e[32];
  for (int i = 0; i < 32; ++i) e[i] = av[512 + ((threadIdx.x + i) & 255)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = EXP ? __builtin_amdgcn_exp2f(e[i]) : e[i] * 0.999f;
    __builtin_amdgcn_sched_barrier(0);
  }
  f4 s = acc[0];
  for (int c = 1; c < 8; ++c) s += acc[c];
  float t = 0;
  for (int i = 0; i < 32; ++i) t += e[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + t;
}
int main() {
  float *out, *av; CK(hipMalloc(&out, (1 << 20) * sizeof(float))); CK(hipMalloc(&av, 1024 * sizeof(float))); CK(hipMemset(av, 0, 1024 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto one = [&](auto kern, int n, const char* what) {
    const int iters = 20000, blocks = 256;
    kern<<<blocks, 256>>>(out, 500, av); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, iters, av); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("16 MFMAs + %2d %s: %.1f cycles per iteration at 2.4 GHz\n", n, what, ms * 1e-3 * 2.4e9 / iters);
  };
  one(k<0, true>, 0, "-");
  one(k<8, true>, 8, "v_exp_f32"); one(k<16, true>, 16, "v_exp_f32"); one(k<32, true>, 32, "v_exp_f32");
  one(k<8, false>, 8, "v_mul_f32"); one(k<16, false>, 16, "v_mul_f32"); one(k<32, false>, 32, "v_mul_f32");
  return 0;
}
