// Cost of a TAKEN forward scalar branch (over a block of never-executed instructions) between runs of 16 fp32 MFMAs, one wave per
// SIMD -- the layer kernel's attention step takes three per step (skip of the rare rescale path, interior/edge select, loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
#define SKIP64 "v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n v_mov_b32 v255, v255\n"
#define SKIPBLK SKIP64 SKIP64 SKIP64 SKIP64 SKIP64 SKIP64 SKIP64 SKIP64
template <int NB, bool TAKEN>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* av, int zero) {
  f4 acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = f4{0, 0, 0, 0};
  float a = av[threadIdx.x], b = av[256 + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (TAKEN) asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n" SKIPBLK "1:\n" ::"s"(zero) : "v255", "scc");
      else asm volatile("s_cmp_eq_u32 %0, 1\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n" ::"s"(zero) : "scc");  // not taken
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  f4 s = acc[0];
  for (int c = 1; c < 8; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
  float *out, *av; CK(hipMalloc(&out, (1 << 20) * sizeof(float))); CK(hipMalloc(&av, 1024 * sizeof(float))); CK(hipMemset(av, 0, 1024 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto one = [&](auto kern, int n, const char* what) {
    const int iters = 20000, blocks = 256;
    kern<<<blocks, 256>>>(out, 500, av, 0); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, iters, av, 0); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("16 MFMAs + %d %s: %.1f cycles per iteration at 2.4 GHz\n", n, what, ms * 1e-3 * 2.4e9 / iters);
  };
  one(k<0, true>, 0, "-");
  one(k<1, true>, 1, "taken forward branches (skip 64 instr)"); one(k<2, true>, 2, "taken forward branches"); one(k<4, true>, 4, "taken forward branches");
  one(k<1, false>, 1, "not-taken branches"); one(k<2, false>, 2, "not-taken branches"); one(k<4, false>, 4, "not-taken branches");
  return 0;
}
