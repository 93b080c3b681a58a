// One wave per SIMD, back-to-back v_mfma_f32_16x16x4_f32 over 16 accumulator tiles: does the ORDER in which the independent chains
// are visited change the issue rate?  (mfma_clock_probe: 4 chains round-robin 34.8 cycles per MFMA, 8 chains 32.5.)
//   pattern 0: pairs   -- A B A B A B A B | C D C D ...        (the k-major phase: 2 frame tiles x 4 k-steps per n-tile)
//   pattern 1: quads   -- A B C D A B C D A B C D A B C D | E F G H ...   (n-tile pairs x 2 frame tiles)
//   pattern 2: octets  -- 8 chains round-robin x 4, then the other 8
//   pattern 3: all 16 round-robin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <int G>  // G = chains visited round-robin before moving on (2, 4, 8, 16)
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* av) {
  f4 acc[16];
  for (int c = 0; c < 16; ++c) acc[c] = f4{0, 0, 0, 0};
  float a[4], b[4];
  for (int r = 0; r < 4; ++r) { a[r] = av[threadIdx.x + 256 * r]; b[r] = av[1024 + threadIdx.x + 256 * r]; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += G)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < G; ++c) acc[g0 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + c) & 3], acc[g0 + c], 0, 0, 0);
  }
  f4 s = acc[0];
  for (int c = 1; c < 16; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
  float *out, *av; CK(hipMalloc(&out, (1 << 20) * sizeof(float))); CK(hipMalloc(&av, 2048 * sizeof(float))); CK(hipMemset(av, 0, 2048 * sizeof(float)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto one = [&](auto kern, int G, int blocks) {
    const int iters = 10000;
    kern<<<blocks, 256>>>(out, 500, av); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, iters, av); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = (double)blocks * 4 * iters * 64;
    printf("%d waves/SIMD, groups of %2d chains: %.2f cycles per MFMA per SIMD at 2.4 GHz\n", blocks / 256, G, ms * 1e-3 * 2.4e9 / (mf / 1024.0));
  };
  for (int blocks : {256, 512}) { one(k<2>, 2, blocks); one(k<4>, 4, blocks); one(k<8>, 8, blocks); one(k<16>, 16, blocks); }
  return 0;
}
