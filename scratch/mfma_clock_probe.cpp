// Sustained rate of back-to-back v_mfma_f32_16x16x4_f32 on the WHOLE chip (1 and 2 waves per SIMD, 256 CUs): the practical
// ceiling of any fp32-MFMA kernel at the clock the chip holds under this load (spec: 157.3 TFLOP/s at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  f4 s = acc[0];
  for (int c = 1; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(out + (1 << 20))[0] = t1 - t0;
}
int main() {
  float* out; CK(hipMalloc(&out, (2 << 20) * sizeof(float) + 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {256, 512, 1024}) {
    const int iters = 20000;
    k<4><<<blocks, 256>>>(out, 1000); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<4><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = (double)blocks * 4 * iters * 8 * 4, fl = mf * 2048;
    // waves per SIMD resident: blocks*4 waves over 1024 SIMDs (rounds when > 2 per SIMD fit: registers allow 8)
    printf("blocks %4d (%.1f waves/SIMD): %.3f ms, %.1f TFLOP/s = %.3f of 157.3; per-SIMD MFMA issue interval %.2f cycles at 2.4 GHz\n",
           blocks, blocks * 4 / 1024.0, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, ms * 1e-3 * 2.4e9 / (mf / 1024.0));
  }
  // one wave per SIMD: does the number of independent accumulator chains matter?
  auto one = [&](auto kern, int chains) {
    const int iters = 20000, blocks = 256;
    kern<<<blocks, 256>>>(out, 1000); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = (double)blocks * 4 * iters * 8 * chains;
    printf("1 wave/SIMD, %d chains: issue interval %.2f cycles at 2.4 GHz\n", chains, ms * 1e-3 * 2.4e9 / (mf / 1024.0));
  };
  one(k<2>, 2); one(k<8>, 8); one(k<16>, 16);
  return 0;
}
