// Probe 4: one wave per SIMD at 32 frames/wave (the shipped regime) against two waves per SIMD at 16 frames/wave, with the
// layer kernel's dynamic instruction mix (0.9 VALU and 1/8 resp. 1/4 weight-fragment loads per MFMA) and a real 1.6 MB weight
// stream read by every wave (L2 / L1 resident, like the packed layer weights).
//   MF  = MFMAs per weight fragment (8 at 32 frames/wave, 4 at 16 frames/wave)
//   WPS = waves per SIMD (blocks of 256*WPS threads, one block per CU, forced by a 96 KiB LDS allocation)
// Work per wave-iteration: 4 fragment loads (burst), 4*MF MFMAs on 4 chains, VAL VALU instructions in one clump.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template <int MF, int WPS, int VAL>
__global__ __launch_bounds__(256 * WPS) void stream(const f32x4* __restrict__ w, float* out, int nfrag, int reps) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 in[2] = {{0.37f * lane, 1.1f, 0.3f, 0.5f}, {0.1f, 0.2f * lane, 0.3f, 0.4f}};
  float x0 = 1.0f + lane, x1 = 0.5f;
  const float k = 0.999f;
  const f32x4* p = w + lane;
  f32x4 ring[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = p[i * 64];
  int pos = 8;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // consume 4 fragments
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const f32x4 a = ring[4 * half + f];
#pragma unroll
        for (int m = 0; m < MF; ++m)
          acc[(f * MF + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m & 3], in[m & 1][m & 3], acc[(f * MF + m) & 3], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // refill them in one burst
#pragma unroll
      for (int f = 0; f < 4; ++f) ring[4 * half + f] = p[(size_t)(pos + f) * 64];
      pos = pos + 8 > nfrag ? 0 : pos + 4;  // nfrag is a multiple of 4
      __builtin_amdgcn_sched_barrier(0);
      // VALU clump
#pragma unroll
      for (int v = 0; v < VAL / 2; ++v) { x0 = __builtin_fmaf(x0, k, k); x1 = __builtin_fmaf(x1, k, x0); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + x0 + x1 + (lds[0] = 0.f);
}

template <int MF, int WPS, int VAL>
void run(const f32x4* w, float* out, int nfrag, const char* label) {
  const int reps = 4000 / WPS * (8 / MF);  // same MFMA count per SIMD in every variant
  auto launch = [&] { hipLaunchKernelGGL((stream<MF, WPS, VAL>), dim3(256), dim3(256 * WPS), 96 * 1024, 0, w, out, nfrag, reps); };
  CK(hipFuncSetAttribute((const void*)stream<MF, WPS, VAL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  launch(); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma = 256.0 * 4 * WPS * reps * 8.0 * MF;  // per launch
  printf("%-56s %7.1f TFLOP/s (%4.1f %% of 157.3)\n", label, mfma * 2048 / (ms * 1e-3) / 1e12, mfma * 2048 / (ms * 1e-3) / 157.3e10);
}

int main() {
  const int nfrag = 1600;  // 1.6 MB stream
  f32x4* w; CK(hipMalloc(&w, (size_t)nfrag * 64 * sizeof(f32x4))); CK(hipMemset(w, 0, (size_t)nfrag * 64 * sizeof(f32x4)));
  float* out; CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  run<8, 1, 0>(w, out, nfrag, "1 wave/SIMD, 8 MFMA/fragment, no VALU");
  run<8, 1, 28>(w, out, nfrag, "1 wave/SIMD, 8 MFMA/fragment, 0.9 VALU/MFMA  (shipped)");
  run<8, 1, 56>(w, out, nfrag, "1 wave/SIMD, 8 MFMA/fragment, 1.75 VALU/MFMA (attention)");
  run<4, 2, 0>(w, out, nfrag, "2 waves/SIMD, 4 MFMA/fragment, no VALU");
  run<4, 2, 14>(w, out, nfrag, "2 waves/SIMD, 4 MFMA/fragment, 0.9 VALU/MFMA");
  run<4, 2, 28>(w, out, nfrag, "2 waves/SIMD, 4 MFMA/fragment, 1.75 VALU/MFMA");
  run<8, 2, 28>(w, out, nfrag, "2 waves/SIMD, 8 MFMA/fragment, 0.9 VALU/MFMA  (if it fitted)");
  run<8, 2, 56>(w, out, nfrag, "2 waves/SIMD, 8 MFMA/fragment, 1.75 VALU/MFMA (if it fitted)");
  return 0;
}
