// Standalone probe of the bf16 GEMM phase loop (edtts_bf16.h): where do the cycles of a 16-fragment phase go?
//   hipcc -O3 --offload-arch=gfx950 scratch/ring_probe.cpp -o scratch/ring_probe && scratch/ring_probe
// Variants (cumulative): 0 MFMAs on register operands only; 1 + fragments read from LDS; 2 + one barrier per phase;
//                        3 + LDS-DMA refill of the ring (4 x 1 KiB per wave and phase) with the counted vmcnt wait; 4 = 3 + epilogue stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
constexpr int PH = 16, NS = 6, WAVES = 4;

template <int V>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ stream, f4* __restrict__ out, int phases) {
  extern __shared__ __attribute__((aligned(16))) f4 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf8 in[8][2];
  for (int i = 0; i < 8; ++i)
    for (int ft = 0; ft < 2; ++ft) in[i][ft] = __builtin_bit_cast(bf8, stream[(i * 2 + ft) * 64 + lane]);
  const f4* src = stream + lane;
  auto issue = [&](int phase) {
    const int slot = phase % NS;
    for (int i = 0; i < PH / WAVES; ++i) {
      const int f = (PH / WAVES) * wave + i;
      __builtin_amdgcn_global_load_lds(src + ((size_t)(phase % 64) * PH + f) * 64, (lds_ptr_t)(lds + (slot * PH + f) * 64), 16, 0, 0);
    }
  };
  if (V >= 3) for (int p = 0; p < NS - 1; ++p) issue(p);
  else {  // fill the ring once
    for (int i = threadIdx.x; i < NS * PH * 64; i += 256) lds[i] = stream[i];
    __syncthreads();
  }
  f4 acc[4] = {};
  f4 keep = {};
  for (int p = 0; p < phases; ++p) {
    if (V == 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PH / WAVES) * (NS - 2) + 2 * (NS - 1)) : "memory");
    else if (V >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PH / WAVES) * (NS - 2)) : "memory");
    if (V >= 2) __builtin_amdgcn_s_barrier();
    if (V >= 3) issue(p + NS - 1);
    f4 fg[PH];
    if (V >= 1) {
      const f4* fr = lds + (p % NS) * PH * 64 + lane;
#pragma unroll
      for (int i = 0; i < PH; ++i) fg[i] = fr[i * 64];
    } else {
#pragma unroll
      for (int i = 0; i < PH; ++i) fg[i] = __builtin_bit_cast(f4, in[i & 7][0]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const bf8 fa = __builtin_bit_cast(bf8, fg[2 * kt]), fb = __builtin_bit_cast(bf8, fg[2 * kt + 1]);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        acc[ft] = MFMA16(fa, in[kt][ft], acc[ft]);
        acc[2 + ft] = MFMA16(fb, in[kt][ft], acc[2 + ft]);
      }
    }
    if (V >= 4) {
      f4* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + (size_t)(p & 7) * gridDim.x * 1024;
      if (V == 4) {
        __builtin_nontemporal_store(acc[0] + acc[2], o);
        __builtin_nontemporal_store(acc[1] + acc[3], o + 1);
      } else {
        o[0] = acc[0] + acc[2];
        o[1] = acc[1] + acc[3];
      }
      acc[0] = acc[1] = acc[2] = acc[3] = f4{0, 0, 0, 0};
    } else {
      keep += acc[0];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = keep + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int V>
void run(const f4* stream, f4* out, int blocks, int phases) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const size_t ldsb = NS * PH * 1024;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), ldsb, 0, stream, out, phases);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), ldsb, 0, stream, out, phases);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double waves_per_simd = blocks / 256.0;  // one 4-wave block per CU at a time
  const double cyc = ms * 1e-3 * 2.4e9 / waves_per_simd / phases;
  const double tf = (double)blocks * 4 * phases * 32 * 16384 / (ms * 1e-3) / 1e12;
  printf("variant %d: %.3f ms, %.0f cycles (at 2.4 GHz) per 32-MFMA phase (ideal 512), %.0f TFLOP/s\n", V, ms, cyc, tf);
}

int main() {
  const size_t n = (size_t)80 * PH * 64;  // 80 phases of fragments
  std::vector<float> h(n * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  f4 *stream, *out;
  hipMalloc(&stream, n * 16);
  hipMalloc(&out, ((size_t)2048 * 256 * 4 + (size_t)8 * 2048 * 1024 + 1024) * sizeof(f4));  // covers the variant-4 store pattern: (block*256+tid)*4 + 1 + (p&7)*blocks*1024
  hipMemcpy(stream, h.data(), n * 16, hipMemcpyHostToDevice);
  const int blocks = 2048, phases = 96;
  run<0>(stream, out, blocks, phases);
  run<1>(stream, out, blocks, phases);
  run<2>(stream, out, blocks, phases);
  run<3>(stream, out, blocks, phases);
  run<4>(stream, out, blocks, phases);
  run<5>(stream, out, blocks, phases);  // plain (cached) stores instead of nontemporal ones
  run<6>(stream, out, blocks, phases);  // plain stores, wait count that accounts for the 2 stores per phase
  return 0;
}
