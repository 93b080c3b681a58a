// Probe 2: what limits the register-chained f32 MFMA stream at ~80% of peak?
// Variants: number of independent accumulators per A fragment (frame tiles), with/without streamed weights, MFMA shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// NF = frame tiles per wave (independent accumulators per A fragment); LOADS = stream weights from memory
template <int KT, int NT, int NF, bool LOADS>
__global__ __launch_bounds__(256) void stream16(const f32x4* __restrict__ wp, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 in[KT][NF];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) in[kt][ft] = f32x4{0.37f * lane + kt, 1.1f, 0.3f * ft, 0.5f};
  f32x4 sum[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) sum[ft] = f32x4{0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    const f32x4* w = wp + lane;
    f32x4 frag[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) frag[kt] = w[kt * 64];
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc[ft] = f32x4{0, 0, 0, 0};
      const f32x4* wn = w + (size_t)((nt + 1 < NT ? nt + 1 : 0) * KT) * 64;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        f32x4 a = frag[kt];
        if (LOADS) { frag[kt] = wn[kt * 64]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], in[kt][ft][r], acc[ft], 0, 0, 0);
      }
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) sum[ft] += acc[ft];
    }
  }
  f32x4 s = sum[0];
#pragma unroll
  for (int ft = 1; ft < NF; ++ft) s += sum[ft];
  out[(blockIdx.x * blockDim.x + threadIdx.x)] = s[0] + s[1] + s[2] + s[3];
}

// 32x32x2 shape: one wave = 32 frames as ONE frame tile; weights fragment [32 n][k]: lane (n=l&31, h=l>>5): float4 = k 8j+4h..+3
template <int K8, int NT, int NF, bool LOADS>
__global__ __launch_bounds__(256) void stream32(const f32x4* __restrict__ wp, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 in[K8][NF];
#pragma unroll
  for (int k = 0; k < K8; ++k)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) in[k][ft] = f32x4{0.37f * lane + k, 1.1f, 0.3f * ft, 0.5f};
  f32x16 sum[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft)
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[ft][i] = 0.f;
  for (int rep = 0; rep < reps; ++rep) {
    const f32x4* w = wp + lane;
    f32x4 frag[K8];
#pragma unroll
    for (int k = 0; k < K8; ++k) frag[k] = w[k * 64];
    for (int nt = 0; nt < NT; ++nt) {
      f32x16 acc[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ft][i] = 0.f;
      const f32x4* wn = w + (size_t)((nt + 1 < NT ? nt + 1 : 0) * K8) * 64;
#pragma unroll
      for (int k = 0; k < K8; ++k) {
        f32x4 a = frag[k];
        if (LOADS) { frag[k] = wn[k * 64]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) acc[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], in[k][ft][r], acc[ft], 0, 0, 0);
      }
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) sum[ft] += acc[ft];
    }
  }
  float s = 0;
#pragma unroll
  for (int ft = 0; ft < NF; ++ft)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sum[ft][i];
  out[(blockIdx.x * blockDim.x + threadIdx.x)] = s;
}

// variant: refill the ring slot AFTER its MFMAs were issued (no copies), optional split-K into 2 chains per frame tile
template <int KT, int NT, int NF, bool SPLIT>
__global__ __launch_bounds__(256) void stream16b(const f32x4* __restrict__ wp, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 in[KT][NF];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) in[kt][ft] = f32x4{0.37f * lane + kt, 1.1f, 0.3f * ft, 0.5f};
  f32x4 sum[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) sum[ft] = f32x4{0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    const f32x4* w = wp + lane;
    f32x4 frag[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) frag[kt] = w[kt * 64];
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc[2][NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = f32x4{0, 0, 0, 0};
      const f32x4* wn = w + (size_t)((nt + 1 < NT ? nt + 1 : 0) * KT) * 64;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft)
            acc[SPLIT ? (kt & 1) : 0][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[kt][r], in[kt][ft][r], acc[SPLIT ? (kt & 1) : 0][ft], 0, 0, 0);
        frag[kt] = wn[kt * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) sum[ft] += acc[0][ft] + acc[1][ft];
    }
  }
  f32x4 s = sum[0];
#pragma unroll
  for (int ft = 1; ft < NF; ++ft) s += sum[ft];
  out[(blockIdx.x * blockDim.x + threadIdx.x)] = s[0] + s[1] + s[2] + s[3];
}

// variant: refill slot kt-LAG after the MFMAs of slot kt (write-after-read distance of LAG fragments)
template <int KT, int NT, int NF, int LAG>
__global__ __launch_bounds__(256) void stream16c(const f32x4* __restrict__ wp, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 in[KT][NF];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) in[kt][ft] = f32x4{0.37f * lane + kt, 1.1f, 0.3f * ft, 0.5f};
  f32x4 sum[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) sum[ft] = f32x4{0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    const f32x4* w = wp + lane;
    f32x4 frag[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) frag[kt] = w[kt * 64];
    const f32x4* wn = w;
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc[2][NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = f32x4{0, 0, 0, 0};
      // wn points at the phase whose fragments are in flight / resident; slot j of the NEXT phase is wn[(KT + j) * 64]
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft)
            acc[kt & 1][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[kt][r], in[kt][ft][r], acc[kt & 1][ft], 0, 0, 0);
        if (kt >= LAG) { frag[kt - LAG] = wn[(KT + kt - LAG) * 64]; __builtin_amdgcn_sched_barrier(0); }
      }
#pragma unroll
      for (int j = KT - LAG; j < KT; ++j) { frag[j] = wn[(KT + j) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
      wn += KT * 64;
      if (nt + 2 >= NT) wn = w - KT * 64 + 0;  // stay in bounds (timing probe only)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) sum[ft] += acc[0][ft] + acc[1][ft];
    }
  }
  f32x4 s = sum[0];
#pragma unroll
  for (int ft = 1; ft < NF; ++ft) s += sum[ft];
  out[(blockIdx.x * blockDim.x + threadIdx.x)] = s[0] + s[1] + s[2] + s[3];
}

// variant: fragment addresses = wave-uniform base (SGPR) + constant per-lane byte offset (no per-load VALU address math)
template <int KT, int NT, int NF>
__global__ __launch_bounds__(256) void stream16d(const f32x4* __restrict__ wp, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  const unsigned loff = lane * 16;
  f32x4 in[KT][NF];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) in[kt][ft] = f32x4{0.37f * lane + kt, 1.1f, 0.3f * ft, 0.5f};
  f32x4 sum[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) sum[ft] = f32x4{0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep) {
    const char* base = (const char*)wp;  // uniform
    f32x4 frag[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) frag[kt] = *(const f32x4*)(base + kt * 1024 + loff);
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc[2][NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = f32x4{0, 0, 0, 0};
      const char* nb = base + (nt + 1 < NT ? KT * 1024 : 0);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft)
            acc[kt & 1][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[kt][r], in[kt][ft][r], acc[kt & 1][ft], 0, 0, 0);
        frag[kt] = *(const f32x4*)(nb + kt * 1024 + loff);
        __builtin_amdgcn_sched_barrier(0);
      }
      base = nb;
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) sum[ft] += acc[0][ft] + acc[1][ft];
    }
  }
  f32x4 s = sum[0];
#pragma unroll
  for (int ft = 1; ft < NF; ++ft) s += sum[ft];
  out[(blockIdx.x * blockDim.x + threadIdx.x)] = s[0] + s[1] + s[2] + s[3];
}

template <class F> void timeit(const char* name, F launch, double flops) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(1); CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0)); launch(12); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  printf("%-46s %.3f ms  %.1f TFLOP/s\n", name, best, flops * 12 / best * 1e-9);
}

int main() {
  constexpr int KT = 10, NT = 120;
  size_t wfloats = (size_t)NT * 20 * 64 * 4;
  std::vector<float> W(wfloats); for (size_t i = 0; i < wfloats; ++i) W[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  f32x4* dW; CK(hipMalloc(&dW, wfloats * 4)); CK(hipMemcpy(dW, W.data(), wfloats * 4, hipMemcpyHostToDevice));
  float* dO; CK(hipMalloc(&dO, 4096 * 256 * 4));
  const int blocks = 256;  // 1 wave per SIMD
  double f16 = (double)blocks * 4 * NT * KT * 4 * 2048.0;  // per rep per frame tile
  timeit("16x16x4 NF=2 loads (current design)", [&](int r) { stream16<KT, NT, 2, true><<<blocks, 256>>>(dW, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=2 loads refill-after-use", [&](int r) { stream16b<KT, NT, 2, false><<<blocks, 256>>>(dW, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=2 loads refill-after-use + splitK", [&](int r) { stream16b<KT, NT, 2, true><<<blocks, 256>>>(dW, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=2 splitK refill LAG=1", [&](int r) { stream16c<KT, NT, 2, 1><<<blocks, 256>>>(dW + 640, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=2 splitK refill LAG=2", [&](int r) { stream16c<KT, NT, 2, 2><<<blocks, 256>>>(dW + 640, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=2 splitK refill LAG=0", [&](int r) { stream16c<KT, NT, 2, 0><<<blocks, 256>>>(dW + 640, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=2 splitK saddr-base loads", [&](int r) { stream16d<KT, NT, 2><<<blocks, 256>>>(dW, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=4 loads refill-after-use", [&](int r) { stream16b<KT, NT, 4, false><<<blocks, 256>>>(dW, dO, r); }, f16 * 4);
  timeit("16x16x4 NF=2 no loads", [&](int r) { stream16<KT, NT, 2, false><<<blocks, 256>>>(dW, dO, r); }, f16 * 2);
  timeit("16x16x4 NF=4 loads", [&](int r) { stream16<KT, NT, 4, true><<<blocks, 256>>>(dW, dO, r); }, f16 * 4);
  timeit("16x16x4 NF=4 no loads", [&](int r) { stream16<KT, NT, 4, false><<<blocks, 256>>>(dW, dO, r); }, f16 * 4);
  timeit("16x16x4 NF=3 loads", [&](int r) { stream16<KT, NT, 3, true><<<blocks, 256>>>(dW, dO, r); }, f16 * 3);
  double f32_ = (double)blocks * 4 * NT * 20 * 4 * 4096.0;  // K8 = 20 (K=160), per frame tile of 32
  timeit("32x32x2 NF=1 loads", [&](int r) { stream32<20, NT, 1, true><<<blocks, 256>>>(dW, dO, r); }, f32_);
  timeit("32x32x2 NF=1 no loads", [&](int r) { stream32<20, NT, 1, false><<<blocks, 256>>>(dW, dO, r); }, f32_);
  timeit("32x32x2 NF=2 loads", [&](int r) { stream32<20, NT, 2, true><<<blocks, 256>>>(dW, dO, r); }, f32_ * 2);
  timeit("32x32x2 NF=2 no loads", [&](int r) { stream32<20, NT, 2, false><<<blocks, 256>>>(dW, dO, r); }, f32_ * 2);
  // 2 waves per SIMD (512 blocks) for the current design
  timeit("16x16x4 NF=2 loads, 512 blocks (2 waves/SIMD)", [&](int r) { stream16<KT, NT, 2, true><<<512, 256>>>(dW, dO, r); }, f16 * 2 * 2);
  return 0;
}
