// Probe 3: what does ONE wave per SIMD overlap with a v_mfma_f32_16x16x4_f32 stream?
// Body (exact order, inline asm): 4 independent MFMA chains; after every MFMA, NV instructions of one kind:
//   fma  : v_fma_f32 on VGPRs (8 independent chains)          exp  : v_exp_f32 (quarter rate)
//   acc  : v_accvgpr_read + v_accvgpr_write on spare AGPRs     pk   : v_pk_mul_f32
//   accC : v_accvgpr_read of the accumulator the MFMA issued 3 slots ago wrote (true dependency on an MFMA result)
// Prints cycles per MFMA (wall clock * 2.4 GHz) -- 32 means the matrix pipe never waits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

#define REP0(X)
#define REP1(X) X
#define REP2(X) X X
#define REP3(X) X X X
#define REP4(X) X X X X
#define REP5(X) X X X X X
#define REP6(X) X X X X X X
#define REP8(X) X X X X X X X X

#define MF(i) "v_mfma_f32_16x16x4_f32 %" #i ", %8, %9, %" #i "\n"
#define FMA "v_fma_f32 %4, %4, %10, %10\n v_fma_f32 %5, %5, %10, %10\n"   /* 2 instr */
#define EXPI "v_exp_f32 %4, %4\n"
#define PKM "v_pk_mul_f32 %6, %6, %6\n"
#define ACC "v_accvgpr_read_b32 %4, %7\n v_accvgpr_write_b32 %7, %5\n"  /* 2 instr, %10 spare agpr */

template <int KIND, int NV>
__global__ __launch_bounds__(256) void probe(float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float wa = 0.001f * lane, wb = 0.5f;
  float x0 = 1.0f + lane, x1 = 0.5f, k = 0.999f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk = {1.0f, 1.0f};
  float sp = 0.f;
  for (int r = 0; r < reps; ++r) {
#define BODY(V)                                                                                                     \
  asm volatile(MF(0) V MF(1) V MF(2) V MF(3) V MF(0) V MF(1) V MF(2) V MF(3) V MF(0) V MF(1) V MF(2) V MF(3) V MF(0) V MF(1) V MF(2) V MF(3) V \
               : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+v"(x0), "+v"(x1), "+v"(pk), "+a"(sp)                     \
               : "v"(wa), "v"(wb), "v"(k));
    if (KIND == 0) {  // fma pairs: NV = number of PAIRS
      if (NV == 0) { BODY(REP0(FMA)) } else if (NV == 1) { BODY(REP1(FMA)) } else if (NV == 2) { BODY(REP2(FMA)) }
      else if (NV == 3) { BODY(REP3(FMA)) } else if (NV == 4) { BODY(REP4(FMA)) } else if (NV == 5) { BODY(REP5(FMA)) }
      else if (NV == 6) { BODY(REP6(FMA)) } else { BODY(REP8(FMA)) }
    } else if (KIND == 1) {
      if (NV == 1) { BODY(REP1(EXPI)) } else if (NV == 2) { BODY(REP2(EXPI)) } else if (NV == 3) { BODY(REP3(EXPI)) } else { BODY(REP4(EXPI)) }
    } else if (KIND == 2) {
      if (NV == 1) { BODY(REP1(ACC)) } else if (NV == 2) { BODY(REP2(ACC)) } else if (NV == 3) { BODY(REP3(ACC)) } else { BODY(REP4(ACC)) }
    } else if (KIND == 3) {
      if (NV == 1) { BODY(REP1(PKM)) } else if (NV == 2) { BODY(REP2(PKM)) } else if (NV == 4) { BODY(REP4(PKM)) } else { BODY(REP6(PKM)) }
    }
  }
  f32x4 s = a0 + a1 + a2 + a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + x0 + x1 + pk[0] + sp;
}

// true dependency: a VALU read (via accvgpr_read) of the accumulator written D MFMAs ago, then written back
#define MFD(i) "v_mfma_f32_16x16x4_f32 %" #i ", %4, %5, %" #i "\n"
template <int MODE>
__global__ __launch_bounds__(256) void probe_dep(float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float wa = 0.001f * lane, wb = 0.5f, t = 0.f;
  for (int r = 0; r < reps; ++r) {
    // 16 MFMAs on 4 chains, then (MODE 1) read one accumulator register right away, (MODE 2) after 4 unrelated MFMAs
    asm volatile(MFD(0) MFD(1) MFD(2) MFD(3) MFD(0) MFD(1) MFD(2) MFD(3) MFD(0) MFD(1) MFD(2) MFD(3) MFD(0) MFD(1) MFD(2) MFD(3)
                 : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(wa), "v"(wb));
    if (MODE == 1) { t += a3[0]; }
    if (MODE == 2) { t += a0[0]; }
    if (MODE == 3) { a3[0] *= 0.5f; }
  }
  f32x4 s = a0 + a1 + a2 + a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + t;
}


// Cost model for the other instruction kinds, per group of 16 MFMAs (512 cycles of matrix pipe):
#define MFX(i) "v_mfma_f32_16x16x4_f32 %" #i ", %7, %8, %" #i "\n"
#define M4 MFX(0) MFX(1) MFX(2) MFX(3)
#define LD "global_load_dwordx4 %4, %9, off\n"
#define VA "v_fma_f32 %5, %5, %10, %10\n"
#define SA "s_add_u32 %6, %6, 1\n"
template <int MODE>
__global__ __launch_bounds__(256) void probe_mix(const f32x4* __restrict__ src, float* out, int reps) {
  const int lane = threadIdx.x & 63;
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, ld = a0;
  float wa = 0.001f * lane, wb = 0.5f, x = 1.f, k = 0.999f;
  const f32x4* ptr = src + lane;
  int sreg = 0;
#define MIX(BODY) asm volatile(BODY : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3), "+v"(ld), "+v"(x), "+s"(sreg) : "v"(wa), "v"(wb), "v"(ptr), "v"(k) : "memory", "scc");
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) MIX(M4 M4 M4 M4)
    if (MODE == 1) MIX(M4 M4 LD M4 M4 LD)                      // 1 load per 8 MFMAs
    if (MODE == 2) MIX(M4 M4 M4 M4 LD LD)                      // 2 loads per 16, together
    if (MODE == 3) MIX(M4 M4 "s_waitcnt vmcnt(2)\n" LD M4 M4 "s_waitcnt vmcnt(2)\n" LD)  // the ring's cadence
    if (MODE == 4) MIX(M4 M4 M4 M4 "s_waitcnt vmcnt(2)\n" LD LD)
    if (MODE == 5) MIX(M4 SA M4 SA M4 SA M4 SA)                // 4 SALU spread
    if (MODE == 6) MIX(M4 M4 M4 M4 SA SA SA SA)                // 4 SALU together
    if (MODE == 7) MIX(M4 M4 M4 M4 VA VA VA VA VA VA VA VA)    // 8 VALU together
    if (MODE == 8) MIX(MFX(0) MFX(1) VA MFX(2) MFX(3) VA MFX(0) MFX(1) VA MFX(2) MFX(3) VA MFX(0) MFX(1) VA MFX(2) MFX(3) VA MFX(0) MFX(1) VA MFX(2) MFX(3) VA)  // 8 VALU spread
    if (MODE == 9) MIX(M4 M4 "s_waitcnt vmcnt(0)\n" M4 M4 "s_waitcnt vmcnt(0)\n")   // satisfied waitcnt only
    if (MODE == 10) MIX(M4 M4 "s_nop 0\n" M4 M4 "s_nop 0\n")
    if (MODE == 11) MIX(M4 M4 M4 M4 LD LD LD LD M4 M4 M4 M4)   // 4 loads per 32 (counts 32 MFMAs)
  }
  asm volatile("s_waitcnt vmcnt(0)");
  f32x4 s = a0 + a1 + a2 + a3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + x + ld[0];
}

template <class F>
double time_ms(F&& launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  float* out; CK(hipMalloc(&out, 256 * 256 * 4 * sizeof(float)));
  const int reps = 5000;
  const double ghz = 2.4;
#define RUN(K, N, label, extra)                                                                          \
  { double ms = time_ms([&] { hipLaunchKernelGGL((probe<K, N>), dim3(256), dim3(256), 0, 0, out, reps); }); \
    printf("%-5s x%d/MFMA (%2d instr): %6.1f cycles per MFMA\n", label, N, N * extra, ms * 1e-3 * ghz * 1e9 / (reps * 16.0)); }
  RUN(0, 0, "none", 0)
  RUN(0, 1, "fma", 2) RUN(0, 2, "fma", 2) RUN(0, 3, "fma", 2) RUN(0, 4, "fma", 2) RUN(0, 5, "fma", 2) RUN(0, 6, "fma", 2) RUN(0, 8, "fma", 2)
  RUN(1, 1, "exp", 1) RUN(1, 2, "exp", 1) RUN(1, 3, "exp", 1) RUN(1, 4, "exp", 1)
  RUN(2, 1, "acc", 2) RUN(2, 2, "acc", 2) RUN(2, 3, "acc", 2) RUN(2, 4, "acc", 2)
  RUN(3, 1, "pk", 1) RUN(3, 2, "pk", 1) RUN(3, 4, "pk", 1) RUN(3, 6, "pk", 1)
#define RUND(M, label)                                                                                    \
  { double ms = time_ms([&] { hipLaunchKernelGGL((probe_dep<M>), dim3(256), dim3(256), 0, 0, out, reps); }); \
    printf("dep %-28s: %6.1f cycles per 16-MFMA group (512 = no stall)\n", label, ms * 1e-3 * ghz * 1e9 / reps); }
  RUND(0, "none") RUND(1, "read last acc") RUND(2, "read acc of 4 MFMAs ago") RUND(3, "rmw last acc")
  f32x4* src; CK(hipMalloc(&src, 64 * sizeof(f32x4))); CK(hipMemset(src, 0, 64 * sizeof(f32x4)));
#define RUNM(M, label, nm)                                                                                   \
  { double ms = time_ms([&] { hipLaunchKernelGGL((probe_mix<M>), dim3(256), dim3(256), 0, 0, src, out, reps); }); \
    printf("mix %-44s: %6.1f extra cycles per 16 MFMAs\n", label, ms * 1e-3 * ghz * 1e9 / reps * 16.0 / nm - 16 * 32.0); }
  RUNM(0, "none", 16) RUNM(1, "1 load / 8 MFMAs", 16) RUNM(2, "2 loads / 16 MFMAs", 16) RUNM(3, "(waitcnt + load) / 8 MFMAs", 16)
  RUNM(4, "(waitcnt + 2 loads) / 16 MFMAs", 16) RUNM(5, "4 SALU spread", 16) RUNM(6, "4 SALU together", 16)
  RUNM(7, "8 VALU together", 16) RUNM(8, "8 VALU spread (1 per 2 MFMAs)", 16) RUNM(9, "2 satisfied s_waitcnt", 16)
  RUNM(10, "2 s_nop 0", 16) RUNM(11, "4 loads / 32 MFMAs", 32)
  return 0;
}
