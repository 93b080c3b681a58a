// edtts_device.h -- device-side building blocks for the gfx950 (MI355X) sampler kernels.
//
// Data model ("frame on the lane"):
//   One wavefront (64 lanes) owns 32 consecutive mel frames of one utterance = two 16-frame tiles (ft = 0, 1).
//   A [32 x N] activation is held TRANSPOSED in the C/D layout of v_mfma_f32_16x16x4_f32:
//       reg[nt][ft] (a float4), lane = 16*g + fq  holds  act[frame = 16*ft + fq][feature = 16*nt + 4*g + r], r = 0..3
//   i.e. frame on (lane & 15), the four lane groups g = lane >> 4 hold different feature quads.
//   With out^T = W x^T  (A operand = weights [n][k], B operand = activations [k][frame]) the C/D layout of one
//   GEMM is *exactly* the B-operand layout of the next one when the k index is walked in the order
//   k = 16*kt + 4*g + r  (MFMA step (kt, r) contracts the four k's {16kt+4g+r : g=0..3}).  Chains of per-frame
//   linear layers, norms, activations and residuals therefore never leave registers, and the attention
//   probabilities (C/D of K Q^T) feed the P V product the same way.
//   Weights are pre-packed (edtts_pack_weights) so that the A fragment of step group (nt, kt) is one float4
//   per lane, 1 KiB contiguous per wave instruction:  frag[lane] = W[16nt + fq][16kt + 4g + 0..3].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Every EDTTS_* / EDTTS16_* macro that is not part of the public header is a tuning, ablation or diagnostic switch of scratch/:
// some change results ("wrong by construction"), some select measured-and-rejected variants, some only move a default.  None may
// reach a product build by accident: setting ANY of them from the command line needs -DEDTTS_EXPERIMENTS next to it.  (This test
// sits above every `#ifndef X / #define X default`, so `defined(X)` here means "set from outside".)
#if !defined(EDTTS_EXPERIMENTS) && (                                                                                               \
    defined(EDTTS_ABLATE_QKVSTORES) || defined(EDTTS_ABLATE_KVLOADS) || defined(EDTTS_DIAG) || defined(EDTTS_KV2) ||                \
    defined(EDTTS_PERSIST) || defined(EDTTS_H_DMA) || defined(EDTTS_SPLITLOAD) || defined(EDTTS_STAMPS) || defined(EDTTS_DS_ABL) || \
    defined(EDTTS_FAST_BUILD) || defined(EDTTS_W2) || defined(EDTTS_WAVELOG) || defined(EDTTS_PIN_MASK) || defined(EDTTS_HANDOVER) || defined(EDTTS_DS_BURST) || defined(EDTTS16_MFMA_SUM) || defined(EDTTS_DS_STAMPS) || defined(EDTTS_TAIL_RING2) || defined(EDTTS_RB) || defined(EDTTS_WMAX) ||        \
    defined(EDTTS_NF_DEFAULT) || defined(EDTTS_NF_FFN) || defined(EDTTS_STAMP_THREAD) || defined(EDTTS_STAMP_HEAD) ||               \
    defined(EDTTS16_ABLATE_BARRIER) || defined(EDTTS16_ABLATE_DMA) || defined(EDTTS16_SPLIT_BUILD) || defined(EDTTS16_PHASES) ||    \
    defined(EDTTS16_NF) || defined(EDTTS16_CTX_F32) || defined(EDTTS16_WIDE_KD_SELF) || defined(EDTTS16_WIDE_KD) ||                 \
    defined(EDTTS16_W1) || defined(EDTTS16_QLDS) || defined(EDTTS16_KVDEPTH) || defined(EDTTS16_IMG) || defined(EDTTS16_HP) ||      \
    defined(EDTTS16_FRAG_GROUP) || defined(EDTTS16_EVEN_STEPS) || defined(EDTTS16_ATT_OCC) || defined(EDTTS16_WIDE_BUILD))
#error "tuning / ablation / diagnostic / measured-and-rejected variant switches are scratch-only: add -DEDTTS_EXPERIMENTS"
#endif

#ifndef EDTTS_PIN_MASK
// Instruction types that may cross the scheduling pin IN FRONT of a K / V^T re-request inside an MFMA run (the pin behind it stays
// total).  0x4 = SALU: the request's address arithmetic (7-12 scalar instructions per burst) then sits in the shadow of the run's
// first MFMAs instead of in a clump between two MFMA groups whose excess over one MFMA's 32 cycles is exposed.  Round 4, same
// device, three interleaved runs: 0.9084 vs 0.9140 ms per layer launch (0: round 3's total pin).  Results bitwise unchanged.
#define EDTTS_PIN_MASK 0x4
#endif
#ifndef EDTTS_W2
#define EDTTS_W2 0  // measured and rejected (round 4): the default decoder's 32-frame instance at TWO waves per SIMD (<= 256 registers: attention outputs wait in LDS for one projection over all heads, residual through global memory, no AGPR-class pins)
#endif
#if EDTTS_W2
#define EDTTS_PIN_ACC(x) asm volatile("" : "+v"(x))
#else
#define EDTTS_PIN_ACC(x) asm volatile("" : "+a"(x))
#endif
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define EDTTS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define EDTTS_DEV __device__ __forceinline__

// Diagnostic builds (-DEDTTS_EXPERIMENTS -DEDTTS_STAMPS; scratch/stamps_f32.py, stamps_bf16.py): s_memtime stamps of ONE wave.
// Which one is chosen at run time (environment of the stamps build: EDTTS_STAMP_BLOCK / EDTTS_STAMP_WAVE / EDTTS_STAMP_HEAD ->
// KArgs::diag_skip = head | wave << 8 | block << 16; defaults: block 8 = logical block 1 after the XCD remap, wave 0, head 1).
#ifdef EDTTS_STAMPS
#define STAMP_SEL_HEAD(sel) ((sel) & 0xff)
#define STAMPX(p, i, sel) do { if ((p) && (int)blockIdx.x == ((sel) >> 16) && (int)threadIdx.x == 64 * (((sel) >> 8) & 0xff)) (p)[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMPX(p, i, sel) do { } while (0)
#endif

namespace edtts {

constexpr int kCtxWaves = 4;     // waves per block of the context kernel (no block-level sharing there)

// NF = 16-frame tiles per wave.  Every weight fragment (1 KiB) feeds 4*NF MFMAs, and the bare register-chained stream
// measures 141 TFLOP/s at NF = 2 against 150 at NF = 4 (scratch/mfma_probe2.cpp), so the default decoder runs NF = 4
// (64 frames per wave, the residual tile alone is 160 registers); attention works on pairs of frame tiles at a time.
template <int H_, int HEADS_, int MEL_, int NF_ = 2>
struct Cfg {
  static constexpr int H = H_, HEADS = HEADS_, MEL = MEL_, NF = NF_;
  static constexpr int WF = 16 * NF;              // frames per wave
  static constexpr int QT = NF >= 2 ? 2 : 1;      // query tiles per attention pass
  static constexpr int NHALF = NF / QT;           // attention passes per wave (QT frame tiles each)
  static constexpr int DH = H / HEADS;            // head dim (40 for the default decoder)
  static constexpr int DFULL = DH / 16;           // full 16-wide groups of the head dim
  static constexpr int DREM = DH % 16;            // remainder (0 or 8 supported)
  static constexpr int DT = (DH + 15) / 16;       // 16-row tiles of the P.V output per head
  static constexpr int DHP = DT * 16;             // head dim padded to the MFMA tile
  static constexpr int HT = H / 16;               // feature tiles of the hidden dim
  static constexpr int MT = MEL / 16;             // feature tiles of the mel dim
  static constexpr int R = H / 2, RT = R / 16;    // kv_lora_rank (transformer.py:113) and its tiles
  static constexpr int VR = (HEADS - 1) * DH + DHP;  // rows of a V^T buffer (last head padded)
  // Waves per block.  The waves of a block never synchronise, so the block size only sets how many waves must FINISH before the
  // CU's freed SIMDs get new work: two-wave blocks (two resident per CU at H = 160) measured 0.6 % faster than four-wave ones
  // (B=256, T=512: 0.9455 vs 0.9517 ms per layer launch), one-wave blocks the same as two.  LDS: each wave parks HT*NF KiB.
#ifndef EDTTS_WMAX
#define EDTTS_WMAX 2
#endif
  static constexpr int WAVES0 = (HT * NF * 4 <= 160) ? 4 : 2;
  // (The 16-frames-per-wave instances run ONE-wave blocks: they fit 256 registers, and with two-wave blocks the dispatcher put the
  // two waves of a CU's second block onto the same two SIMDs as the first -- B=32, T=512: 0.225 ms per layer launch against 0.154
  // with four-wave or one-wave blocks; one-wave blocks also spread a tiny grid over the most CUs: B=1, T=256 at 0.107 ms per
  // layer launch against 0.125 with four-wave blocks.  The 32-frame instances need > 256 registers: one wave per SIMD is all
  // that fits, whatever the block size.)
  static constexpr int WAVES = NF == 1 ? 1 : (WAVES0 < EDTTS_WMAX ? WAVES0 : EDTTS_WMAX);
  // the cross-attention q tile goes through LDS when it fits next to the parked residual tiles (160 KiB per block at H = 160,
  // NF = 2), else through this wave's (already consumed) self-attention q rows in global memory
  static constexpr bool Q_IN_LDS = !(EDTTS_W2 && NF == 2) && WF * H * 4 * 2 <= 40 * 1024;  // per wave: a quarter of the CU's 160 KiB (four waves per CU, in 1, 2 or 4 blocks)
  static constexpr int THREADS = 64 * WAVES;
  // DEFER: the per-head attention outputs O^T wait in LDS (B-operand layout) and ONE projection over all heads follows the last
  // head, so the 16*NF*HT-register branch tile is not live during the attention steps (see attention_fused)
  static constexpr bool DEFER = EDTTS_W2 && NF == 2 && H == 160;
  static constexpr int OHEAD_BYTES = DFULL * NF * 1024 + (DREM ? NF * 512 : 0);  // LDS bytes of one head's O^T tiles (remainder rows as f2)
  static_assert(NF == 1 || NF == 2 || NF == 4, "frame tiles per wave");
  static_assert(H % 32 == 0 && MEL % 16 == 0 && H % HEADS == 0, "dims");
  static_assert(DREM == 0 || DREM == 8, "head_dim % 16 must be 0 or 8");
};

EDTTS_DEV f4 splat(float v) { return f4{v, v, v, v}; }
EDTTS_DEV f4 ldg4(const float* p) { return *reinterpret_cast<const f4*>(p); }
EDTTS_DEV f2 ldg2(const float* p) { return *reinterpret_cast<const f2*>(p); }
EDTTS_DEV void stg4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
// Loads from a wave-uniform base plus a per-lane 32-bit byte offset: the uniform part is SALU arithmetic, the lane part never
// changes, and hipcc keeps one hoisted per-lane pointer per stream instead of rebuilding 64-bit addresses for every load.
EDTTS_DEV f4 ldg4_sbase(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(base) + byte_off);
}
EDTTS_DEV f2 ldg2_sbase(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const f2*>(reinterpret_cast<const char*>(base) + byte_off);
}

// LDS-DMA of a register-layout tile: for every (nt, ft) one global_load_lds_dwordx4 moves this lane's float4 at
// src[16 nt + ft * 16 * ld] to lds_wave[(nt * NF + ft) * 64 + lane] (64 lanes x 16 B land lane-contiguous) without passing through
// registers.  lds_wave is wave-uniform; src is this lane's pointer (row fq, feature quad g).
// Written as inline asm ON PURPOSE: with the builtin, hipcc's waitcnt pass cannot count an outstanding LDS-DMA and turns every
// later wait of the attention loop into s_waitcnt vmcnt(0) (the K / V^T requests of the next step are then waited for at once).
// Hidden from the pass, the DMAs are just OLDER entries of the in-order vmcnt queue: every counted wait the compiler emits for
// a later load also covers them (conservatively), so by the first use of ANY load issued after this call the tile has landed --
// the caller reads it many thousands of cycles later.  (s_nop: one wait state between the SALU write of M0 and the DMA.)
template <int NT, int NF>
EDTTS_DEV void dma_tile_to_lds(const float* src, int ld, f4* lds_wave) {
  typedef __attribute__((address_space(3))) char* lds_char_t;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_t)(char*)lds_wave);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) {
      const float* p = src + 16 * nt + (size_t)ft * 16 * ld;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(base + (unsigned)(nt * NF + ft) * 1024u) : "m0", "memory");
    }
}

// Buffer loads: address = descriptor base (4 SGPRs) + wave-uniform byte offset in an SGPR (soffset) + per-lane byte offset in ONE
// VGPR that never changes (+ a 12-bit immediate).  All the per-load address arithmetic of a stream is then SALU work -- the
// global_load forms hipcc picks for "uniform base + lane offset" rebuild a 64-bit lane address with a VALU instruction per load
// (v_lshl_add_u64: 12 per attention step, ~14 per FFN iteration in the round-2 kernel; VALU time is MFMA time on fp32).
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
EDTTS_DEV __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  // raw buffer, stride 0, no range limit (the offsets are the kernel's own tile arithmetic), gfx9 word 3 = DATA_FORMAT 32-bit
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFF, 0x00020000);
}
EDTTS_DEV f4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
EDTTS_DEV f2 bufld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// Correctly rounded fp32 sqrt / divide via fp64 (53 >= 2*24+2 bits, so rounding the fp64 result to fp32 is the
// IEEE fp32 result) -- the device's fp32 sqrt is not guaranteed correctly rounded, the reference's CPU one is.
EDTTS_DEV float sqrt_rn(float x) { return (float)sqrt((double)x); }
EDTTS_DEV float div_rn(float a, float b) { return (float)((double)a / (double)b); }

// reduce a per-lane value over the four lane groups that hold the same frame (lanes fq, fq+16, fq+32, fq+48).
// v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / 32-lane halves in the VALU (no LDS crossbar trip as
// ds_bpermute would take): after swap(x, x) the two results hold {own, partner} in some order, and the ops are
// commutative.
EDTTS_DEV float group_sum(float v) {
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
EDTTS_DEV float group_max(float v) {
  unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
EDTTS_DEV float hsum(f4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
EDTTS_DEV float hmax(f4 v) { return fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])); }

// ---------------------------------------------------------------------------------------------------------
// Weight fragment stream.  The packed weights of a kernel are laid out in CONSUMPTION order as one linear stream of
// 1-KiB fragments.  FragRing<RN> keeps the next RN fragments in registers: position i of the current phase lives in
// slot i % RN, and once the MFMAs that read a slot have been ISSUED the slot is refilled with stream element i + RN
// (prefetch distance RN fragments = 4*NF*RN MFMAs; RN must divide every phase length).  Refilling after use matters:
// loading into the slot before its old value is consumed makes hipcc copy the whole ring and wait for every load at
// the top of each phase (measured 126 -> 137 TFLOP/s on the bare stream).  The sched_barrier pins the issue point:
// without it the pre-RA scheduler sinks each load to its use RN fragments later, i.e. load -> s_waitcnt -> MFMA.
// ---------------------------------------------------------------------------------------------------------
// Refill burst: round 2 used 4 (an interruption of the MFMA run by VMEM issue cost ~20 cycles + 3 per load with the global_load
// forms); with buffer loads (no address VALU) smaller bursts win because every slot is re-requested sooner: B=256, T=512 layer launch
// 0.9280 ms at 4, 0.9186 at 2, 0.9184 at 1 (same device, interleaved runs).
#ifndef EDTTS_RB
#define EDTTS_RB 2
#endif
constexpr int kRefillBurst = EDTTS_RB;
template <int RN>
struct FragRing {
  static constexpr int RN_ = RN;
  __amdgpu_buffer_rsrc_t rs;  // the kernel's fragment stream
  unsigned soff;              // byte offset of position 0 of the current phase (wave-uniform: SALU arithmetic)
  unsigned voff;              // lane * 16
  f4 r[RN];
  EDTTS_DEV f4 frag(int i) const {
    return bufld4(rs, voff, soff + (unsigned)i * 1024u);
  }
  EDTTS_DEV void prime(const float* base, int lane) {
    rs = make_rsrc(base);
    soff = 0;
    voff = (unsigned)lane * 16u;
#pragma unroll
    for (int i = 0; i < RN; ++i) r[i] = frag(i);
  }
  EDTTS_DEV const f4& at(int i) const { return r[i % RN]; }
  // Refill the slots of positions [lo, hi) (all consumed) in ONE burst.  Every interruption of the MFMA stream by VMEM
  // issue costs ~20 cycles plus ~3 per load (scratch/mfma_probe3.cpp), so the loads go out kRefillBurst at a time.
  EDTTS_DEV void refill(int lo, int hi) {
#pragma unroll
    for (int i = lo; i < hi; ++i) r[i % RN] = frag(i + RN);
    __builtin_amdgcn_sched_barrier(0);
  }
  // called after position i of an N-fragment phase has been consumed
  template <int N>
  EDTTS_DEV void refill_after(int i) {
    constexpr int B = kRefillBurst < RN ? kRefillBurst : RN;
    if ((i + 1) % B == 0) refill(i + 1 - B, i + 1);
    else if (i == N - 1) refill(N - N % B, N);
  }
  EDTTS_DEV void advance(int n) { soff += (unsigned)n * 1024u; }
};

// out^T tile (16 features x 16*NF frames) += sum_kt frag(kt) * in[kt]: one n-major phase of KT fragments.  At NF = 2 even /
// odd k-tiles go to separate accumulators and are summed at the end, so that 4 independent MFMA chains are in flight
// (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency); at NF = 4 the frame tiles are the 4 chains.
template <int KT, int RN, int NF>
EDTTS_DEV void gemm_phase(FragRing<RN>& ring, const f4 (&in)[KT][NF], f4 (&acc)[NF]) {
  static_assert(KT % RN == 0, "phase length must be a multiple of the ring size");
  constexpr bool SPLIT = NF < 4;
  f4 b[NF];
  if (SPLIT) {
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) b[ft] = splat(0.f);
  }
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const f4& a = ring.at(kt);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        if (SPLIT && (kt & 1)) b[ft] = EDTTS_MFMA(a[r], in[kt][ft][r], b[ft]);
        else acc[ft] = EDTTS_MFMA(a[r], in[kt][ft][r], acc[ft]);
      }
    ring.template refill_after<KT>(kt);
  }
  if (SPLIT) {
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) acc[ft] += b[ft];
  }
  ring.advance(KT);
}

// Two n-tiles at once from a stream that interleaves their fragments per k-tile ([kt][tile a | tile b]).
// Used for the FFN value/gate pair (layers/transformer.py:21-23).
template <int KT, int RN, int NF>
EDTTS_DEV void gemm_phase_pair(FragRing<RN>& ring, const f4 (&in)[KT][NF], f4 (&a)[NF], f4 (&b)[NF]) {
  static_assert((2 * KT) % RN == 0, "phase length must be a multiple of the ring size");
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const f4& fa = ring.at(2 * kt);
    const f4& fb = ring.at(2 * kt + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) a[ft] = EDTTS_MFMA(fa[r], in[kt][ft][r], a[ft]);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) b[ft] = EDTTS_MFMA(fb[r], in[kt][ft][r], b[ft]);
    }
    ring.template refill_after<2 * KT>(2 * kt);
    ring.template refill_after<2 * KT>(2 * kt + 1);
  }
  ring.advance(2 * KT);
}

// acc[nt] += frag(nt) * in   for one k-tile of a k-major packed matrix (NT fragments).  At NF = 2 two n-tiles are
// interleaved so that four accumulator chains are in flight.
// NR < 4 issues only the first NR of the k-tile's MFMA steps (the remaining weights of the fragments are zero by packing).
template <int NT, int RN, int NF, int NR = 4>
EDTTS_DEV void ktile_phase(FragRing<RN>& ring, const f4 (&in)[NF], f4 (&acc)[NT][NF]) {
  static_assert(NT % RN == 0, "phase length must be a multiple of the ring size");
  constexpr int STEP = NF < 4 ? 2 : 1;
#pragma unroll
  for (int nt = 0; nt < NT; nt += STEP) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int u = 0; u < STEP; ++u)
        if (nt + u < NT) {
          const f4& fa = ring.at(nt + u);
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) acc[nt + u][ft] = EDTTS_MFMA(fa[r], in[ft][r], acc[nt + u][ft]);
        }
#pragma unroll
    for (int u = 0; u < STEP; ++u)
      if (nt + u < NT) ring.template refill_after<NT>(nt + u);
  }
  ring.advance(NT);
}

// ---------------------------------------------------------------------------------------------------------
// Norms in the register layout.  x[t][ft] covers features 16t+4g+r of frame (ft, fq).
// ---------------------------------------------------------------------------------------------------------
// RMSNorm (layers/mla.py:46-58): x * rsqrt(mean(x^2) + 1e-6) * w ; optional AdaLN modulation
// (layers/transformer.py:64-68): y * (1 + scale) + shift, with mod = {1+scale [H], shift [H]} rows.
template <int NT, int NF>
EDTTS_DEV void rms_norm_tile(const f4 (&x)[NT][NF], const float* __restrict__ w, const float* __restrict__ mod, int g,
                             f4 (&y)[NT][NF]) {
  constexpr int N = NT * 16;
  float rs[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) ss += hsum(x[t][ft] * x[t][ft]);
    rs[ft] = rsqrtf(group_sum(ss) * (1.0f / N) + 1e-6f);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f4 wv = ldg4(w + 16 * t + 4 * g);
    f4 sc = splat(1.f), sh = splat(0.f);
    if (mod != nullptr) {
      sc = ldg4(mod + 16 * t + 4 * g);
      sh = ldg4(mod + N + 16 * t + 4 * g);
    }
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) {
      f4 a = x[t][ft] * rs[ft] * wv;
      if (mod != nullptr) a = a * sc + sh;
      y[t][ft] = a;
    }
  }
}

// LayerNorm(eps 1e-5, affine) (models/decoder.py:59,108)
template <int NT, int NF>
EDTTS_DEV void layer_norm_tile(const f4 (&x)[NT][NF], const float* __restrict__ w, const float* __restrict__ b, int g,
                               f4 (&y)[NT][NF]) {
  constexpr int N = NT * 16;
  float mu[NF], rs[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += hsum(x[t][ft]);
    mu[ft] = group_sum(s) * (1.0f / N);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f4 d = x[t][ft] - mu[ft];
      v += hsum(d * d);
    }
    rs[ft] = rsqrtf(group_sum(v) * (1.0f / N) + 1e-5f);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f4 wv = ldg4(w + 16 * t + 4 * g), bv = ldg4(b + 16 * t + 4 * g);
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) y[t][ft] = (x[t][ft] - mu[ft]) * rs[ft] * wv + bv;
  }
}

EDTTS_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// g * sigmoid(g); v_rcp_f32 (1 ulp) instead of the 9-instruction IEEE divide -- relative error ~1e-7, far inside the parity budget
EDTTS_DEV float silu(float g) { return g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)); }

// v, gt: value / gate accumulators of one hidden tile (before bias) -> value * silu(gate)   (layers/transformer.py:21-23)
EDTTS_DEV f4 swiglu_tile(f4 v, f4 gt, f4 vb, f4 gb) {
  v += vb;  // bias after the GEMM: its load is off the MFMA critical path
  gt += gb;
  // written on vectors so that the mul / add halves pack (v_pk_*)
  const f4 e = {fast_exp2(gt[0] * -1.4426950408889634f), fast_exp2(gt[1] * -1.4426950408889634f),
                fast_exp2(gt[2] * -1.4426950408889634f), fast_exp2(gt[3] * -1.4426950408889634f)};
  const f4 d = e + 1.0f;
  const f4 rc = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
  return (v * gt) * rc;
}

// weight stream of the layer / prologue kernels: a per-wave register ring of HT fragments straight from L2; the four waves
// of a block share each fragment through the CU's L1 (measured 86 % TCP hit rate).
// (A block-shared LDS ring -- each fragment fetched once per block, ds_read_b128 to the MFMA -- was built and measured:
// 6 % slower at B=256, T=512, because its per-phase block barrier costs more than the L2 traffic it saves; see
// DESIGN.md "What was tried".)
// Ring size: HT fragments at NF = 2, HT/2 at NF = 4 -- the same prefetch distance in MFMAs (4*NF per fragment).
// Hidden 256 (HT = 16) takes HALF of that: its residual + normalised tiles are 256 registers by themselves, and with a 16-fragment
// ring the QKV-tail instance spilled (76 B/lane of scratch); 8 fragments = 64 MFMAs of prefetch distance.
template <class C> constexpr int wstream_ring() {
  constexpr int full = ((C::HT * 2) / C::NF >= 2 && C::HT % ((C::HT * 2) / C::NF) == 0) ? (C::HT * 2) / C::NF : C::HT;
  if (EDTTS_W2 && C::NF == 2 && C::H == 160) return 5;  // (half the prefetch distance in the wave's own MFMAs, the same in time at two waves per SIMD)
  return (C::HT >= 16 && full % 2 == 0) ? full / 2 : full;
}
template <class C> using WStream = FragRing<wstream_ring<C>()>;

// acc *= alpha for an MFMA accumulator without exposing VALU arithmetic on it to the compiler.  On the rare rescale path of the
// online softmax a plain `O *= alpha` makes hipcc hoist 24 v_accvgpr_reads of O into EVERY softmax step (speculatively, above the
// branch).  The asm keeps O in the AGPR class; s_nops cover the MFMA-write -> read, VALU -> accvgpr_write and write -> MFMA-read
// hazards, which the hazard recognizer cannot see inside inline asm (the path runs about once per head).
EDTTS_DEV void scale_acc(f4& o, float alpha) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = o[r], t;
#if EDTTS_W2
    asm volatile("s_nop 15\n\tv_mul_f32 %0, %0, %2\n\ts_nop 3" : "+v"(x), "=&v"(t) : "v"(alpha));
#else
    asm volatile("s_nop 15\n\tv_accvgpr_read_b32 %1, %0\n\ts_nop 1\n\tv_mul_f32 %1, %1, %2\n\ts_nop 1\n\tv_accvgpr_write_b32 %0, %1\n\ts_nop 3"
                 : "+a"(x), "=&v"(t)
                 : "v"(alpha));
#endif
    o[r] = x;
  }
}

// -DEDTTS_KV2=1: double-buffered K / V^T fragments (tiles of step s + 2 requested while step s computes).  Measured on MI355X
// (B=256, T=512): k_layer 0.992 ms against 0.954 ms with the single buffers -- the 44 extra registers cost more in accumulator
// shuffling than the 5.7 % of s_waitcnt time they could hide (the bf16 kernel, whose steps are 5x shorter, does gain from it).
#ifndef EDTTS_KV2
#define EDTTS_KV2 0
#endif
constexpr int kChunk = 2;  // key tiles (16 keys each) per online-softmax step
constexpr float kDefer = 32.f;  // octaves a chunk may exceed the softmax reference point before it is moved

template <class C>
struct KVFrag {  // MFMA A operands of one chunk of key tiles, for one head
  f4 ka[kChunk][C::DFULL > 0 ? C::DFULL : 1];
  f2 kr[kChunk];
};

template <class C>
struct VFrag {  // V^T fragments (MFMA A operand of P V) of one chunk of key tiles, for one head
  f4 v[kChunk][C::DT];
};

// ---------------------------------------------------------------------------------------------------------
// Multi-head attention for one wave's 16*NF query frames, fused with the output projection:
//   h += W_o . concat_heads( softmax(q k^T / sqrt(d) [band mask]) v )
// q comes from global memory (SELF: the q rows written by the previous kernel) or from this wave's LDS tile
// (cross-attention); K is row-major [key][H]; V is stored transposed [feature][key] so that both MFMA A operands
// are 16-byte loads.  Scores live only in registers: S^T tile = K Q^T (keys on MFMA rows, queries on lanes), an online
// softmax over 32-key chunks, and P^T (the C/D registers) is directly the B operand of O^T = V^T P^T.  The attention
// proper runs on PAIRS of query tiles ("halves" of 32 frames, NF/2 per wave); the per-head outputs O^T[dt] of all NF
// tiles then feed the k-major projection weights from the fragment ring in one NF-wide phase.
//   SELF : keys are frames of the same utterance, band |i-j| <= window (layers/attention.py:27-30,108-112)
//   !SELF: keys are the S context tokens, no mask (layers/mla.py:158-179)
// qload.q4(row16, col) / q2(row16, col) return q[m0 + 16*row16 + fq][col + 4 g ...] / [col + 2 g ...] for this lane (col wave-uniform).
// ---------------------------------------------------------------------------------------------------------
// OMODE: what happens to a head's normalised output O^T --
//   O_FUSED  project it at once: h += Wo[:, head] . O  (the product kernels)
//   O_DEFER  park it in LDS (obuf, B-operand layout) and run ONE projection over all heads behind the last head (EDTTS_W2)
//   O_LDS    park it in LDS and return: the caller projects (cooperative kernel, edtts_coop.h: heads hd0, hd0 + hstep, ... of a
//            tile are this wave's, the projection is split over the waves by OUTPUT tiles)
enum { O_FUSED = 0, O_DEFER = 1, O_LDS = 2 };
template <class C, bool SELF, int OMODE, class QLoad, class BeforeProject>
EDTTS_DEV void attention_fused(QLoad&& qload, const float* __restrict__ Kb, const float* __restrict__ VTb, int ldv,
                               int nkeys, int window, int m0w, int lane, WStream<C>& ring, f4 (&h)[C::HT][C::NF],
                               char* obuf, BeforeProject&& before_project,
                               unsigned long long* stamps = nullptr, int stamp_sel = 0, int hd0 = 0, int hstep = 1) {
  int sidx = 0;  // (EDTTS_STAMPS diagnostic builds: four stamps per step of the selected head)
  (void)sidx; (void)stamps; (void)stamp_sel;
  constexpr int DH = C::DH, DFULL = C::DFULL, DREM = C::DREM, DT = C::DT, H = C::H, CH = kChunk, NF = C::NF;
  constexpr int NHALF = C::NHALF;
  constexpr int QT = C::QT;  // query tiles per half: 2 (32 frames), or 1 in the small-batch instance (NF = 1)
  const int fq = lane & 15, g = lane >> 4;
  const float NEG_INF = -__builtin_inff();
  // Row map of the 8-feature remainder tile of V^T (head_dim % 16 == 8): MFMA row i = 4*gO + reg of the P V product carries
  // feature 2*gO + reg for reg < 2 (rows with reg >= 2 repeat a valid row and are never read), so the valid features of O^T land
  // in C/D registers 0 and 1 of every lane group and the output projection skips MFMA steps 2 and 3 of that k-tile.
  const int vrow_rem = 2 * (fq >> 2) + (fq & 1);

  // per-half geometry (a half = 32 query frames starting at m0 = m0w + 32*half)
  struct Geo {
    int m0, kt_lo, kt_hi, nchunk, klim;
    int cdiag;  // chunk that holds the keys of this half's own frames (self-attention; 0 otherwise)
    unsigned interior;  // bit c: chunk c (of the first 32 = 1024 keys) is interior (see chunk_is_interior) -- evaluated once per wave, one SALU bit test per step
    int lo_d[QT], span[QT];  // per-lane band limits on d = key - query: valid <=> (unsigned)(d - lo_d) <= span
  };
  auto make_geo = [&](int half) {
    Geo q;
    q.interior = 0u;
    q.m0 = m0w + 16 * QT * half;
    // The chunk partition and order are those of the enclosing 32-frame pair of query tiles also when a pass covers ONE tile
    // (small-batch instance, QT = 1): a query row then meets exactly the chunk sequence it meets in the 32-frame instances, and
    // the two give bitwise identical results (an utterance alone == the same utterance inside a large batch).
    const int mg = q.m0 & ~31;
    if (SELF && window >= 0) {
      const int lo = mg - window;
      q.kt_lo = ((lo > 0 ? lo : 0) >> 4) & ~(CH - 1);  // chunk grid aligned to the 32-frame tiles (see cdiag)
      const int hi = mg + 31 + window;  // last key any query of this pair may see
      const int last = (hi < nkeys - 1 ? hi : nkeys - 1);
      q.kt_hi = (last >> 4) + 1;
    } else {
      q.kt_lo = 0;
      q.kt_hi = (nkeys + 15) >> 4;
    }
    q.nchunk = (q.kt_hi - q.kt_lo + CH - 1) / CH;
    // The softmax is order-independent, and every valid query row can see its own key: starting with the chunk that holds the
    // diagonal guarantees that the (peeled) first step finds a visible key in every such row and can set its reference point.
    // (With the band start not on a chunk boundary -- windows that are not multiples of 16 -- chunk 0 is fully masked for some
    // rows that do see later keys; a reference left at 0 there underflows every exp2 when all scores are far below zero.)
    // Only needed when the band start is not on a chunk boundary (else chunk 0 already shows a key to every row that has one).
    const bool off_grid = SELF && window >= 0 && (mg - window > (q.kt_lo << 4));
    q.cdiag = off_grid ? ((mg >> 4) - q.kt_lo) / CH : 0;
    if (q.cdiag >= q.nchunk) q.cdiag = q.nchunk - 1;  // half entirely past the end of the utterance
    q.klim = (q.kt_hi << 4) < nkeys ? (q.kt_hi << 4) : nkeys;  // keys >= klim are never valid
#pragma unroll
    for (int ft = 0; ft < QT; ++ft) {
      const int qi = q.m0 + 16 * ft + fq;
      int lo = -(1 << 28), hi = q.klim - 1 - qi;
      if (SELF && window >= 0) {
        lo = -window;
        hi = hi < window ? hi : window;
      }
      q.lo_d[ft] = lo;
      q.span[ft] = hi - lo;  // negative -> nothing valid
    }
    return q;
  };

  // loads of one chunk's K / V^T fragments (tile index clamped: tiles past kt_hi are fully masked by klim).  Every address is
  // a wave-uniform base (SALU arithmetic on the tile / head indices) plus a per-lane 32-bit byte offset that never changes:
  // 1.5 % faster on the layer kernel than per-load 64-bit address arithmetic on the lanes.
  const unsigned koff = (unsigned)(fq * H + 4 * g) * 4u, koff_rem = (unsigned)(fq * H + 2 * g) * 4u;
  const unsigned voff = (unsigned)(fq * ldv + 4 * g) * 4u, voff_rem = (unsigned)(vrow_rem * ldv + 4 * g) * 4u;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(Kb), rsV = make_rsrc(VTb);
  auto load_k = [&](const Geo& q, int hd, int c, KVFrag<C>& f) {
    c = c < q.nchunk ? c : q.nchunk - 1;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      int kt = q.kt_lo + c * CH + t;
      kt = kt < q.kt_hi ? kt : q.kt_hi - 1;
      const unsigned so = (unsigned)((kt << 4) * H + hd * DH) * 4u;  // uniform byte offset of (key tile, head)
#pragma unroll
      for (int a = 0; a < DFULL; ++a) f.ka[t][a] = bufld4(rsK, koff + 64u * a, so);
      if (DREM) f.kr[t] = bufld2(rsK, koff_rem + 64u * DFULL, so);
    }
  };
  auto load_v = [&](const Geo& q, int hd, int c, VFrag<C>& f) {
    c = c < q.nchunk ? c : q.nchunk - 1;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      int kt = q.kt_lo + c * CH + t;
      kt = kt < q.kt_hi ? kt : q.kt_hi - 1;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const unsigned so = (unsigned)((hd * DH + 16 * dt) * ldv + (kt << 4)) * 4u;  // uniform
        f.v[t][dt] = bufld4(rsV, (DREM && dt == DT - 1) ? voff_rem : voff, so);
      }
    }
  };
#ifndef EDTTS_SPLITLOAD
#define EDTTS_SPLITLOAD 1
#endif
#define EDTTS_SPLIT_ON (EDTTS_SPLITLOAD && !EDTTS_KV2)
#if EDTTS_SPLIT_ON
  // The same requests in pieces, for re-requesting a buffer piece by piece as soon as the MFMAs that read that piece have been
  // issued (see step): group a of the K fragments of both key tiles (a == DFULL: the 8-feature remainder), key tile t of V^T.
  auto chunk_tile = [&](const Geo& q, int c, int t) {
    c = c < q.nchunk ? c : q.nchunk - 1;
    const int kt = q.kt_lo + c * CH + t;
    return kt < q.kt_hi ? kt : q.kt_hi - 1;
  };
  auto load_k_group = [&](const Geo& q, int hd, int c, KVFrag<C>& f, int a) {
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const unsigned so = (unsigned)((chunk_tile(q, c, t) << 4) * H + hd * DH) * 4u;
      if (a < DFULL) f.ka[t][a < DFULL ? a : 0] = bufld4(rsK, koff + 64u * a, so);
      else if (DREM) f.kr[t] = bufld2(rsK, koff_rem + 64u * DFULL, so);
    }
  };
  auto load_v_tile = [&](const Geo& q, int hd, int c, VFrag<C>& f, int t) {
    const int kt = chunk_tile(q, c, t);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const unsigned so = (unsigned)((hd * DH + 16 * dt) * ldv + (kt << 4)) * 4u;
      f.v[t][dt] = bufld4(rsV, (DREM && dt == DT - 1) ? voff_rem : voff, so);
    }
  };
#endif
  // Mask of chunk c as the INITIAL accumulator of its K Q^T product: 0 where the key is visible, -inf elsewhere
  // (-inf + finite products = -inf), so no VALU work sits between the MFMA results and the softmax.  Interior chunks (every
  // key inside the band of every query of the half and below klim: 3 of the 5 chunks at window 64, all of the cross-attention
  // when S % 32 == 0) take a wave-uniform path whose first MFMA of each chain has the inline constant 0 as accumulator input:
  // no mask arithmetic and no accumulator initialisation at all (32 VALU instructions per chunk less).
  auto chunk_is_interior = [&](const Geo& q, int c) {
    c = c < q.nchunk ? c : q.nchunk - 1;
    const int k0 = (q.kt_lo + c * CH) << 4, k1 = k0 + 16 * CH - 1;
    bool full = k1 < q.klim && (q.kt_lo + (c + 1) * CH) <= q.kt_hi;
    if (SELF && window >= 0) full = full && (k1 - q.m0 <= window) && (k0 - (q.m0 + 16 * QT - 1) >= -window);
    return full;
  };
  auto mask_init = [&](const Geo& q, int c, f4 (&S)[CH][QT], const float (&vis)[QT]) {  // vis: value of a VISIBLE position
    c = c < q.nchunk ? c : q.nchunk - 1;
    const int k0 = (q.kt_lo + c * CH) << 4;
#pragma unroll
    for (int ft = 0; ft < QT; ++ft) {
      const int d0 = k0 + 4 * g - (q.m0 + 16 * ft + fq) - q.lo_d[ft];  // (key - query - lo_d) of r = 0, tile 0
      const unsigned sp = q.span[ft] >= 0 ? (unsigned)q.span[ft] : 0u;
      const int bias = q.span[ft] >= 0 ? 0 : (1 << 30);                 // nothing valid for this query
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][ft][r] = (unsigned)(d0 + bias + 16 * t + r) <= sp ? vis[ft] : NEG_INF;
    }
  };
  // S^T chunk = mask + K Q^T for both query tiles, accumulators interleaved (2*CH independent chains)
  // With FOLD the accumulators start at -m (the softmax reference point of the row) instead of 0, so the scores come out as
  // s - m and exp2 applies to them directly: m only changes on the rare rescale path, and folding it into the accumulator input
  // removes a v_sub per score from every step.  nm[ft] = -m as a scalar (edge chunks: the mask select picks it instead of 0),
  // NM[ft] = the same value as a whole accumulator tile (interior chunks).
  auto qk = [&](auto fold_tag, const Geo& q, int c, const KVFrag<C>& f, const f4 (&qa)[QT][DFULL > 0 ? DFULL : 1],
                const f2 (&qr)[QT], f4 (&S)[CH][QT], const f4 (&NM)[QT], const float (&nm)[QT], auto&& after_group) {
    static_assert(DFULL >= 1, "head_dim >= 16 expected");
    constexpr bool FOLD = decltype(fold_tag)::value;
    const int cc = c < q.nchunk ? c : q.nchunk - 1;
    if (cc < 32 && ((q.interior >> cc) & 1u)) {  // (chunks past 31 -- beyond the reference's length limits -- take the always-correct masked path)
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int ft = 0; ft < QT; ++ft)
          S[t][ft] = FOLD ? EDTTS_MFMA(f.ka[t][0][0], qa[ft][0][0], NM[ft]) : EDTTS_MFMA(f.ka[t][0][0], qa[ft][0][0], splat(0.f));
    } else {
      const float zero[QT] = {};
      mask_init(q, c, S, FOLD ? nm : zero);
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int ft = 0; ft < QT; ++ft) S[t][ft] = EDTTS_MFMA(f.ka[t][0][0], qa[ft][0][0], S[t][ft]);
    }
#pragma unroll
    for (int a = 0; a < DFULL; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (a == 0 && b == 0) continue;
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
          for (int ft = 0; ft < QT; ++ft) S[t][ft] = EDTTS_MFMA(f.ka[t][a][b], qa[ft][a][b], S[t][ft]);
        if (b == 3) after_group(a);  // the MFMAs that read fragment group a have been issued
      }
    if (DREM) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
          for (int ft = 0; ft < QT; ++ft) S[t][ft] = EDTTS_MFMA(f.kr[t][b], qr[ft][b], S[t][ft]);
      after_group(DFULL);
    }
  };

  Geo geo[NHALF];
#pragma unroll
  for (int hf = 0; hf < NHALF; ++hf) {
    geo[hf] = make_geo(hf);
    unsigned im = 0;
    for (int c = 0; c < geo[hf].nchunk && c < 32; ++c) im |= (chunk_is_interior(geo[hf], c) ? 1u : 0u) << c;
    geo[hf].interior = __builtin_amdgcn_readfirstlane(im);
  }

  // q fragments (B operand: lane (fq,g) holds q[query][hd*DH + 16a + 4g + b]) and the K / V^T fragments of the first
  // chunks of a (head, half).  For half 0 they are fetched while the PREVIOUS head's projection phases run, so no head
  // starts on an exposed load.
  f4 qa_n[QT][DFULL > 0 ? DFULL : 1];
  f2 qr_n[QT];
  KVFrag<C> KA;
  VFrag<C> VA;
#if EDTTS_KV2
  KVFrag<C> KB;  // second K / V^T buffer: the tiles of step s + 2 are requested while step s computes
  VFrag<C> VB;
#endif
  // step s = 0: the diagonal chunk; steps 1 .. nchunk-1: the other chunks in ascending order
  auto chunk_at = [&](const Geo& q, int st) {
    return st >= q.nchunk ? q.nchunk - 1 : (st == 0 ? q.cdiag : (st <= q.cdiag ? st - 1 : st));
  };
  auto load_q = [&](int hd, int half) {
#pragma unroll
    for (int ft = 0; ft < QT; ++ft) {
#pragma unroll
      for (int a = 0; a < DFULL; ++a) qa_n[ft][a] = qload.q4(QT * half + ft, hd * DH + 16 * a);  // (+ 4 g: the lane's quad, in the loader)
      if (DREM) qr_n[ft] = qload.q2(QT * half + ft, hd * DH + 16 * DFULL);                          // (+ 2 g)
    }
  };
  auto prefetch = [&](const Geo& q, int hd, int half) {
    load_q(hd, half);
    load_k(q, hd, q.cdiag, KA);  // the first step processes the diagonal chunk
    load_v(q, hd, q.cdiag, VA);
#if EDTTS_KV2
    load_k(q, hd, chunk_at(q, 1), KB);
    load_v(q, hd, chunk_at(q, 1), VB);
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  if (hd0 < C::HEADS) prefetch(geo[0], hd0, 0);

  for (int hd = hd0; hd < C::HEADS; hd += hstep) {
    f4 O[DT][NF];
#pragma unroll
    for (int hf = 0; hf < NHALF; ++hf) {
      const Geo& q = geo[hf];
      if (hf > 0) prefetch(q, hd, hf);
      // (q arrives pre-scaled by log2(e)/sqrt(d) -- the factor is folded into the packed query weights -- so the scores come
      // out of the MFMA in the exp2 domain, and the prefetched fragments are used as they are)
      const auto& qa = qa_n;
      const auto& qr = qr_n;
      f4 lvec[QT];  // per-lane partial row sums (reduced over r and the lane groups at the end)
#pragma unroll
      for (int ft = 0; ft < QT; ++ft) lvec[ft] = splat(0.f);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ft = 0; ft < QT; ++ft) O[dt][QT * hf + ft] = splat(0.f);
      const int nchunk = q.nchunk;

      // Strictly serial step, single S / K / V^T buffers:  K Q^T(c) | softmax(c) | P V(c).  Nothing overlaps an fp32 MFMA on this
      // chip, so producing the scores of chunk c+1 "under" the softmax of chunk c buys nothing -- but its second
      // S buffer and the double-buffered K / V^T fragments cost 60 registers and the rotation moves between them.  Each buffer is
      // reloaded right after the MFMAs that read it were issued and consumed one step later (loop-carried: hipcc cannot sink it).
      //
      // Online softmax with a DEFERRED running maximum: fp32 accumulators have ~2^127 of headroom, so the reference point m only
      // has to move when a chunk exceeds it by more than 2^kDefer; until then P = 2^(s - m) <= 2^kDefer and neither O nor the
      // row sums need rescaling (softmax is invariant to m).  The first step sets m to its chunk's row maximum; from the second
      // step on the scores come out of the MFMAs as s - m (see qk) and the wave-uniform rescale branch is taken only when some
      // row jumps by more than kDefer octaves -- the read-modify-write of O, the cross-lane max and the subtraction of m stay
      // out of the common path.  (Chunk order: see Geo::cdiag.)
      f4 S[CH][QT];
      f4 NM[QT];
      float nm[QT];
#pragma unroll
      for (int ft = 0; ft < QT; ++ft) {
        NM[ft] = splat(0.f);
        nm[ft] = 0.f;
      }
      using Yes = std::integral_constant<bool, true>;
      using No = std::integral_constant<bool, false>;
      auto step = [&](auto fold_tag, int c, int cnext, KVFrag<C>& KA, VFrag<C>& VA, int hd_req) {  // c: this step's chunk; (hd_req, cnext): the head and chunk to request into the buffers
        constexpr bool FOLD = decltype(fold_tag)::value;
#ifdef EDTTS_STAMPS
        if (hd == STAMP_SEL_HEAD(stamp_sel)) STAMPX(stamps, sidx++, stamp_sel);
#endif
#if EDTTS_SPLIT_ON
        // Each K fragment group is re-requested right after the score MFMAs that read it were issued (3 bursts of 2 requests inside
        // the 40-MFMA run instead of one burst of 6 behind it): the group the next step needs FIRST is in flight ~1 000 cycles
        // longer.  PMC: s_waitcnt took 6 % of the wave's cycles, most of it here (profiles/r03_diag_phases.txt).
        qk(fold_tag, q, c, KA, qa, qr, S, NM, nm, [&](int a) {
          __builtin_amdgcn_sched_barrier(EDTTS_PIN_MASK);
          load_k_group(q, hd_req, cnext, KA, a);
          __builtin_amdgcn_sched_barrier(0);
        });
        __builtin_amdgcn_sched_barrier(0);
#else
        qk(fold_tag, q, c, KA, qa, qr, S, NM, nm, [](int) {});
        __builtin_amdgcn_sched_barrier(0);
#ifndef EDTTS_ABLATE_KVLOADS  // timing ablation only
        load_k(q, hd_req, cnext, KA);  // (the last step re-reads its own tiles)
#endif
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef EDTTS_STAMPS
        if (hd == STAMP_SEL_HEAD(stamp_sel)) STAMPX(stamps, sidx++, stamp_sel);
#endif
        // one VGPR copy of the scores serves exp2 and the rare rescale (the pin keeps hipcc from re-reading the accumulators
        // after the branch)
        f4 sv[CH][QT];
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
          for (int ft = 0; ft < QT; ++ft) {
            sv[t][ft] = S[t][ft];
            asm volatile("" : "+v"(sv[t][ft]));
          }
        auto lane_max = [&](int ft) {  // maximum over this lane's 4*CH scores of query tile ft
          f4 mv = sv[0][ft];
#pragma unroll
          for (int t = 1; t < CH; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mv[r], sv[t][ft][r]);
          return hmax(mv);
        };
        f4 P[CH][QT], ps[QT];  // ps: this chunk's partial row sums (per lane)
        auto exp_and_sum = [&](int ft, float m) {
#pragma unroll
          for (int t = 0; t < CH; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[t][ft][r] = fast_exp2(sv[t][ft][r] - m);
          ps[ft] = P[0][ft];
#pragma unroll
          for (int t = 1; t < CH; ++t) ps[ft] += P[t][ft];
        };
        if (!FOLD) {
          // first chunk: the reference point is this chunk's row maximum (0 for a row without any visible key: stays finite)
#pragma unroll
          for (int ft = 0; ft < QT; ++ft) {
            const float gm = group_max(lane_max(ft));  // identical on the 4 lanes of a row
            const float m = gm > -1e30f ? gm : 0.f;
            nm[ft] = -m;
            NM[ft] = splat(-m);
            EDTTS_PIN_ACC(NM[ft]);  // the -m tile lives in AGPRs: it is an accumulator input and nothing else
            exp_and_sum(ft, m);
          }
        } else {
          // The scores are already relative to the reference point.  Whether it has to move is read off the partial row sums,
          // which are needed anyway: sum(P) > 2^kDefer (or not finite) <=> some score may exceed m by more than kDefer octaves
          // (no false negatives: sum >= max; a false positive only moves the reference early).  The per-lane maximum, the
          // cross-lane max and the read-modify-write of O then stay out of the common path entirely.
          const float lim = 4294967296.f;  // 2^kDefer
          static_assert(kDefer == 32.f, "lim above is 2^kDefer");
          // ONE test per step: all probabilities are >= 0, so the sum over both query tiles exceeds the limit (or is not finite)
          // whenever one of them does
          f4 pt = splat(0.f);
#pragma unroll
          for (int ft = 0; ft < QT; ++ft) {
            exp_and_sum(ft, 0.f);
            pt += ps[ft];
          }
          const bool over = !(hsum(pt) <= lim);
          if (__any(over)) {
#pragma unroll
            for (int ft = 0; ft < QT; ++ft) {
              const float delta = fmaxf(0.f, group_max(lane_max(ft)));  // rows that did not jump keep their reference
              const float alpha = fast_exp2(-delta);
              nm[ft] -= delta;
              NM[ft] = splat(nm[ft]);
              EDTTS_PIN_ACC(NM[ft]);
              lvec[ft] *= alpha;
#pragma unroll
              for (int dt = 0; dt < DT; ++dt) scale_acc(O[dt][QT * hf + ft], alpha);
              exp_and_sum(ft, delta);
            }
          }
        }
#pragma unroll
        for (int ft = 0; ft < QT; ++ft) lvec[ft] += ps[ft];
        __builtin_amdgcn_sched_barrier(0);
#ifdef EDTTS_STAMPS
        if (hd == STAMP_SEL_HEAD(stamp_sel)) STAMPX(stamps, sidx++, stamp_sel);
#endif
#pragma unroll
        for (int t = 0; t < CH; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
              for (int ft = 0; ft < QT; ++ft) O[dt][QT * hf + ft] = EDTTS_MFMA(VA.v[t][dt][r], P[t][ft][r], O[dt][QT * hf + ft]);
#if EDTTS_SPLIT_ON
          __builtin_amdgcn_sched_barrier(EDTTS_PIN_MASK);
          load_v_tile(q, hd_req, cnext, VA, t);  // this key tile's V^T fragments have been read: re-request them now
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#if !EDTTS_SPLIT_ON && !defined(EDTTS_ABLATE_KVLOADS)
        load_v(q, hd_req, cnext, VA);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef EDTTS_STAMPS
        if (hd == STAMP_SEL_HEAD(stamp_sel)) STAMPX(stamps, sidx++, stamp_sel);
#endif
      };
      // step s = 0: the diagonal chunk; steps 1 .. nchunk-1: the other chunks in ascending order
      const int cd = q.cdiag;
      auto chunk_of = [&](int st) { return st >= nchunk ? nchunk - 1 : (st == 0 ? cd : (st <= cd ? st - 1 : st)); };
#if EDTTS_KV2
      step(No{}, cd, chunk_of(2), KA, VA, hd);
      for (int st = 1; st < nchunk; st += 2) {
        step(Yes{}, chunk_of(st), chunk_of(st + 2), KB, VB, hd);
        if (st + 1 < nchunk) step(Yes{}, chunk_of(st + 1), chunk_of(st + 3), KA, VA, hd);
      }
#else
      // (Requesting the NEXT head's first q / K / V^T tiles from the head's last step instead of from its projection phase -- so
      // that the phase's weight-ring waits do not queue behind HBM-latency requests in the in-order vmcnt -- was built and measured:
      // 0.9369 vs 0.9307 ms per launch, slower; the per-step q re-request it needs costs more than the phase gains.)
#ifndef EDTTS_HANDOVER
#define EDTTS_HANDOVER 1
#endif
      // EDTTS_HANDOVER (round 4): a head's LAST step requests the NEXT head's first K / V^T tiles instead of re-reading its own -- the
      // requests are issued anyway, and the next head starts on chunk cd too (same geometry) -- and only the q fragments are fetched
      // from the projection phase: the phase's ring refills then queue behind 5 q requests (rows this XCD wrote: L2 hits) instead of 17
      // HBM-latency ones in the in-order vmcnt.  Same device, three interleaved runs: k_layer 0.9014-0.9062 ms against 0.9150-0.9168, the
      // whole call 15.02-15.09 against 15.26-15.31 ms; results bitwise unchanged.  (Round 3's attempt re-requested q in every step
      // and lost 0.7 %.)
      constexpr bool HANDOVER = EDTTS_HANDOVER && NHALF == 1;
      const bool more = HANDOVER && hd + hstep < C::HEADS;
      auto req_head = [&](int st) { return (more && st == nchunk - 1) ? hd + hstep : hd; };
      auto req_chunk = [&](int st) { return (more && st == nchunk - 1) ? cd : chunk_of(st + 1); };
      step(No{}, cd, req_chunk(0), KA, VA, req_head(0));
      for (int st = 1; st < nchunk; ++st) step(Yes{}, chunk_of(st), req_chunk(st), KA, VA, req_head(st));
#endif
      // normalise this half's rows
#pragma unroll
      for (int ft = 0; ft < QT; ++ft) {
        const float lt = group_sum(hsum(lvec[ft]));
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;  // (IEEE divide, once per head and row: its rounding is part of the parity record)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) O[dt][QT * hf + ft] *= inv;
      }
    }
    // ---- project: h[nt] += Wo[:, head features] . O  (all NF frame tiles at once) ------------------------------------
#ifdef EDTTS_STAMPS
    if (hd == STAMP_SEL_HEAD(stamp_sel)) STAMPX(stamps, sidx++, stamp_sel);  // (normalisation done: start of the projection phases)
#endif
    if (hd + hstep < C::HEADS) {
      if constexpr (EDTTS_HANDOVER && NHALF == 1) {
        load_q(hd + hstep, 0);  // (its K / V^T tiles were requested by the last step)
        __builtin_amdgcn_sched_barrier(0);
      } else {
        prefetch(geo[0], hd + hstep, 0);
      }
    }
    if constexpr (OMODE != O_FUSED) {
      // this head's O^T tiles wait in LDS as they stand (C/D layout = the projection's B operand: lane-contiguous, conflict-free)
      char* ob = obuf + hd * C::OHEAD_BYTES;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          if (DREM && dt == DT - 1) reinterpret_cast<f2*>(ob + DFULL * NF * 1024 + ft * 512)[lane] = f2{O[dt][ft][0], O[dt][ft][1]};
          else reinterpret_cast<f4*>(ob + (dt * NF + ft) * 1024)[lane] = O[dt][ft];
        }
    } else {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (DREM && dt == DT - 1) ktile_phase<C::HT, WStream<C>::RN_, NF, 2>(ring, O[dt], h);  // remainder tile: valid k in steps 0, 1 only
        else ktile_phase<C::HT>(ring, O[dt], h);
      }
    }
#ifdef EDTTS_STAMPS
    if (hd == STAMP_SEL_HEAD(stamp_sel)) STAMPX(stamps, sidx++, stamp_sel);  // end of the head
#endif
  }
  if constexpr (OMODE == O_DEFER) {
    // one projection over all heads, in the order the per-head phases have (same accumulation order: bitwise the same result)
    before_project();
    for (int hd = 0; hd < C::HEADS; ++hd) {
      const char* ob = obuf + hd * C::OHEAD_BYTES;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f4 in[NF];
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          if (DREM && dt == DT - 1) {
            const f2 t = reinterpret_cast<const f2*>(ob + DFULL * NF * 1024 + ft * 512)[lane];
            in[ft] = f4{t[0], t[1], 0.f, 0.f};
          } else {
            in[ft] = reinterpret_cast<const f4*>(ob + (dt * NF + ft) * 1024)[lane];
          }
        }
        if (DREM && dt == DT - 1) ktile_phase<C::HT, WStream<C>::RN_, NF, 2>(ring, in, h);
        else ktile_phase<C::HT>(ring, in, h);
      }
    }
  }
}

// Elementwise DDIM arithmetic with the reference's operation order and IEEE rounding of every operation
// (hipcc defaults to -ffp-contract=fast-honor-pragmas: the pragma keeps mul/sub and mul/add unfused).
EDTTS_DEV void ddim_elem(float x, float e, float s1m, float sab, float sabp, float cdir, float& x0, float& xp) {
#pragma clang fp contract(off)
  float t1 = s1m * e;
  float t2 = x - t1;
  float v = div_rn(t2, sab);          // schedule.py:189  (x_t - sqrt(1-ab)*eps) / sqrt(ab)
  v = fminf(fmaxf(v, -3.0f), 3.0f);   // schedule.py:190
  x0 = v;
  float t3 = sabp * v;
  float t4 = cdir * e;                // schedule.py:196  direction uses the raw eps
  xp = t3 + t4;                       // schedule.py:200
}
struct DdimCoef {
  float s1m, sab, sabp, cdir, sigma;
};
EDTTS_DEV DdimCoef ddim_coef(float ab, float abp, float eta) {
#pragma clang fp contract(off)
  DdimCoef c;
  c.s1m = sqrt_rn(1.0f - ab);
  c.sab = sqrt_rn(ab);
  float r1 = div_rn(1.0f - abp, 1.0f - ab);
  float r2 = 1.0f - div_rn(ab, abp);
  float pr = r1 * r2;
  c.sigma = eta * sqrt_rn(pr);        // schedule.py:193-195
  float s2 = c.sigma * c.sigma;
  float in = 1.0f - abp;
  in = in - s2;
  c.cdir = sqrt_rn(in);               // schedule.py:196
  c.sabp = sqrt_rn(abp);
  return c;
}
EDTTS_DEV float add_mul_rn(float p, float s, float n) {
#pragma clang fp contract(off)
  float t = s * n;
  return p + t;
}

struct DdpmCoef {
  float coef1, coef2, sd;
};
EDTTS_DEV DdpmCoef ddpm_coef(float al, float ab, float be, float var, bool nonzero) {
#pragma clang fp contract(off)
  DdpmCoef c;
  c.coef1 = div_rn(1.0f, sqrt_rn(al));    // schedule.py:227
  float om = 1.0f - ab;
  c.coef2 = div_rn(be, sqrt_rn(om));      // schedule.py:228
  c.sd = (nonzero ? 1.0f : 0.0f) * sqrt_rn(var);  // mask * sqrt(var), schedule.py:232-237
  return c;
}
EDTTS_DEV float ddpm_elem(float x, float e, float n, DdpmCoef c) {
#pragma clang fp contract(off)
  float t1 = c.coef2 * e;
  float t2 = x - t1;
  float mean = c.coef1 * t2;
  float t3 = c.sd * n;
  return mean + t3;
}

// Multistep x0-solver update of one element (schedule.py:339-438), same operation order, no fma contraction.
struct LmsCoef {
  int mode;
  float p0, p1, c0, c1, rinv, cB, cC;
};
EDTTS_DEV void lms_elem(float x, float out, float h_new, float h_old, const LmsCoef& k, float& x0, float& xn) {
#pragma clang fp contract(off)
  float a = k.p0 * x;
  float b = k.p1 * out;
  float v = a + b;                        // schedule.py:125 (sqrt_ab * x_t - sqrt_1mab * v), or the model output itself
  v = fminf(fmaxf(v, -3.0f), 3.0f);       // schedule.py:487
  x0 = v;
  float t0 = k.c0 * x;
  float t1 = k.c1 * v;
  float acc = t0 + t1;
  if (k.mode == 2) {
    float d = v - h_new;
    float d1 = k.rinv * d;                // D1 = (1 / r) * (x0_pred - x0_prev)
    float t2 = k.cB * d1;
    t2 = t2 * 0.5f;
    acc = acc + t2;
  } else if (k.mode == 3) {
    float d1 = v - h_old;                 // D1 = x0_preds[0] - x0_preds[1]
    float two = 2.0f * h_old;
    float d2 = v - two;
    d2 = d2 + h_new;                      // D2 = x0_preds[0] - 2 * x0_preds[1] + x0_preds[2]
    float t2 = k.cB * d1;
    t2 = t2 * 0.5f;
    float t3 = k.cC * d2;
    t3 = t3 / 6.0f;
    acc = acc + t2;
    acc = acc + t3;
  }
  xn = acc;
}

// v-prediction sampler update of one element (inference_pipeline.py:127-132 / :187-192 with schedule.py:121-125,138-140), same
// operation order, no fma contraction:  x0 = clamp(sab x - s1m v);  eps = s1m x + sab v;  x_next = san x0 + s1mn eps
struct VpredCoef {
  float sab, s1m, san, s1mn, cfg;
};
EDTTS_DEV float vpred_elem(float x, float v, const VpredCoef& k) {
#pragma clang fp contract(off)
  float a = k.sab * x;
  float b = k.s1m * v;
  float x0 = a - b;
  x0 = fminf(fmaxf(x0, -3.0f), 3.0f);
  float c = k.s1m * x;
  float d = k.sab * v;
  float e = c + d;
  float p = k.san * x0;
  float q = k.s1mn * e;
  return p + q;
}
// q_sample (schedule.py:81-84): sqrt_ab * x0 + sqrt_1mab * noise
EDTTS_DEV float qsample_elem(float x0, float sab, float noise, float s1m) {
#pragma clang fp contract(off)
  float a = sab * x0;
  float b = s1m * noise;
  return a + b;
}
// classifier-free guidance (inference_pipeline.py:183): v = v_uncond + scale * (v_cond - v_uncond)
EDTTS_DEV float cfg_combine(float vc, float vu, float scale) {
#pragma clang fp contract(off)
  float d = vc - vu;
  float s = scale * d;
  return vu + s;
}

// Philox4x32-10 counter-based generator (Salmon et al., SC'11) -> four standard normals per call (Box-Muller).
// counter = (element index lo, hi, step, 0), key = (seed lo, hi): every (seed, step, element) gets its own stream, so the
// result does not depend on how elements are distributed over waves / GPUs PROVIDED the caller passes the GLOBAL element index
// (the sharded callers add their shard's first-element offset: KArgs::philox_base, edtts_randn's elem_offset).
EDTTS_DEV f4 philox_normal4(unsigned long long seed, unsigned step, unsigned long long index) {
  unsigned c0 = (unsigned)index, c1 = (unsigned)(index >> 32), c2 = step, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const float inv = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = ((float)c0 + 1.0f) * inv, u1 = (float)c1 * inv, u2 = ((float)c2 + 1.0f) * inv, u3 = (float)c3 * inv;
  const float r0 = sqrtf(-2.0f * __logf(u0 > 1.0f ? 1.0f : u0)), r1 = sqrtf(-2.0f * __logf(u2 > 1.0f ? 1.0f : u2));
  const float a0 = 6.28318530717958647692f * u1, a1 = 6.28318530717958647692f * u3;
  return f4{r0 * __cosf(a0), r0 * __sinf(a0), r1 * __cosf(a1), r1 * __sinf(a1)};
}

}  // namespace edtts
