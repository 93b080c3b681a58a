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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define EDTTS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define EDTTS_DEV __device__ __forceinline__

namespace edtts {

constexpr int kWaveFrames = 32;  // frames per wave
constexpr int kCtxWaves = 4;     // waves per block of the context kernel (no block-level sharing there)

template <int H_, int HEADS_, int MEL_>
struct Cfg {
  static constexpr int H = H_, HEADS = HEADS_, MEL = MEL_;
  static constexpr int DH = H / HEADS;            // head dim (40 for the default decoder)
  static constexpr int DFULL = DH / 16;           // full 16-wide groups of the head dim
  static constexpr int DREM = DH % 16;            // remainder (0 or 8 supported)
  static constexpr int DT = (DH + 15) / 16;       // 16-row tiles of the P.V output per head
  static constexpr int DHP = DT * 16;             // head dim padded to the MFMA tile
  static constexpr int HT = H / 16;               // feature tiles of the hidden dim
  static constexpr int MT = MEL / 16;             // feature tiles of the mel dim
  static constexpr int R = H / 2, RT = R / 16;    // kv_lora_rank (transformer.py:113) and its tiles
  static constexpr int VR = (HEADS - 1) * DH + DHP;  // rows of a V^T buffer (last head padded)
  static constexpr int QLD = H + 4;               // LDS row stride (floats) of the cross-attention q tile
  static constexpr int WAVES = H > 192 ? 2 : 4;   // waves per block: bounded by LDS (weight ring + one q tile per wave)
  static constexpr int THREADS = 64 * WAVES;
  static_assert(H % 32 == 0 && MEL % 16 == 0 && H % HEADS == 0, "dims");
  static_assert(DREM == 0 || DREM == 8, "head_dim % 16 must be 0 or 8");
};

EDTTS_DEV f4 splat(float v) { return f4{v, v, v, v}; }
EDTTS_DEV f4 ldg4(const float* p) { return *reinterpret_cast<const f4*>(p); }
EDTTS_DEV f2 ldg2(const float* p) { return *reinterpret_cast<const f2*>(p); }
EDTTS_DEV void stg4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

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
// (prefetch distance RN fragments = 8*RN MFMAs; RN must divide every phase length).  Refilling after use matters:
// loading into the slot before its old value is consumed makes hipcc copy the whole ring and wait for every load at
// the top of each phase (measured 126 -> 137 TFLOP/s on the bare stream).  The sched_barrier pins the issue point:
// without it the pre-RA scheduler sinks each load to its use RN fragments later, i.e. load -> s_waitcnt -> MFMA.
// ---------------------------------------------------------------------------------------------------------
template <int RN>
struct FragRing {
  const f4* p;  // lane-offset pointer to position 0 of the current phase
  f4 r[RN];
  EDTTS_DEV void prime(const float* base, int lane) {
    p = reinterpret_cast<const f4*>(base) + lane;
#pragma unroll
    for (int i = 0; i < RN; ++i) r[i] = p[i * 64];
  }
  EDTTS_DEV const f4& at(int i) const { return r[i % RN]; }
  EDTTS_DEV void refill(int i) {
    r[i % RN] = p[(i + RN) * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  EDTTS_DEV void advance(int n) { p += n * 64; }
};

// out^T tile (16 features x 32 frames) += sum_kt frag(kt) * in[kt]: one n-major phase of KT fragments.  Even / odd k-tiles
// go to separate accumulators (4 independent MFMA chains per wave: v_mfma_f32_16x16x4_f32 has a 40-cycle dependent
// latency at a 32-cycle issue interval) and are summed at the end, which also halves each fp32 summation chain.
template <int KT, int RN>
EDTTS_DEV void gemm_phase(FragRing<RN>& ring, const f4 (&in)[KT][2], f4& acc0, f4& acc1) {
  static_assert(KT % RN == 0, "phase length must be a multiple of the ring size");
  f4 b0 = splat(0.f), b1 = splat(0.f);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const f4& a = ring.at(kt);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (kt & 1) {
        b0 = EDTTS_MFMA(a[r], in[kt][0][r], b0);
        b1 = EDTTS_MFMA(a[r], in[kt][1][r], b1);
      } else {
        acc0 = EDTTS_MFMA(a[r], in[kt][0][r], acc0);
        acc1 = EDTTS_MFMA(a[r], in[kt][1][r], acc1);
      }
    }
    ring.refill(kt);
  }
  acc0 += b0;
  acc1 += b1;
  ring.advance(KT);
}

// Two n-tiles at once from a stream that interleaves their fragments per k-tile ([kt][tile a | tile b]): the four
// accumulators are the four independent chains.  Used for the FFN value/gate pair (layers/transformer.py:21-23).
template <int KT, int RN>
EDTTS_DEV void gemm_phase_pair(FragRing<RN>& ring, const f4 (&in)[KT][2], f4& a0, f4& a1, f4& b0, f4& b1) {
  static_assert((2 * KT) % RN == 0, "phase length must be a multiple of the ring size");
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const f4& fa = ring.at(2 * kt);
    const f4& fb = ring.at(2 * kt + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0 = EDTTS_MFMA(fa[r], in[kt][0][r], a0);
      a1 = EDTTS_MFMA(fa[r], in[kt][1][r], a1);
      b0 = EDTTS_MFMA(fb[r], in[kt][0][r], b0);
      b1 = EDTTS_MFMA(fb[r], in[kt][1][r], b1);
    }
    ring.refill(2 * kt);
    ring.refill(2 * kt + 1);
  }
  ring.advance(2 * KT);
}

// acc[nt] += frag(nt) * in   for one k-tile of a k-major packed matrix (NT fragments); two n-tiles are interleaved so
// that four accumulator chains are in flight.
template <int NT, int RN>
EDTTS_DEV void ktile_phase(FragRing<RN>& ring, f4 in0, f4 in1, f4 (&acc)[NT][2]) {
  static_assert(NT % RN == 0, "phase length must be a multiple of the ring size");
#pragma unroll
  for (int nt = 0; nt < NT; nt += 2) {
    const f4& fa = ring.at(nt);
    const f4& fb = ring.at(nt + 1 < NT ? nt + 1 : nt);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[nt][0] = EDTTS_MFMA(fa[r], in0[r], acc[nt][0]);
      acc[nt][1] = EDTTS_MFMA(fa[r], in1[r], acc[nt][1]);
      if (nt + 1 < NT) {
        acc[nt + 1][0] = EDTTS_MFMA(fb[r], in0[r], acc[nt + 1][0]);
        acc[nt + 1][1] = EDTTS_MFMA(fb[r], in1[r], acc[nt + 1][1]);
      }
    }
    ring.refill(nt);
    if (nt + 1 < NT) ring.refill(nt + 1);
  }
  ring.advance(NT);
}

// ---------------------------------------------------------------------------------------------------------
// Norms in the register layout.  x[t][ft] covers features 16t+4g+r of frame (ft, fq).
// ---------------------------------------------------------------------------------------------------------
// RMSNorm (layers/mla.py:46-58): x * rsqrt(mean(x^2) + 1e-6) * w ; optional AdaLN modulation
// (layers/transformer.py:64-68): y * (1 + scale) + shift, with mod = {1+scale [H], shift [H]} rows.
template <int NT>
EDTTS_DEV void rms_norm_tile(const f4 (&x)[NT][2], const float* __restrict__ w, const float* __restrict__ mod, int g,
                             f4 (&y)[NT][2]) {
  constexpr int N = NT * 16;
  float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    ss0 += hsum(x[t][0] * x[t][0]);
    ss1 += hsum(x[t][1] * x[t][1]);
  }
  const float r0 = rsqrtf(group_sum(ss0) * (1.0f / N) + 1e-6f);
  const float r1 = rsqrtf(group_sum(ss1) * (1.0f / N) + 1e-6f);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f4 wv = ldg4(w + 16 * t + 4 * g);
    f4 a = x[t][0] * r0 * wv, b = x[t][1] * r1 * wv;
    if (mod != nullptr) {
      const f4 sc = ldg4(mod + 16 * t + 4 * g), sh = ldg4(mod + N + 16 * t + 4 * g);
      a = a * sc + sh;
      b = b * sc + sh;
    }
    y[t][0] = a;
    y[t][1] = b;
  }
}

// LayerNorm(eps 1e-5, affine) (models/decoder.py:59,108)
template <int NT>
EDTTS_DEV void layer_norm_tile(const f4 (&x)[NT][2], const float* __restrict__ w, const float* __restrict__ b, int g,
                               f4 (&y)[NT][2]) {
  constexpr int N = NT * 16;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s0 += hsum(x[t][0]);
    s1 += hsum(x[t][1]);
  }
  const float mu0 = group_sum(s0) * (1.0f / N), mu1 = group_sum(s1) * (1.0f / N);
  float v0 = 0.f, v1 = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f4 d0 = x[t][0] - mu0, d1 = x[t][1] - mu1;
    v0 += hsum(d0 * d0);
    v1 += hsum(d1 * d1);
  }
  const float r0 = rsqrtf(group_sum(v0) * (1.0f / N) + 1e-5f), r1 = rsqrtf(group_sum(v1) * (1.0f / N) + 1e-5f);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f4 wv = ldg4(w + 16 * t + 4 * g), bv = ldg4(b + 16 * t + 4 * g);
    y[t][0] = (x[t][0] - mu0) * r0 * wv + bv;
    y[t][1] = (x[t][1] - mu1) * r1 * wv + bv;
  }
}

EDTTS_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// g * sigmoid(g); v_rcp_f32 (1 ulp) instead of the 9-instruction IEEE divide -- relative error ~1e-7, far inside the parity budget
EDTTS_DEV float silu(float g) { return g * __builtin_amdgcn_rcpf(1.0f + __expf(-g)); }

// ---------------------------------------------------------------------------------------------------------
// Multi-head attention for one wave's 32 query frames, fused with the output projection:
//   h += W_o . concat_heads( softmax(q k^T / sqrt(d) [band mask]) v )
// q comes from global memory (SELF: the q rows written by the previous kernel) or from this wave's LDS tile
// (cross-attention); K is row-major [key][H]; V is stored transposed [feature][key] so that both MFMA A operands
// are 16-byte loads.  Scores live only in registers: S^T tile = K Q^T (keys on MFMA rows, queries on lanes), an
// online softmax over 16-key tiles, and P^T (the C/D registers) is directly the B operand of O^T = V^T P^T.
// The per-head output tiles O^T[dt] then feed the k-major projection weights from the fragment ring.
//   SELF : keys are frames of the same utterance, band |i-j| <= window (layers/attention.py:27-30,108-112)
//   !SELF: keys are the S context tokens, no mask (layers/mla.py:158-179)
// ---------------------------------------------------------------------------------------------------------
// weight stream of the layer / prologue kernels: a per-wave register ring of HT/2 fragments straight from L2.
// (A block-shared LDS ring -- each fragment fetched once per block, ds_read_b128 to the MFMA -- was built and measured:
// 6 % slower at B=256, T=512, because its per-phase block barrier costs more than the 4x L2 traffic it saves; see
// DESIGN.md "What was tried".)
#ifndef EDTTS_RING_DIV
#define EDTTS_RING_DIV 1
#endif
template <class C> using WStream = FragRing<(C::HT / EDTTS_RING_DIV >= 2 ? C::HT / EDTTS_RING_DIV : 2)>;

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

template <class C, bool SELF, class QLoad>
EDTTS_DEV void attention_fused(QLoad&& qload, const float* __restrict__ Kb, const float* __restrict__ VTb, int ldv,
                               int nkeys, int window, int m0, int lane, WStream<C>& ring, f4 (&h)[C::HT][2]) {
  constexpr int DH = C::DH, DFULL = C::DFULL, DREM = C::DREM, DT = C::DT, H = C::H, CH = kChunk;
  const int fq = lane & 15, g = lane >> 4;
  // softmax in base 2: p = 2^((s - m) * c), c = log2(e) / sqrt(d)
  const float c2 = 1.4426950408889634f * rsqrtf((float)DH);
  const float NEG_INF = -__builtin_inff();
  int kt_lo, kt_hi;
  if (SELF && window >= 0) {
    const int lo = m0 - window;
    kt_lo = (lo > 0 ? lo : 0) >> 4;
    const int hi = m0 + kWaveFrames - 1 + window;  // last key any query of this wave may see
    const int last = (hi < nkeys - 1 ? hi : nkeys - 1);
    kt_hi = (last >> 4) + 1;
  } else {
    kt_lo = 0;
    kt_hi = (nkeys + 15) >> 4;
  }
  const int nchunk = (kt_hi - kt_lo + CH - 1) / CH;
  const int klim = (kt_hi << 4) < nkeys ? (kt_hi << 4) : nkeys;  // keys >= klim are never valid
  // per-lane band limits on d = key - query:  lo_d <= d <= hi_d   (one unsigned compare per score)
  int lo_d[2], span[2];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft) {
    const int qi = m0 + 16 * ft + fq;
    int lo = -(1 << 28), hi = klim - 1 - qi;
    if (SELF && window >= 0) {
      lo = -window;
      hi = hi < window ? hi : window;
    }
    lo_d[ft] = lo;
    span[ft] = hi - lo;  // negative -> nothing valid
  }

  // loads of one chunk's K fragments (tile index clamped: tiles past kt_hi are fully masked by klim)
  auto load_k = [&](int hd, int c, KVFrag<C>& f) {
    c = c < nchunk ? c : nchunk - 1;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      int kt = kt_lo + c * CH + t;
      kt = kt < kt_hi ? kt : kt_hi - 1;
      const float* kp = Kb + (size_t)((kt << 4) + fq) * H + hd * DH;
#pragma unroll
      for (int a = 0; a < DFULL; ++a) f.ka[t][a] = ldg4(kp + 16 * a + 4 * g);
      if (DREM) f.kr[t] = ldg2(kp + 16 * DFULL + 2 * g);
    }
  };
  auto load_v = [&](int hd, int c, VFrag<C>& f) {
    c = c < nchunk ? c : nchunk - 1;
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      int kt = kt_lo + c * CH + t;
      kt = kt < kt_hi ? kt : kt_hi - 1;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) f.v[t][dt] = ldg4(VTb + (size_t)(hd * DH + 16 * dt + fq) * ldv + (kt << 4) + 4 * g);
    }
  };
  // Mask of chunk c as the INITIAL accumulator of its K Q^T product: 0 where the key is visible, -inf elsewhere
  // (-inf + finite products = -inf).  Computed before the MFMAs are issued, so no VALU work sits between the MFMA
  // results and the softmax; interior chunks (every key inside the band of every query of the wave and below klim: 3 of
  // the 5 chunks at window 64, all of the cross-attention when S % 32 == 0) take the wave-uniform zero path.
  auto mask_init = [&](int c, f4 (&S)[CH][2]) {
    c = c < nchunk ? c : nchunk - 1;
    const int k0 = (kt_lo + c * CH) << 4, k1 = k0 + 16 * CH - 1;
    bool full = k1 < klim && (kt_lo + (c + 1) * CH) <= kt_hi;
    if (SELF && window >= 0) full = full && (k1 - m0 <= window) && (k0 - (m0 + kWaveFrames - 1) >= -window);
#pragma unroll
    for (int t = 0; t < CH; ++t) S[t][0] = S[t][1] = splat(0.f);
    if (!full) {
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const int d0 = k0 + 4 * g - (m0 + 16 * ft + fq) - lo_d[ft];  // (key - query - lo_d) of r = 0, tile 0
        const unsigned sp = span[ft] >= 0 ? (unsigned)span[ft] : 0u;
        const int bias = span[ft] >= 0 ? 0 : (1 << 30);               // nothing valid for this query
#pragma unroll
        for (int t = 0; t < CH; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) S[t][ft][r] = (unsigned)(d0 + bias + 16 * t + r) <= sp ? 0.f : NEG_INF;
      }
    }
  };
  // S^T chunk += K Q^T for both query tiles, accumulators interleaved (2*CH independent chains)
  auto qk = [&](const KVFrag<C>& f, const f4 (&qa)[2][DFULL > 0 ? DFULL : 1], const f2 (&qr)[2], f4 (&S)[CH][2]) {
#pragma unroll
    for (int a = 0; a < DFULL; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          S[t][0] = EDTTS_MFMA(f.ka[t][a][b], qa[0][a][b], S[t][0]);
          S[t][1] = EDTTS_MFMA(f.ka[t][a][b], qa[1][a][b], S[t][1]);
        }
    if (DREM) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < CH; ++t) {
          S[t][0] = EDTTS_MFMA(f.kr[t][b], qr[0][b], S[t][0]);
          S[t][1] = EDTTS_MFMA(f.kr[t][b], qr[1][b], S[t][1]);
        }
    }
  };

  // q fragments (B operand: lane (fq,g) holds q[query][hd*DH + 16a + 4g + b]) and the K fragments of the first two chunks
  // of a head.  They are fetched while the PREVIOUS head's projection phases run, so no head starts on an exposed load.
  f4 qa_n[2][DFULL > 0 ? DFULL : 1];
  f2 qr_n[2];
  KVFrag<C> KA, KB;
  VFrag<C> VA, VB;
  auto prefetch_head = [&](int hd) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
      for (int a = 0; a < DFULL; ++a) qa_n[ft][a] = qload.q4(ft, hd * DH + 16 * a + 4 * g);
      if (DREM) qr_n[ft] = qload.q2(ft, hd * DH + 16 * DFULL + 2 * g);
    }
    load_k(hd, 0, KA);
    load_k(hd, 1, KB);
    load_v(hd, 0, VA);
    __builtin_amdgcn_sched_barrier(0);
  };
  prefetch_head(0);

  for (int hd = 0; hd < C::HEADS; ++hd) {
    // pre-scale q by log2(e)/sqrt(d): the scores then come out of the MFMA in the exp2 domain
    f4 qa[2][DFULL > 0 ? DFULL : 1];
    f2 qr[2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
      for (int a = 0; a < DFULL; ++a) qa[ft][a] = qa_n[ft][a] * c2;
      if (DREM) qr[ft] = qr_n[ft] * c2;
    }
    float mrun[2] = {-1e30f, -1e30f};
    f4 lvec[2] = {splat(0.f), splat(0.f)};  // per-lane partial row sums (reduced over r and the lane groups at the end)
    f4 O[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) O[dt][0] = O[dt][1] = splat(0.f);

    // One step: finish chunk c (softmax + P V) while the scores of chunk c+1 are produced.
    //   Sc   : scores of chunk c (complete)          Sn : receives the scores of chunk c+1 (its mask is already in it)
    //   Kuse : K fragments of chunk c+1 (loaded one step ago)     Kld : receives the K fragments of chunk c+2
    //   Vuse : V^T fragments of chunk c (loaded one step ago)     Vld : receives the V^T fragments of chunk c+1
    // Every load is consumed in a LATER step (loop-carried), which is what keeps hipcc from sinking it next to its use:
    // a V load issued and used within the same step was moved across the rescale branch right in front of the P V MFMAs.
    // The caller alternates the S / K / V buffers, so nothing is copied between steps.
    auto step = [&](auto has_next, int c, f4 (&Sc)[CH][2], f4 (&Sn)[CH][2], const KVFrag<C>& Kuse, KVFrag<C>& Kld,
                    const VFrag<C>& Vuse, VFrag<C>& Vld) {
      if (decltype(has_next)::value) {
        load_v(hd, c + 1, Vld);
        load_k(hd, c + 2, Kld);
      }
      __builtin_amdgcn_sched_barrier(0);
      // scores of the NEXT chunk: independent MFMA work that overlaps this chunk's softmax VALU (same scheduling region)
      if (decltype(has_next)::value) qk(Kuse, qa, qr, Sn);
      // Online softmax with a DEFERRED running maximum: fp32 accumulators have ~2^127 of headroom, so the reference point
      // m only has to move when a chunk exceeds it by more than 2^kDefer; until then P = 2^(s - m) <= 2^kDefer and neither O
      // nor the row sums need rescaling (softmax is invariant to m).  The wave-uniform branch is taken on the first chunk
      // (m starts at -1e30) and then only when some row's scores jump by > kDefer octaves; it removes the per-chunk
      // read-modify-write of the O accumulators and the cross-lane max from the common path.
      float mxl[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        f4 mv = Sc[0][ft];
#pragma unroll
        for (int t = 1; t < CH; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mv[r], Sc[t][ft][r]);
        mxl[ft] = hmax(mv);
      }
      if (__any((mxl[0] > mrun[0] + kDefer) || (mxl[1] > mrun[1] + kDefer))) {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          const float mnew = fmaxf(mrun[ft], group_max(mxl[ft]));  // identical on the 4 lanes of a row
          const float alpha = fast_exp2(mrun[ft] - mnew);           // mrun starts finite (-1e30): never NaN
          mrun[ft] = mnew;
          lvec[ft] *= alpha;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) O[dt][ft] *= alpha;
        }
      }
      f4 P[CH][2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
#pragma unroll
        for (int t = 0; t < CH; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) P[t][ft][r] = fast_exp2(Sc[t][ft][r] - mrun[ft]);
          lvec[ft] += P[t][ft];
        }
      }
      // O^T += V^T P^T : DT x 2 independent accumulators, r outermost
#pragma unroll
      for (int t = 0; t < CH; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            O[dt][0] = EDTTS_MFMA(Vuse.v[t][dt][r], P[t][0][r], O[dt][0]);
            O[dt][1] = EDTTS_MFMA(Vuse.v[t][dt][r], P[t][1][r], O[dt][1]);
          }
    };

    f4 SA[CH][2], SB[CH][2];
    mask_init(0, SA);
    qk(KA, qa, qr, SA);  // scores of chunk 0; KB holds chunk 1
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    int c = 0;
    for (; c + 2 < nchunk; c += 2) {
      mask_init(c + 1, SB);
      step(Yes{}, c, SA, SB, KB, KA, VA, VB);      // finishes chunk c; scores of c+1 -> SB (from KB); loads K(c+2) -> KA, V(c+1) -> VB
      mask_init(c + 2, SA);
      step(Yes{}, c + 1, SB, SA, KA, KB, VB, VA);  // finishes chunk c+1; scores of c+2 -> SA (from KA); loads K(c+3) -> KB, V(c+2) -> VA
    }
    if (nchunk - c == 2) {  // two chunks left: scores of c are in SA, V(c) in VA
      mask_init(c + 1, SB);
      step(Yes{}, c, SA, SB, KB, KA, VA, VB);
      step(No{}, c + 1, SB, SA, KA, KB, VB, VA);
    } else {                // one chunk left
      step(No{}, c, SA, SB, KB, KA, VA, VB);
    }
    // ---- normalise and project: h[nt] += Wo[:, head features] . O ------------------------------------------
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      const float lt = group_sum(hsum(lvec[ft]));
      const float inv = lt > 0.f ? 1.0f / lt : 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) O[dt][ft] *= inv;
    }
    if (hd + 1 < C::HEADS) prefetch_head(hd + 1);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) ktile_phase<C::HT>(ring, O[dt][0], O[dt][1], h);
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

}  // namespace edtts
