// edtts_bf16.h -- bf16 instance of the decoder kernels (BASELINE config 3: hidden=256, heads=8, head_dim=32).
// Included by edtts_kernels.hip after KArgs / tail_apply / wave_tile are defined.
//
// Arithmetic: every contraction runs on v_mfma_f32_16x16x32_bf16 (bf16 operands, fp32 accumulate); the residual stream h, all
// norms, the softmax (max, exp2, sums) and the sampler update stay fp32.  Weights are rounded to bf16 once (edtts_pack_weights),
// activations are rounded to bf16 where they enter an MFMA (normalised tiles, q / k / v, probabilities, attention outputs, SwiGLU
// outputs) -- the reference's own AMP precedent (utils/speed_utils.py:70, train_v2.py:290) rounds the same tensors.
//
// Data model ("frame on the lane", as in the fp32 kernels):
//   a wave owns 32 frames = two 16-frame tiles ft; C/D of the MFMA: lane (fq = lane & 15, g = lane >> 4) holds
//   out[feature 16 nt + 4 g + r][frame fq], r = 0..3.  The B operand of the 16x16x32 MFMA wants, per lane, 8 bf16 = the k's
//   {8 g + j}.  MFMA contraction order is free, so the 32 features of a k-tile are assigned to the (g, j) slots as
//       slot(g, j) = 4 g + j            (j < 4)      <- C/D registers of n-tile 2 kt
//                    16 + 4 g + (j - 4) (j >= 4)     <- C/D registers of n-tile 2 kt + 1
//   and the weights are packed with the same assignment: the C/D registers of two consecutive n-tiles, converted pairwise with
//   v_cvt_pk_bf16_f32, ARE the B operand of the next GEMM's k-tile.  Nothing leaves registers inside a layer except q / k / v^T
//   (needed by other waves) and the layer-boundary residual.
//   Memory images follow from that.  q is row-major [frame][H] bf16 with the 32 features of a head stored in slot order (one
//   16-byte load per lane is an MFMA operand; q.k is order-agnostic as long as both use the same order).  K (self and the cross
//   cache) is TILE-CONTIGUOUS: [utterance][head][key tile][16 keys][32 d-slots], v^T likewise: [utterance][head][32-key chunk]
//   [d-tile][16 d][32 key-slots] with the keys of a chunk in slot order (position 8 g' + 4 t + r <-> key 16 t + 4 g' + r), which is
//   what a lane of the SWAPPED product (activations as A, weights as B: C/D = [frame][feature]) holds.  Every K / v^T operand
//   tile is one contiguous KiB: one 16-byte store per lane when written, one fully coalesced load per wave when read.
#pragma once

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define EDTTS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

#ifndef EDTTS_STAMP_HEAD
#define EDTTS_STAMP_HEAD 0     // head (pair) whose attention steps carry the fine-grained stamps (EDTTS_STAMPS builds)
#endif
#ifdef EDTTS_STAMPS
#ifndef EDTTS_STAMP_THREAD
#define EDTTS_STAMP_THREAD 0   // first lane of the stamped wave of block 0 (192: wave 3 = frames 96..127, an interior tile)
#endif
#define STAMP16(i) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == EDTTS_STAMP_THREAD) a.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPX16(p, i) do { if ((p) && blockIdx.x == 0 && threadIdx.x == EDTTS_STAMP_THREAD) (p)[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP16(i) do { } while (0)
#define STAMPX16(p, i) do { } while (0)
#endif

namespace edtts16 {
using namespace edtts;

// NF_ = 16-frame tiles per wave.  NF = 2: four waves per block (one per SIMD, up to 512 registers each).  NF = 1: EIGHT waves per
// block, two per SIMD (<= 256 registers each): one wave's waits and softmax arithmetic run under the other's MFMAs -- with one
// wave per SIMD 27 % of the cycles were s_waitcnt time and 24 % issue stalls -- at the price of LDS bandwidth (every wave reads
// every weight fragment: 8 KiB per fragment and block).
template <int H_, int HEADS_, int MEL_, int NF_ = 2>
struct Cfg16 {
  static constexpr int H = H_, HEADS = HEADS_, MEL = MEL_, NF = NF_;
  static constexpr int WF = 16 * NF;            // frames per wave
  static constexpr int DH = H / HEADS;          // head dim: one k-tile
  static constexpr int HT = H / 16;             // n-tiles of the hidden dim
  static constexpr int KT = H / 32;             // k-tiles of the hidden dim (= heads)
  static constexpr int MT = MEL / 16;           // n-tiles of the mel dim
  static constexpr int MKT = (MEL + 31) / 32;   // k-tiles of the mel dim (zero padded)
  static constexpr int MTP = (MT + 1) / 2;      // n-tile PAIRS of the mel dim (the last one may be half empty)
  static constexpr int R = H / 2;
  static constexpr int VR = H;                  // rows of a v^T buffer
#ifndef EDTTS16_W1
#define EDTTS16_W1 4   // waves per block of the NF = 1 variant: 8 = one block per CU, two waves per SIMD in lockstep; 4 = two independent blocks per CU
#endif
  static constexpr int WAVES = NF == 1 ? EDTTS16_W1 : 4, THREADS = 64 * WAVES;
  // (the eight-wave NF = 1 block is the two-waves-per-SIMD experiment: 256 registers each; the four-wave one -- the run-time choice
  // for small grids -- keeps the whole register file: under a 256-register cap it spills to scratch)
  static constexpr int MIN_WAVES_PER_SIMD = (NF == 1 && WAVES == 8) ? 2 : 1;
  // Weight stream: every GEMM unit of the kernels (an n-tile pair over all k-tiles, or one k-tile over all n-tiles) consumes
  // exactly PH = HT fragments (1 KiB each) -- one PHASE.  The block shares one LDS ring of NS phase slots (see LdsRing).
  static constexpr int PH = HT;                 // fragments per phase
#ifndef EDTTS16_QLDS
#define EDTTS16_QLDS 1   // cross-attention q of all heads parked in LDS (1) or in the wave's rows of the q buffer (0)
#endif
  // cross-attention q operands of a wave's frames, all heads: [head][frame tile][lane] x 16 B -- written once after the q
  // projection, read back head by head (64 registers less during the cross-attention; the ring gives up one slot for it)
  // (NF = 4: 128 KiB would not fit beside the ring; the wave parks them in its own rows of the layer's q buffer instead, which the
  // self-attention has finished with by then)
  static constexpr bool QLDS = EDTTS16_QLDS && NF <= 2;
#ifndef EDTTS16_FRAG_GROUP
#define EDTTS16_FRAG_GROUP 4
#endif
  static constexpr int FRAG_GROUP = EDTTS16_FRAG_GROUP;  // NF = 4: weight fragments read from the ring per group (see gemm16_pair)
  static constexpr int QLDS_BYTES = QLDS ? WAVES * HEADS * NF * 1024 : 0;
  static constexpr int NS_MAX = (NF == 1 && WAVES == 4) ? 4 : 6;  // ring slots (phases): NS - 1 phases are in flight ahead of the consumers
  static constexpr int UPB_FLOATS = 8 * H;          // room for the FFN up bias at the largest ffn_mult (4): 2 * 4 * H floats
  static constexpr int PARAM_FLOATS = UPB_FLOATS + MEL;  // FFN up bias (stream order) + out_proj bias, staged in LDS (see k_layer16)
  static constexpr int NS_FIT = (160 * 1024 - PARAM_FLOATS * 4 - QLDS_BYTES) / (PH * 1024);
  static constexpr int NS = NS_FIT < NS_MAX ? NS_FIT : NS_MAX;
  static constexpr int LDS_BYTES = NS * PH * 1024 + PARAM_FLOATS * 4 + QLDS_BYTES;
  static_assert(NS >= 3, "ring depth");
  static_assert(DH == 32, "the bf16 instance is built for head_dim 32 (one MFMA k-tile per head)");
  static_assert(NF == 1 || NF == 2 || NF == 4, "frame tiles per wave");
  static_assert(H % 64 == 0 && MEL % 16 == 0 && PH % WAVES == 0 && LDS_BYTES <= 160 * 1024, "dims vs ring");
};

EDTTS_DEV bf8 as_bf8(f4 v) { return __builtin_bit_cast(bf8, v); }
EDTTS_DEV f4 as_f4(bf8 v) { return __builtin_bit_cast(f4, v); }
// C/D registers of two consecutive n-tiles -> the 8 bf16 of this lane's k-slots (v_cvt_pk_bf16_f32, round to nearest even)
EDTTS_DEV bf8 pack8(f4 a, f4 b) {
  const bf2 p0 = __builtin_convertvector(f2v{a[0], a[1]}, bf2), p1 = __builtin_convertvector(f2v{a[2], a[3]}, bf2);
  const bf2 p2 = __builtin_convertvector(f2v{b[0], b[1]}, bf2), p3 = __builtin_convertvector(f2v{b[2], b[3]}, bf2);
  return bf8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}
typedef float f2s __attribute__((ext_vector_type(2)));
// four C/D registers -> 4 bf16 (8 bytes)
EDTTS_DEV f2s pack4(f4 a) {
  const bf2 p0 = __builtin_convertvector(f2v{a[0], a[1]}, bf2), p1 = __builtin_convertvector(f2v{a[2], a[3]}, bf2);
  typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f2s, bf4{p0[0], p0[1], p1[0], p1[1]});
}
EDTTS_DEV bf8 ldg_bf8(const __bf16* base, unsigned byte_off) {
  return *reinterpret_cast<const bf8*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Private images.  The residual tile h (fp32) and the q rows (bf16) of a wave's frames are written and read back only by the wave
// that owns those frames, lane for lane -- so inside the tile's span of the buffer they are stored as the REGISTER IMAGE,
// [n-tile | head][frame tile][lane] x 16 bytes: every access of a wave is one contiguous KiB instead of 16 row segments of 64 B.
// (EDTTS16_IMG 0: row-major [frame][H], as the fp32 kernels keep them.)  tile0 = ((b * Tp + m0) * H: the span starts where the
// row-major rows of the tile would.
#ifndef EDTTS16_IMG
#define EDTTS16_IMG 1
#endif
template <class C>
EDTTS_DEV size_t h_at(size_t tile0, int lane, int nt, int ft) {  // floats
  if (EDTTS16_IMG) return tile0 + (size_t)(nt * C::NF + ft) * 256 + lane * 4;
  return tile0 + (size_t)((lane & 15) + 16 * ft) * C::H + 16 * nt + 4 * (lane >> 4);
}
template <class C>
EDTTS_DEV size_t q_at(size_t tile0, int lane, int hd, int ft) {  // bf16 elements
  if (EDTTS16_IMG) return tile0 + (size_t)(hd * C::NF + ft) * 512 + lane * 8;
  return tile0 + (size_t)((lane & 15) + 16 * ft) * C::H + hd * C::DH + 8 * (lane >> 4);
}
// (the same as a per-lane byte offset + a wave-uniform byte offset, for buffer loads / stores from the tile's base)
template <class C> EDTTS_DEV unsigned q_voff(int lane) { return EDTTS16_IMG ? lane * 16u : (unsigned)((lane & 15) * C::H + 8 * (lane >> 4)) * 2u; }
template <class C> EDTTS_DEV unsigned q_soff(int hd, int ft) { return EDTTS16_IMG ? (unsigned)(hd * C::NF + ft) * 1024u : (unsigned)(ft * 16 * C::H + hd * C::DH) * 2u; }

// 64-frame waves (NF = 4) keep the residual tile in the 256 AGPRs: it is written by the MFMAs of ktile16 (which name the class
// themselves) and by acc_put, and read by the VALU through acc_get -- an opaque read, so that hipcc cannot keep a second, VGPR copy
// of the tile alive from one reader to the next (it did: 460 spilled registers).  NF <= 2: plain values.
template <bool PIN>
EDTTS_DEV f4 acc_get(const f4& x) {
  if constexpr (!PIN) return x;
  else {
    f4 r = x;
    asm volatile("" : "+v"(r));
    return r;
  }
}
template <bool PIN>
EDTTS_DEV void acc_put(f4& x, f4 v) {
  x = v;
  if constexpr (PIN) asm volatile("" : "+a"(x));
}

// ---------------------------------------------------------------------------------------------------------
// Block-shared weight ring in LDS.  At bf16 rate a 1-KiB fragment feeds only 2 MFMAs = 32 cycles of one wave: four waves
// streaming their own copies would ask the CU's vector L1 for 128 B/clk (it delivers 64) and keep ~1.5 MB per wave and layer
// in flight from L2.  Instead each fragment is fetched ONCE per block by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land
// lane-contiguous = exactly the MFMA operand image, no registers involved) and read by the four waves with ds_read_b128.
// All four waves run the same program, so the ring needs no flags: the stream is cut into PHASES of PH fragments;
//   acquire():  s_waitcnt vmcnt(...)   this wave's share of the phase has landed (younger DMAs may stay in flight)
//               s_barrier              => every wave's share has landed, and every wave is done reading the previous phase
//               issue this wave's share of phase p + NS - 1 into the slot the previous phase occupied
// i.e. one barrier per 2 * PH MFMAs, NS - 1 phases (80 KiB at H = 256) of prefetch depth.  The DMAs are always issued (the blob
// carries NS phases of slack behind the last fragment) so that the counted wait below stays valid up to the last phase.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <class C>
struct LdsRing {
  static constexpr int PH = C::PH, NS = C::NS, SHARE = C::PH / C::WAVES;
  f4* lds;           // ring base (wave-uniform)
  const f4* src;     // this lane's pointer to fragment 0 of the kernel's stream
  int next;          // phase the next acquire() hands out
  int wave, lane;
  EDTTS_DEV void issue(int phase) {
    const int slot = phase % NS;
#pragma unroll
    for (int i = 0; i < SHARE; ++i) {
      const int f = SHARE * wave + i;
      __builtin_amdgcn_global_load_lds(src + ((size_t)phase * PH + f) * 64, (lds_ptr_t)(lds + (slot * PH + f) * 64), 16, 0, 0);
    }
  }
  EDTTS_DEV void start(const float* stream, f4* lds_base, int wave_, int lane_) {
    lds = lds_base; wave = wave_; lane = lane_; next = 0;
    src = reinterpret_cast<const f4*>(stream) + lane;
    for (int p = 0; p < NS - 1; ++p) issue(p);
  }
  // fragments of the next phase: fragment i of this lane at ptr[i * 64]
  EDTTS_DEV const f4* acquire() {
    static_assert(SHARE * (NS - 2) <= 63, "vmcnt immediate");
    // all but the SHARE * (NS - 2) youngest vector-memory operations done => the DMAs of phase `next` (and everything older) landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SHARE * (NS - 2)) : "memory");
#ifndef EDTTS16_ABLATE_BARRIER  // timing ablations only (results wrong by construction)
    __builtin_amdgcn_s_barrier();
#endif
#ifndef EDTTS16_ABLATE_DMA
    issue(next + NS - 1);
#endif
    const f4* p = lds + (next % NS) * PH * 64 + lane;
    ++next;
    return p;
  }
  EDTTS_DEV void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }  // no DMA may outlive the block's LDS allocation
};
template <class C> using Ring16 = LdsRing<C>;

// Fragment i of an acquired phase.  The slot base goes through an opaque register so that hipcc addresses the N reads of a phase
// as ONE base + immediate offsets (ds_read_b128 ... offset:1024 i) instead of materialising N addresses with N v_add_u32.
typedef __attribute__((address_space(3))) const f4* lds_cf4_t;
EDTTS_DEV lds_cf4_t phase_base(const f4* fr) {
  lds_cf4_t p = (lds_cf4_t)fr;
  asm volatile("" : "+v"(p));
  return p;
}
// two n-tiles at once from a phase that interleaves their fragments per k-tile ([kt][tile a | tile b]).
// SWAP: activations as the A operand, weights as B -> C/D = [frame][feature] (used for v^T).
template <int KT, bool SWAP, class C>
EDTTS_DEV void gemm16_pair(LdsRing<C>& ring, const bf8 (&in)[KT][C::NF], f4 (&a)[C::NF], f4 (&b)[C::NF]) {
  static_assert(2 * KT == C::PH, "an n-tile pair over all k-tiles is one phase");
  const f4* fr = ring.acquire();
  // all fragments of the phase are requested from LDS up front (in-order returns: the MFMAs wait with a counted lgkmcnt each),
  // instead of read / wait / use per fragment pair, which exposes the LDS latency KT times per phase.  (Requesting the NEXT
  // phase's fragments right after this phase's MFMAs -- to take the barrier and the LDS round trip off the critical path -- was
  // measured 18 % slower: the 64 registers held across the phase boundary bring the spills back.)
  const lds_cf4_t fb0 = phase_base(fr);
  if constexpr (C::NF > 2) {
    // 64 frames per wave: the residual tile alone is 256 registers, and a fragment feeds four MFMAs (64 cycles) -- the fragments are
    // read in groups of FG, one group ahead of the MFMAs that consume them, instead of all up front (32 registers instead of 64)
    constexpr int FG = C::FRAG_GROUP, NG = 2 * KT / FG;
    static_assert(FG % 2 == 0 && (2 * KT) % FG == 0, "fragment groups");
    f4 fg[2][FG];
#pragma unroll
    for (int i = 0; i < FG; ++i) fg[0][i] = fb0[i * 64];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG)
#pragma unroll
        for (int i = 0; i < FG; ++i) fg[(gi + 1) & 1][i] = fb0[((gi + 1) * FG + i) * 64];
#pragma unroll
      for (int kk = 0; kk < FG / 2; ++kk) {
        const int kt = gi * (FG / 2) + kk;
        const bf8 fa = as_bf8(fg[gi & 1][2 * kk]), fb = as_bf8(fg[gi & 1][2 * kk + 1]);
#pragma unroll
        for (int ft = 0; ft < C::NF; ++ft) {
          a[ft] = SWAP ? EDTTS_MFMA16(in[kt][ft], fa, a[ft]) : EDTTS_MFMA16(fa, in[kt][ft], a[ft]);
          b[ft] = SWAP ? EDTTS_MFMA16(in[kt][ft], fb, b[ft]) : EDTTS_MFMA16(fb, in[kt][ft], b[ft]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    f4 fg[2 * KT];
#pragma unroll
    for (int i = 0; i < 2 * KT; ++i) fg[i] = fb0[i * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const bf8 fa = as_bf8(fg[2 * kt]), fb = as_bf8(fg[2 * kt + 1]);
#pragma unroll
      for (int ft = 0; ft < C::NF; ++ft) {
        a[ft] = SWAP ? EDTTS_MFMA16(in[kt][ft], fa, a[ft]) : EDTTS_MFMA16(fa, in[kt][ft], a[ft]);
        b[ft] = SWAP ? EDTTS_MFMA16(in[kt][ft], fb, b[ft]) : EDTTS_MFMA16(fb, in[kt][ft], b[ft]);
      }
    }
  }
}
// acc[nt] += frag(nt) * in  for one k-tile of a k-major packed matrix (NT fragments = one phase)
template <int NT, class C>
EDTTS_DEV void ktile16(LdsRing<C>& ring, const bf8 (&in)[C::NF], f4 (&acc)[NT][C::NF]) {
  static_assert(NT == C::PH, "one k-tile over all n-tiles is one phase");
  const f4* fr = ring.acquire();
  const lds_cf4_t fb0 = phase_base(fr);
  if constexpr (C::NF > 2) {  // (fragment groups: see gemm16_pair)
    constexpr int FG = C::FRAG_GROUP, NG = NT / FG;
    static_assert(NT % FG == 0, "fragment groups");
    f4 fg[2][FG];
#pragma unroll
    for (int i = 0; i < FG; ++i) fg[0][i] = fb0[i * 64];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG)
#pragma unroll
        for (int i = 0; i < FG; ++i) fg[(gi + 1) & 1][i] = fb0[((gi + 1) * FG + i) * 64];
      // The residual tile is 256 registers = the whole accumulator half of the register file.  hipcc either keeps all MFMA results
      // in AGPRs (then the 32 accumulators of the other GEMMs do not fit beside it) or none (-amdgpu-mfma-vgpr-form: then the tile
      // competes with everything else for the 256 VGPRs); these MFMAs therefore name their accumulator class themselves.
      static_assert(FG == 4 && C::NF == 4, "operand list of the block below");
      const int n0 = gi * FG;
      asm volatile(
          "s_nop 1\n"
          "v_mfma_f32_16x16x32_bf16 %0, %16, %20, %0\n"
          "v_mfma_f32_16x16x32_bf16 %1, %16, %21, %1\n"
          "v_mfma_f32_16x16x32_bf16 %2, %16, %22, %2\n"
          "v_mfma_f32_16x16x32_bf16 %3, %16, %23, %3\n"
          "v_mfma_f32_16x16x32_bf16 %4, %17, %20, %4\n"
          "v_mfma_f32_16x16x32_bf16 %5, %17, %21, %5\n"
          "v_mfma_f32_16x16x32_bf16 %6, %17, %22, %6\n"
          "v_mfma_f32_16x16x32_bf16 %7, %17, %23, %7\n"
          "v_mfma_f32_16x16x32_bf16 %8, %18, %20, %8\n"
          "v_mfma_f32_16x16x32_bf16 %9, %18, %21, %9\n"
          "v_mfma_f32_16x16x32_bf16 %10, %18, %22, %10\n"
          "v_mfma_f32_16x16x32_bf16 %11, %18, %23, %11\n"
          "v_mfma_f32_16x16x32_bf16 %12, %19, %20, %12\n"
          "v_mfma_f32_16x16x32_bf16 %13, %19, %21, %13\n"
          "v_mfma_f32_16x16x32_bf16 %14, %19, %22, %14\n"
          "v_mfma_f32_16x16x32_bf16 %15, %19, %23, %15\n"
          : "+a"(acc[n0][0]), "+a"(acc[n0][1]), "+a"(acc[n0][2]), "+a"(acc[n0][3]), "+a"(acc[n0 + 1][0]), "+a"(acc[n0 + 1][1]),
            "+a"(acc[n0 + 1][2]), "+a"(acc[n0 + 1][3]), "+a"(acc[n0 + 2][0]), "+a"(acc[n0 + 2][1]), "+a"(acc[n0 + 2][2]),
            "+a"(acc[n0 + 2][3]), "+a"(acc[n0 + 3][0]), "+a"(acc[n0 + 3][1]), "+a"(acc[n0 + 3][2]), "+a"(acc[n0 + 3][3])
          : "v"(fg[gi & 1][0]), "v"(fg[gi & 1][1]), "v"(fg[gi & 1][2]), "v"(fg[gi & 1][3]), "v"(in[0]), "v"(in[1]), "v"(in[2]),
            "v"(in[3]));
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the last results are 8 passes away: nothing the compiler places behind this block may read the tile earlier)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  } else {
    f4 fg[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) fg[i] = fb0[i * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf8 fa = as_bf8(fg[nt]);
#pragma unroll
      for (int ft = 0; ft < C::NF; ++ft) acc[nt][ft] = EDTTS_MFMA16(fa, in[ft], acc[nt][ft]);
    }
  }
}

// RMSNorm (+ optional AdaLN modulation) of the residual tile, straight into packed bf16 B operands
template <class C>
EDTTS_DEV void rms_norm_pack(const f4 (&x)[C::HT][C::NF], const float* __restrict__ w, const float* __restrict__ mod, int g,
                             bf8 (&y)[C::KT][C::NF]) {
  constexpr int NF = C::NF;
  constexpr bool PIN = NF > 2;
  float rs[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < C::HT; ++t) {
      const f4 xv = acc_get<PIN>(x[t][ft]);
      ss += hsum(xv * xv);
    }
    rs[ft] = rsqrtf(group_sum(ss) * (1.0f / C::H) + 1e-6f);
  }
  if constexpr (NF > 2) {
    // 64 frames per wave: 256 registers of residual + 128 of output leave no room for hipcc's habit of requesting all 48 parameter
    // vectors up front (192 registers: it spilled 340) -- the k-tiles are pinned in order, each requesting the next one's parameters
    f4 pw[2][2], psc[2][2], psh[2][2];
    auto request = [&](int kt, int slot) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * kt + u;
        pw[slot][u] = ldg4(w + 16 * t + 4 * g);
        if (mod != nullptr) {
          psc[slot][u] = ldg4(mod + 16 * t + 4 * g);
          psh[slot][u] = ldg4(mod + C::H + 16 * t + 4 * g);
        }
      }
    };
    request(0, 0);
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
      if (kt + 1 < C::KT) request(kt + 1, (kt + 1) & 1);
      f4 v[2][NF];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          f4 a = acc_get<PIN>(x[2 * kt + u][ft]) * rs[ft] * pw[kt & 1][u];
          if (mod != nullptr) a = a * psc[kt & 1][u] + psh[kt & 1][u];
          v[u][ft] = a;
        }
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        y[kt][ft] = pack8(v[0][ft], v[1][ft]);
        asm volatile("" : "+v"(y[kt][ft]));  // (ordered with the opaque reads of the next k-tile: the arithmetic stays here)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
#pragma unroll
  for (int kt = 0; kt < C::KT; ++kt) {
    f4 v[2][NF];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * kt + u;
      const f4 wv = ldg4(w + 16 * t + 4 * g);
      f4 sc = splat(1.f), sh = splat(0.f);
      if (mod != nullptr) {
        sc = ldg4(mod + 16 * t + 4 * g);
        sh = ldg4(mod + C::H + 16 * t + 4 * g);
      }
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        f4 a = x[t][ft] * rs[ft] * wv;
        if (mod != nullptr) a = a * sc + sh;
        v[u][ft] = a;
      }
    }
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) y[kt][ft] = pack8(v[0][ft], v[1][ft]);
  }
}
template <class C>
EDTTS_DEV void layer_norm_pack(const f4 (&x)[C::HT][C::NF], const float* __restrict__ w, const float* __restrict__ b, int g,
                               bf8 (&y)[C::KT][C::NF]) {
  constexpr int NF = C::NF;
  constexpr bool PIN = NF > 2;
  float mu[NF], rs[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < C::HT; ++t) s += hsum(acc_get<PIN>(x[t][ft]));
    mu[ft] = group_sum(s) * (1.0f / C::H);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < C::HT; ++t) {
      const f4 d = acc_get<PIN>(x[t][ft]) - mu[ft];
      v += hsum(d * d);
    }
    rs[ft] = rsqrtf(group_sum(v) * (1.0f / C::H) + 1e-5f);
  }
  if constexpr (NF > 2) {  // (k-tiles pinned in order, parameters one k-tile ahead: see rms_norm_pack)
    f4 pw[2][2], pbv[2][2];
    auto request = [&](int kt, int slot) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pw[slot][u] = ldg4(w + 16 * (2 * kt + u) + 4 * g);
        pbv[slot][u] = ldg4(b + 16 * (2 * kt + u) + 4 * g);
      }
    };
    request(0, 0);
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
      if (kt + 1 < C::KT) request(kt + 1, (kt + 1) & 1);
      f4 v[2][NF];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) v[u][ft] = (acc_get<PIN>(x[2 * kt + u][ft]) - mu[ft]) * rs[ft] * pw[kt & 1][u] + pbv[kt & 1][u];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        y[kt][ft] = pack8(v[0][ft], v[1][ft]);
        asm volatile("" : "+v"(y[kt][ft]));  // (ordered with the opaque reads of the next k-tile: the arithmetic stays here)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
#pragma unroll
  for (int kt = 0; kt < C::KT; ++kt) {
    f4 v[2][NF];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * kt + u;
      const f4 wv = ldg4(w + 16 * t + 4 * g), bv = ldg4(b + 16 * t + 4 * g);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) v[u][ft] = (x[t][ft] - mu[ft]) * rs[ft] * wv + bv;
    }
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) y[kt][ft] = pack8(v[0][ft], v[1][ft]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Multi-head attention for one wave's 32 query frames, fused with the output projection (delta += W_o . concat_heads(...)).
// One head = one MFMA k-tile: per 32-key chunk  S^T = K Q^T (4 MFMAs: 2 key tiles x 2 query tiles), online softmax in fp32 with a
// deferred running maximum (same scheme as the fp32 kernel: the reference point rides in as the accumulator input of the score
// MFMAs and only moves when a chunk exceeds it by 2^32), P -> bf16, O^T += V^T P^T (4 MFMAs).
//   SELF : keys are frames of the same utterance, band |i-j| <= window (layers/attention.py:27-30,108-112)
//   !SELF: keys are the S context tokens, no mask (layers/mla.py:158-179)
// qf(hd, ft): this lane's q operand (8 bf16) of head hd, query tile ft.
// ---------------------------------------------------------------------------------------------------------
// sink(hd, ob): consumes head hd's normalised output (the lane's 8 bf16 per query tile = the B operand of the output projection):
// the fused layer kernel feeds it straight to the projection (ktile16), the stand-alone attention kernel stores it.
template <class C, bool SELF, class QF, class SINK>
EDTTS_DEV void attention16(QF&& qf, const __bf16* __restrict__ Kb, const __bf16* __restrict__ VTb, int kpad, int nkeys,
                           int window, int m0, int lane, SINK&& sink, unsigned long long* stamps = nullptr) {
  constexpr int NF = C::NF;
  int sidx = 0;  // (EDTTS_STAMPS diagnostic builds: fine-grained stamps of head 0)
  (void)sidx; (void)stamps;
  // Kb / VTb: this utterance's K / V^T images, TILE-CONTIGUOUS: K[head][key tile][16 keys][32 d-slots] and
  // V^T[head][32-key chunk][2 d-tiles][16 d][32 key-slots] -- every MFMA operand tile is one contiguous KiB, i.e. one fully
  // coalesced load per wave (kpad = padded key count: head stride = 32 * kpad elements in both images).
  const int fq = lane & 15, g = lane >> 4;
  const float NEG_INF = -__builtin_inff();
  // chunk geometry as in edtts_device.h attention_fused: partition and order are those of the enclosing 32-frame pair of query
  // tiles (mg), also when the wave owns one tile (NF = 1); the per-lane band limits use the wave's own rows
  // (NF = 4: the wave's own 64 frames are the group -- one chunk walk for all four query tiles)
  constexpr int GROUP = NF > 2 ? 16 * NF : 32;
  const int mg = m0 & ~(GROUP - 1);
  int kt_lo, kt_hi;
  if (SELF && window >= 0) {
    const int lo = mg - window;
    kt_lo = ((lo > 0 ? lo : 0) >> 4) & ~1;
    const int hi = mg + GROUP - 1 + window;
    const int last = hi < nkeys - 1 ? hi : nkeys - 1;
    kt_hi = (last >> 4) + 1;
  } else {
    kt_lo = 0;
    kt_hi = (nkeys + 15) >> 4;
  }
  const int nchunk = (kt_hi - kt_lo + 1) / 2;
  const bool off_grid = SELF && window >= 0 && (mg - window > (kt_lo << 4));
  int cdiag = off_grid ? ((mg >> 4) - kt_lo) / 2 : 0;
  if (cdiag >= nchunk) cdiag = nchunk - 1;
  const int klim = (kt_hi << 4) < nkeys ? (kt_hi << 4) : nkeys;
  int lo_d[NF], span[NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    const int qi = m0 + 16 * ft + fq;
    int lo = -(1 << 28), hi = klim - 1 - qi;
    if (SELF && window >= 0) {
      lo = -window;
      hi = hi < window ? hi : window;
    }
    lo_d[ft] = lo;
    span[ft] = hi - lo;
  }
  const unsigned toff = (unsigned)(fq * 32 + 8 * g) * 2u;  // this lane's 16 bytes inside a [16][32] tile
  const size_t hstride = (size_t)32 * kpad;               // elements between heads
  auto clampc = [&](int c) { return c < nchunk ? c : nchunk - 1; };
  // A chunk = two key tiles = 2 KiB of the K image and 2 KiB (both d-tiles) of the V^T image, both contiguous: chunk c of head hd
  // starts at kh / vh + 1024 c elements.  (A tile past kt_hi is read as it is -- the padded images hold finite values -- and masked.)
  // Every instruction counts here: with one wave per SIMD an attention step is bound by its TOTAL instruction count (~5 cycles
  // each, scalar ones included), so the per-step address and geometry arithmetic is kept to a clamp, a shift and an add.
  // (Buffer loads -- descriptor + SGPR offset, which pay on the fp32 kernel -- were measured here and are 2 % SLOWER: 34.9 vs 34.2 ms
  // per call; these steps are bound by their total instruction count, scalar ones included, and the descriptor form adds SALU work.)
  const __bf16* const kh0 = Kb + (size_t)kt_lo * 512;
  const __bf16* const vh0 = VTb + (size_t)(kt_lo >> 1) * 1024;
  auto load_k = [&](int hd, int c, bf8 (&ka)[2]) {
    const __bf16* kp = kh0 + hd * hstride + (size_t)clampc(c) * 1024;
    ka[0] = ldg_bf8(kp, toff);
    ka[1] = ldg_bf8(kp + 512, toff);
  };
  auto load_v = [&](int hd, int c, bf8 (&va)[2]) {
    const __bf16* vp = vh0 + hd * hstride + (size_t)clampc(c) * 1024;
    va[0] = ldg_bf8(vp, toff);
    va[1] = ldg_bf8(vp + 512, toff);
  };
  const int nfull = SELF ? 0 : (klim >> 5) - (kt_lo >> 1);  // cross-attention: chunks [0, nfull) hold 32 valid keys each
  auto chunk_interior_eval = [&](int c) {
    const int k0 = (kt_lo + 2 * c) << 4, k1 = k0 + 31;
    bool full = k1 < klim && (kt_lo + 2 * (c + 1)) <= kt_hi;
    if (window >= 0) full = full && (k1 - m0 <= window) && (k0 - (m0 + 16 * NF - 1) >= -window);
    return full;
  };
  unsigned imask = 0;  // self-attention: bit c = chunk c (of the first 32) is interior; evaluated once, one SALU bit test per step
  if (SELF) {
    for (int c = 0; c < nchunk && c < 32; ++c) imask |= (chunk_interior_eval(c) ? 1u : 0u) << c;
    imask = __builtin_amdgcn_readfirstlane(imask);
  }
  auto chunk_is_interior = [&](int c) {
    if (!SELF) return c < nfull;
    c = clampc(c);
    return c < 32 && ((imask >> c) & 1u) != 0u;
  };
  auto mask_init = [&](int c, f4 (&S)[2][NF], const float (&vis)[NF]) {
    c = clampc(c);
    const int k0 = (kt_lo + 2 * c) << 4;
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) {
      const int d0 = k0 + 4 * g - (m0 + 16 * ft + fq) - lo_d[ft];
      const unsigned sp = span[ft] >= 0 ? (unsigned)span[ft] : 0u;
      const int bias = span[ft] >= 0 ? 0 : (1 << 30);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) S[t][ft][r] = (unsigned)(d0 + bias + 16 * t + r) <= sp ? vis[ft] : NEG_INF;
    }
  };
  // step s = 0: the diagonal chunk; steps 1 .. nchunk-1: the other chunks in ascending order
  auto chunk_of = [&](int st) { return st >= nchunk ? nchunk - 1 : (st == 0 ? cdiag : (st <= cdiag ? st - 1 : st)); };

#ifndef EDTTS16_HP
#define EDTTS16_HP 2   // heads per attention step (1: the single-head step below)
#endif
#if EDTTS16_HP == 2
  // TWO HEADS PER STEP.  Both heads of a pair walk the same chunks with the same masks, so the interior test, the mask predicates,
  // the loop control and the rescale branch are paid once per 2 x 16 scores, and the step holds two independent
  // MFMA -> exp2 -> pack -> MFMA chains for the scheduler to interleave: with one wave per SIMD nothing else hides their latencies
  // (stand-alone attention, 4.5: a second instruction stream per SIMD is worth 1.45x).
  // (NF = 4: one head per step -- its four query tiles are the independent chains, and each K / V^T tile feeds four MFMAs)
  constexpr int HP = NF > 2 ? 1 : 2;
  static_assert(C::HEADS % HP == 0, "head pairs");
  // K / V^T tiles are requested KD steps ahead into KD register buffer sets.  64-frame waves, cross-attention: 3 -- twice the
  // frames per CU double the context K / V^T working set of an XCD (8 utterances x 512 KiB = its whole L2), and at two steps
  // ahead the V^T tiles of every other step arrived 1 000 - 2 000 cycles late (rocprof: L2 hit rate 78 % -> 67 %, SQ_WAIT_ANY + 36 %)
#ifndef EDTTS16_WIDE_KD
#define EDTTS16_WIDE_KD 3
#endif
#ifndef EDTTS16_WIDE_KD_SELF
#define EDTTS16_WIDE_KD_SELF 2
#endif
  constexpr int KD = NF > 2 ? (SELF ? EDTTS16_WIDE_KD_SELF : EDTTS16_WIDE_KD) : 2;
  static_assert(KD == 2 || KD == 3, "K / V^T prefetch depth");
  bf8 KA0[HP][2], VA0[HP][2], KA1[HP][2], VA1[HP][2], KA2[HP][2], VA2[HP][2], q[HP][NF];
  auto prefetch = [&](int hd) {
#pragma unroll
    for (int h = 0; h < HP; ++h) {
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) q[h][ft] = qf(hd + h, ft);
      load_k(hd + h, chunk_of(0), KA0[h]);
      load_v(hd + h, chunk_of(0), VA0[h]);
      load_k(hd + h, chunk_of(1), KA1[h]);
      load_v(hd + h, chunk_of(1), VA1[h]);
      if constexpr (KD == 3) {
        load_k(hd + h, chunk_of(2), KA2[h]);
        load_v(hd + h, chunk_of(2), VA2[h]);
      }
    }
  };
  prefetch(0);
  // EDTTS16_MFMA_SUM (round 4): the softmax row sums come out of the matrix pipe -- one more MFMA per head and query tile and step with an
  // all-ones A operand leaves sum_k P[k][query] in every C/D register of the lane's query -- instead of 8 v_add per head and tile (the
  // partial sums and their accumulation) plus the sum-based rescale test; the test reads the scores' maximum (v_max3 chains) instead.
  // The step is bound by its VALU issue, the MFMA pipe idles three quarters of it.  The sums are those of the bf16-rounded P the P V
  // product itself uses.
#ifndef EDTTS16_MFMA_SUM
#define EDTTS16_MFMA_SUM 1
#endif
  constexpr bool MSUM = EDTTS16_MFMA_SUM;
  f4 ones_bits = as_f4(bf8{(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f});
  asm volatile("" : "+a"(ones_bits));  // four AGPRs for the life of the attention (an MFMA's A operand may be an AGPR)
  for (int hd = 0; hd < C::HEADS; hd += HP) {
    f4 O[HP][2][NF], lvec[HP][NF], NM[HP][NF];
    float nm[HP][NF];
#pragma unroll
    for (int h = 0; h < HP; ++h)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        O[h][0][ft] = O[h][1][ft] = lvec[h][ft] = NM[h][ft] = splat(0.f);
        nm[h][ft] = 0.f;
      }
    auto step = [&](bool first, int c, int cnext2, bf8 (&KA)[HP][2], bf8 (&VA)[HP][2]) {
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
      f4 S[HP][2][NF];
      const bool dummy = c < 0;  // (EDTTS16_EVEN_STEPS: the padding step of an odd step count: every position masked)
      if (dummy) c = nchunk - 1;
      if (!dummy && chunk_is_interior(c)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft)
#pragma unroll
            for (int h = 0; h < HP; ++h) S[h][t][ft] = EDTTS_MFMA16(KA[h][t], q[h][ft], NM[h][ft]);
      } else {
        // the visibility predicates are the pair's; each head selects its own reference point
        int cc = clampc(c);
        if constexpr (NF > 2) asm volatile("" : "+s"(cc));  // (else the first step's 32-register mask tile is hoisted out of the head loop and spilled)
        const int k0 = (kt_lo + 2 * cc) << 4;
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          const int d0 = k0 + 4 * g - (m0 + 16 * ft + fq) - lo_d[ft];
          const unsigned sp = span[ft] >= 0 ? (unsigned)span[ft] : 0u;
          const int bias = span[ft] >= 0 ? 0 : (1 << 30);
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool vis = !dummy && (unsigned)(d0 + bias + 16 * t + r) <= sp;
#pragma unroll
              for (int h = 0; h < HP; ++h) S[h][t][ft][r] = vis ? nm[h][ft] : NEG_INF;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft)
#pragma unroll
            for (int h = 0; h < HP; ++h) S[h][t][ft] = EDTTS_MFMA16(KA[h][t], q[h][ft], S[h][t][ft]);
      }
#pragma unroll
      for (int h = 0; h < HP; ++h) load_k(hd + h, cnext2, KA[h]);  // (re-reads a valid tile past the last step)
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
      f4 P[HP][2][NF], ps[HP][NF];
      auto exp_and_sum = [&](int h, int ft, float m) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[h][t][ft][r] = fast_exp2(S[h][t][ft][r] - m);
        if constexpr (!MSUM) ps[h][ft] = P[h][0][ft] + P[h][1][ft];
      };
      if (first) {
#pragma unroll
        for (int h = 0; h < HP; ++h)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) {
            f4 mv = S[h][0][ft];
#pragma unroll
            for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mv[r], S[h][1][ft][r]);
            const float gm = group_max(hmax(mv));
            const float m = gm > -1e30f ? gm : 0.f;
            nm[h][ft] = -m;
            NM[h][ft] = splat(-m);
            asm volatile("" : "+a"(NM[h][ft]));  // the -m tile lives in AGPRs: it is an accumulator input and nothing else (else 16 v_accvgpr_write per step)
            exp_and_sum(h, ft, m);
          }
      } else {
        const float lim = 4294967296.f;  // 2^32 (kDefer)
        // one test for the whole step: all probabilities are >= 0, so the sum over both heads and query tiles exceeds the limit
        // (or is not finite) whenever one row's does -- a false positive only moves reference points early
        bool over;
        if constexpr (MSUM) {
          float mx = S[0][0][0][0];  // some score above its reference point by more than 32 octaves <=> some P > 2^32
#pragma unroll
          for (int h = 0; h < HP; ++h)
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
              exp_and_sum(h, ft, 0.f);
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[h][t][ft][r]);
            }
          over = !(mx <= 32.f);
          (void)lim;
        } else {
          f4 tot = splat(0.f);
#pragma unroll
          for (int h = 0; h < HP; ++h)
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
              exp_and_sum(h, ft, 0.f);
              tot += ps[h][ft];
            }
          over = !(hsum(tot) <= lim);
        }
        if (__any(over)) {
          // rare path (inputs through a volatile asm: see the single-head step)
          f4 T[HP][2][NF];
#pragma unroll
          for (int h = 0; h < HP; ++h)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int ft = 0; ft < NF; ++ft) {
                T[h][t][ft] = S[h][t][ft];
                asm volatile("" : "+v"(T[h][t][ft]));
              }
#pragma unroll
          for (int h = 0; h < HP; ++h)
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
              f4 mv = T[h][0][ft];
#pragma unroll
              for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mv[r], T[h][1][ft][r]);
              const float dl = fmaxf(0.f, group_max(hmax(mv)));
              const float alpha = fast_exp2(-dl);
              nm[h][ft] -= dl;
              NM[h][ft] = splat(nm[h][ft]);
              asm volatile("" : "+a"(NM[h][ft]));
              lvec[h][ft] *= alpha;
              O[h][0][ft] *= alpha;
              O[h][1][ft] *= alpha;
#pragma unroll
              for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[h][t][ft][r] = fast_exp2(T[h][t][ft][r] - dl);
              if constexpr (!MSUM) ps[h][ft] = P[h][0][ft] + P[h][1][ft];
            }
        }
      }
      bf8 pb[HP][NF];
#pragma unroll
      for (int h = 0; h < HP; ++h)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          if constexpr (!MSUM) lvec[h][ft] += ps[h][ft];
          pb[h][ft] = pack8(P[h][0][ft], P[h][1][ft]);
        }
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
          for (int h = 0; h < HP; ++h) O[h][dt][ft] = EDTTS_MFMA16(VA[h][dt], pb[h][ft], O[h][dt][ft]);
      if constexpr (MSUM) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft)
#pragma unroll
          for (int h = 0; h < HP; ++h) lvec[h][ft] = EDTTS_MFMA16(as_bf8(ones_bits), pb[h][ft], lvec[h][ft]);
      }
#pragma unroll
      for (int h = 0; h < HP; ++h) load_v(hd + h, cnext2, VA[h]);
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
    };
    // Always two steps per loop iteration, so that no branch surrounds a tile request and hipcc's waitcnt pass counts the
    // outstanding loads exactly (behind a conditional step it waited for tiles that had only just been requested); a step past the
    // end of an odd step count runs fully masked: one step in 17 wasted in the cross-attention, 36.6 -> 35.9 ms per call.
    // (Measured before: the conditional step computed under its condition with its requests made unconditional: no gain;
    // unconditional step pairs plus a tail step, a fourth copy of the step: 53.6 ms, 202 spilled registers.)
    step(true, chunk_of(0), chunk_of(KD), KA0, VA0);
#ifndef EDTTS16_EVEN_STEPS
#define EDTTS16_EVEN_STEPS 1   // 1: padded step groups for the 16- and 32-frame instances; 0: never; 2: always
#endif
    // (64-frame waves: a step is twice the work, and the interior self-attention walks 6 chunks = 1 + 5 steps -- the masked
    // padding step costs more than the exact waits gain: 33.5 vs 34.2 ms per call with the last steps behind their conditions)
    constexpr bool EVEN = EDTTS16_EVEN_STEPS == 2 || (EDTTS16_EVEN_STEPS == 1 && NF <= 2);
    if constexpr (KD == 3) {
      // three steps per iteration, always padded: with these steps behind conditions the 16-step cross-attention loses what the
      // third buffer set gained (35.1 vs 33.5 ms per call: the waits are no longer exact)
      for (int st = 1; st < nchunk; st += 3) {
        step(false, chunk_of(st), chunk_of(st + 3), KA1, VA1);
        step(false, st + 1 < nchunk ? chunk_of(st + 1) : -1, chunk_of(st + 4), KA2, VA2);
        step(false, st + 2 < nchunk ? chunk_of(st + 2) : -1, chunk_of(st + 5), KA0, VA0);
      }
    } else
    for (int st = 1; st < nchunk; st += 2) {
      step(false, chunk_of(st), chunk_of(st + 2), KA1, VA1);
      // EVEN: always two steps per iteration (no branch around a load): a step past the end runs fully masked
      if constexpr (EVEN) step(false, st + 1 < nchunk ? chunk_of(st + 1) : -1, chunk_of(st + 3), KA0, VA0);
      else if (st + 1 < nchunk) step(false, chunk_of(st + 1), chunk_of(st + 3), KA0, VA0);
    }
    bf8 ob[HP][NF];
#pragma unroll
    for (int h = 0; h < HP; ++h)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        const float lt = MSUM ? lvec[h][ft][0] : group_sum(hsum(lvec[h][ft]));  // (MSUM: every C/D row holds the query's total)
        const float inv = lt > 0.f ? 1.0f / lt : 0.f;
        ob[h][ft] = pack8(O[h][0][ft] * inv, O[h][1][ft] * inv);
      }
    if (hd + HP < C::HEADS) prefetch(hd + HP);
#pragma unroll
    for (int h = 0; h < HP; ++h) sink(hd + h, ob[h]);
  }
#else
#ifndef EDTTS16_KVDEPTH
#define EDTTS16_KVDEPTH 2
#endif
  // K / V^T operands are requested EDTTS16_KVDEPTH steps ahead into as many register buffers (2; 4 is a build option).  In-kernel
  // stamps (scratch/stamps_bf16.py) show some steps waiting 1 000 - 2 000 cycles for their tiles at either depth, and the call time
  // is the same (44.3 - 45.0 ms): the extra 32 registers of depth 4 cost in accumulator moves what the distance gains.
  constexpr int KD = EDTTS16_KVDEPTH;
  static_assert(KD == 2 || KD == 4, "K/V prefetch depth");
  bf8 KA0[2], VA0[2], KA1[2], VA1[2], KA2[2], VA2[2], KA3[2], VA3[2], q[NF];
  auto prefetch = [&](int hd) {
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) q[ft] = qf(hd, ft);
    load_k(hd, chunk_of(0), KA0);
    load_v(hd, chunk_of(0), VA0);
    load_k(hd, chunk_of(1), KA1);
    load_v(hd, chunk_of(1), VA1);
    if (KD == 4) {
      load_k(hd, chunk_of(2), KA2);
      load_v(hd, chunk_of(2), VA2);
      load_k(hd, chunk_of(3), KA3);
      load_v(hd, chunk_of(3), VA3);
    }
  };
  prefetch(0);
  for (int hd = 0; hd < C::HEADS; ++hd) {
    f4 O[2][NF], lvec[NF], NM[NF];
    float nm[NF];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) {
      O[0][ft] = O[1][ft] = lvec[ft] = NM[ft] = splat(0.f);
      nm[ft] = 0.f;
    }
    auto step = [&](bool first, int c, int cnext2, bf8 (&KA)[2], bf8 (&VA)[2]) {
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
      f4 S[2][NF];
      if (chunk_is_interior(c)) {
        // the reference tile rides in as the C operand itself (written as S = NM; S = mfma(.., S) it cost 16 accumulator moves
        // and as many reads per step); first step: NM = 0
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) S[t][ft] = EDTTS_MFMA16(KA[t], q[ft], NM[ft]);
      } else {
        mask_init(c, S, nm);  // first step: nm = 0
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) S[t][ft] = EDTTS_MFMA16(KA[t], q[ft], S[t][ft]);
      }
      load_k(hd, cnext2, KA);  // (re-reads a valid tile past the last step)
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
      f4 P[2][NF], ps[NF];
      auto lane_max = [&](int ft) {
        f4 mv = S[0][ft];
#pragma unroll
        for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mv[r], S[1][ft][r]);
        return hmax(mv);
      };
      auto exp_and_sum = [&](int ft, float m) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[t][ft][r] = fast_exp2(S[t][ft][r] - m);
        ps[ft] = P[0][ft] + P[1][ft];
      };
      if (first) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          const float gm = group_max(lane_max(ft));
          const float m = gm > -1e30f ? gm : 0.f;
          nm[ft] = -m;
          NM[ft] = splat(-m);
          exp_and_sum(ft, m);
        }
      } else {
        const float lim = 4294967296.f;  // 2^32 (kDefer)
        bool over = false;
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          exp_and_sum(ft, 0.f);
          over = over || !(hsum(ps[ft]) <= lim);
        }
        if (__any(over)) {
          // Rare path.  Its inputs pass through a volatile asm: hipcc otherwise executes the whole path speculatively in EVERY step
          // (if-conversion: the ISA showed 32 v_exp, 24 v_max and 16 v_sub per step instead of 16 / 0 / 0).
          f4 T[2][NF];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ft = 0; ft < NF; ++ft) {
              T[t][ft] = S[t][ft];
              asm volatile("" : "+v"(T[t][ft]));
            }
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) {
            f4 mv = T[0][ft];
#pragma unroll
            for (int r = 0; r < 4; ++r) mv[r] = fmaxf(mv[r], T[1][ft][r]);
            const float dl = fmaxf(0.f, group_max(hmax(mv)));
            const float alpha = fast_exp2(-dl);
            nm[ft] -= dl;
            NM[ft] = splat(nm[ft]);
            lvec[ft] *= alpha;
            O[0][ft] *= alpha;
            O[1][ft] *= alpha;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) P[t][ft][r] = fast_exp2(T[t][ft][r] - dl);
            ps[ft] = P[0][ft] + P[1][ft];
          }
        }
      }
      bf8 pb[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        lvec[ft] += ps[ft];
        pb[ft] = pack8(P[0][ft], P[1][ft]);
      }
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) O[dt][ft] = EDTTS_MFMA16(VA[dt], pb[ft], O[dt][ft]);
      load_v(hd, cnext2, VA);
      if (hd == EDTTS_STAMP_HEAD) STAMPX16(stamps, sidx++);
    };
    // (Measured alternatives, same device: groups of KD unconditional steps + a load-free tail so that hipcc's waitcnt pass sees
    // the same number of outstanding loads on every path -- KD=2: 50.1 ms, KD=4: 46.2 ms against 45.7 ms for this loop; fully
    // unrolled 16-step runs: 59 ms (spills); a skewed step that issues the score MFMAs of chunk s+1 ahead of the softmax of chunk s
    // (matrix pipe under the VALU clump): 47.2 ms -- its 16 extra accumulators cost more in AGPR<->VGPR moves than the overlap
    // gains.  The step is bound by its instruction count at one wave per SIMD, not by the distance of its loads.)
    step(true, chunk_of(0), chunk_of(KD), KA0, VA0);
    if (KD == 2) {
      for (int st = 1; st < nchunk; st += 2) {
        step(false, chunk_of(st), chunk_of(st + 2), KA1, VA1);
        if (st + 1 < nchunk) step(false, chunk_of(st + 1), chunk_of(st + 3), KA0, VA0);
      }
    } else {
      for (int st = 1; st < nchunk; st += 4) {
        step(false, chunk_of(st), chunk_of(st + 4), KA1, VA1);
        if (st + 1 < nchunk) step(false, chunk_of(st + 1), chunk_of(st + 5), KA2, VA2);
        if (st + 2 < nchunk) step(false, chunk_of(st + 2), chunk_of(st + 6), KA3, VA3);
        if (st + 3 < nchunk) step(false, chunk_of(st + 3), chunk_of(st + 7), KA0, VA0);
      }
    }
    bf8 ob[NF];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) {
      const float lt = group_sum(hsum(lvec[ft]));
      const float inv = lt > 0.f ? 1.0f / lt : 0.f;
      ob[ft] = pack8(O[0][ft] * inv, O[1][ft] * inv);
    }
    if (hd + 1 < C::HEADS) prefetch(hd + 1);
    sink(hd, ob);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// QKV of one layer from the packed normalised tile: q, k row-major [B][Tp][H] (slot order inside each head), v transposed
// [B][H][Tp] (slot order inside each 32-key chunk) -- layers/attention.py:91-93
// ---------------------------------------------------------------------------------------------------------
template <class C>
EDTTS_DEV void qkv_tail16(Ring16<C>& ring, const bf8 (&hn)[C::KT][C::NF], const KArgs& a, int b, int m0, int lane, bool valid) {
  constexpr int NF = C::NF;
  const int fq = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * a.Tp + m0 + fq;
  __bf16* const qo = reinterpret_cast<__bf16*>(a.q_out);
  __bf16* const ko = reinterpret_cast<__bf16*>(a.k_out);
  __bf16* const vo = reinterpret_cast<__bf16*>(a.vT_out);
  STAMP16(120);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    STAMP16(121 + which);  // (121: start of the q phases, 122: start of the k phases)
    for (int p = 0; p < C::KT; ++p) {
      f4 acc[2][NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = splat(0.f);
      gemm16_pair<C::KT, false>(ring, hn, acc[0], acc[1]);
      // q: the wave's private image (q_at; read back by this wave only); k: tile-contiguous image [head p][key tile][16 keys][32 slots].
      // PLAIN stores, not streaming ones: on gfx9 stores retire through the same in-order vmcnt as the ring's DMAs, and a
      // nontemporal store takes microseconds to be acknowledged -- scratch/ring_probe.cpp: 9 060 vs 1 920 cycles per phase.
      // (Measured and dropped: the residual tile's 32 stores spread over the q phases instead of one burst in front of the tail,
      // with and without an allowance for the known younger stores in the ring's counted vmcnt waits -- 38.6 - 38.9 ms either way;
      // skipping the q / k / v^T stores altogether: 36.8 ms.)
      __bf16* dst = which == 0 ? qo + q_at<C>(((size_t)b * a.Tp + m0) * C::H, lane, p, 0)
                               : ko + ((size_t)(b * C::HEADS + p) * (a.Tp >> 4) + (m0 >> 4)) * 512 + fq * 32 + 8 * g;
      const size_t fstride = which == 0 ? q_at<C>(0, 0, 0, 1) : 512;
      if (valid)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft)
        *reinterpret_cast<f4*>(dst + ft * fstride) = as_f4(pack8(acc[0][ft], acc[1][ft]));
    }
  }
  STAMP16(123);  // start of the v^T phases
  for (int p = 0; p < C::KT; ++p) {
    f4 acc[2][NF];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = splat(0.f);
    gemm16_pair<C::KT, true>(ring, hn, acc[0], acc[1]);  // C/D = [frame 4g+r of tile ft][feature 16(2p+u) + fq]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // v^T image: [head p][32-key chunk][d-tile u][16 d][32 key slots]; key 16 t + 4 g + r of the chunk sits at slot 8 g + 4 t + r
      __bf16* dst = vo + ((size_t)(b * C::HEADS + p) * (a.Tp >> 5) + (m0 >> 5)) * 1024 + u * 512 + fq * 32 + 8 * g;
      if (NF >= 2) {
        if (valid)
#pragma unroll
          for (int cp = 0; cp < NF / 2; ++cp)  // one 32-key chunk per pair of frame tiles
            *reinterpret_cast<f4*>(dst + cp * 1024) = as_f4(pack8(acc[u][2 * cp], acc[u][(2 * cp + 1) % NF]));
      } else {  // one 16-frame tile per wave: its 4 keys per lane are half of the 8-slot group (t = tile parity inside the chunk)
        if (valid) *reinterpret_cast<f2s*>(dst + 4 * ((m0 >> 4) & 1)) = pack4(acc[u][0]);
      }
    }
  }
}

// =========================================================================================================
// prologue: h = in_proj(x) + pe ; AdaRMSNorm(layer 0) ; QKV(layer 0)
// =========================================================================================================
template <class C>
__global__ __launch_bounds__(C::THREADS, C::MIN_WAVES_PER_SIMD) void k_prologue16(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) f4 ring_lds16[];
  // (a padding wave of the last block works on a copy of the last tile -- it takes part in the ring's barriers and DMAs -- and
  // stores nothing)
  const TileId tl = wave_tile(a.B, a.Tp, C::WAVES, C::WF);
  const bool valid = tl.valid;
  const int lane = threadIdx.x & 63, fq = lane & 15, g = lane >> 4;
  const int b = tl.b, m0 = tl.m0;
  Ring16<C> ring;
  ring.start(a.stream, ring_lds16, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
  constexpr int NF = C::NF;
  bf8 xin[C::MKT][NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    const int f = m0 + 16 * ft + fq;
    const float* xr = a.x + ((size_t)b * a.T + f) * C::MEL;
#pragma unroll
    for (int kt = 0; kt < C::MKT; ++kt) {
      const int c0 = 32 * kt + 4 * g, c1 = c0 + 16;
      const f4 v0 = (f < a.T && c0 < C::MEL) ? ldg4(xr + c0) : splat(0.f);
      const f4 v1 = (f < a.T && c1 < C::MEL) ? ldg4(xr + c1) : splat(0.f);
      xin[kt][ft] = pack8(v0, v1);
    }
  }
  f4 h[C::HT][NF];
#pragma unroll
  for (int nt = 0; nt < C::HT; ++nt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) acc_put<(NF > 2)>(h[nt][ft], ldg4(a.inp_b + 16 * nt + 4 * g));
#pragma unroll
  for (int kt = 0; kt < C::MKT; ++kt) ktile16<C::HT>(ring, xin[kt], h);
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    int f = m0 + 16 * ft + fq;
    f = f < a.max_pos ? f : a.max_pos - 1;
#pragma unroll
    for (int nt = 0; nt < C::HT; ++nt)
      acc_put<(NF > 2)>(h[nt][ft], acc_get<(NF > 2)>(h[nt][ft]) + ldg4(a.pe + (size_t)f * C::H + 16 * nt + 4 * g));
  }
  if (valid) {
    const size_t tile0 = ((size_t)b * a.Tp + m0) * C::H;
#pragma unroll
    for (int nt = 0; nt < C::HT; ++nt)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) stg4(a.h + h_at<C>(tile0, lane, nt, ft), acc_get<(NF > 2)>(h[nt][ft]));
  }
  bf8 hn[C::KT][NF];
  rms_norm_pack<C>(h, a.n1w, a.cond + (size_t)b * a.cond_bstride, g, hn);
  qkv_tail16<C>(ring, hn, a, b, m0, lane, valid);
  ring.drain();
}

// =========================================================================================================
// transformer layer kernel (bf16 contractions)
// =========================================================================================================
// ---------------------------------------------------------------------------------------------------------
// Stand-alone attention (split layer): the same attention16 as the fused kernel, but with nothing else live -- no residual tile,
// no weight ring, no LDS -- so that several waves share a SIMD and hide each other's softmax / load latency.  One wave = 32 query
// frames, all heads in turn; q rows in, normalised O rows out (both [frame][H] bf16, slot order inside a head).
// ---------------------------------------------------------------------------------------------------------
#ifndef EDTTS16_ATT_OCC
#define EDTTS16_ATT_OCC 2
#endif
template <class C, bool SELF>
__global__ __launch_bounds__(C::THREADS, EDTTS16_ATT_OCC) void k_attn16(KArgs a) {
  const TileId tl = wave_tile(a.B, a.Tp, C::WAVES, C::WF);
  if (!tl.valid) return;  // (no block-level synchronisation in this kernel)
  constexpr int NF = C::NF;
  const int lane = threadIdx.x & 63, fq = lane & 15, g = lane >> 4;
  const int b = tl.b, m0 = tl.m0;
  const size_t rowbase = (size_t)b * a.Tp + m0 + fq;
  const __bf16* qbase = reinterpret_cast<const __bf16*>(a.attn_q);
  __bf16* orow = reinterpret_cast<__bf16*>(a.attn_o) + rowbase * C::H + 8 * g;
  auto qf = [&](int hd, int ft) { return *reinterpret_cast<const bf8*>(qbase + q_at<C>(((size_t)b * a.Tp + m0) * C::H, lane, hd, ft)); };
  auto sink = [&](int hd, const bf8 (&ob)[NF]) {
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) *reinterpret_cast<bf8*>(orow + (size_t)ft * 16 * C::H + hd * C::DH) = ob[ft];
  };
  if (SELF)
    attention16<C, true>(qf, reinterpret_cast<const __bf16*>(a.k) + (size_t)b * a.Tp * C::H,
                         reinterpret_cast<const __bf16*>(a.vT) + (size_t)b * C::H * a.Tp, a.Tp, a.T, a.window, m0, lane, sink);
  else
    attention16<C, false>(qf, reinterpret_cast<const __bf16*>(a.kc) + (size_t)b * a.Sp * C::H,
                          reinterpret_cast<const __bf16*>(a.vcT) + (size_t)b * C::H * a.Sp, a.Sp, a.S, -1, m0, lane, sink);
}

// PART16_ALL: the whole layer in one launch.  Split layer: k_attn16<self> | PART16_MID (self out-projection, norm2, cross q) |
// k_attn16<cross> | PART16_POST (cross out-projection, FFN, tail) -- same arithmetic in the same order, bitwise the same result.
enum { PART16_ALL = 0, PART16_MID = 1, PART16_POST = 2 };
// (register-pressure probes: -DEDTTS_EXPERIMENTS -DEDTTS16_PHASES=<mask> compiles only the masked phases of the layer -- 1 self-attention,
// 2 cross-attention, 4 FFN, 8 tail; results wrong by construction)
#ifndef EDTTS16_PHASES
#define EDTTS16_PHASES 15
#endif
template <class C, int TAIL, int PART = PART16_ALL>
__global__ __launch_bounds__(C::THREADS, C::MIN_WAVES_PER_SIMD) void k_layer16(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) f4 ring_lds16[];
  const TileId tl = wave_tile(a.B, a.Tp, C::WAVES, C::WF);
  const bool valid = tl.valid;  // a padding wave works on a copy of the last tile (ring barriers, DMAs) and stores nothing
  const int lane = threadIdx.x & 63, fq = lane & 15, g = lane >> 4;
  const int b = tl.b, m0 = tl.m0;
  const size_t rowbase = (size_t)b * a.Tp + m0 + fq;
  Ring16<C> ring;
  // weight stream of a layer: self out-projection (HEADS k-tiles of HT fragments) | cross q (KT pairs of n-tiles over KT k-tiles) |
  // cross out-projection | FFN | tail; a fragment = 16 x 32 bf16 = 256 floats
  constexpr int kPostStreamFloats = (C::HEADS * C::HT + 2 * C::KT * C::KT) * 256;
  static_assert((C::HEADS * C::HT + 2 * C::KT * C::KT) % C::PH == 0, "the POST part starts on a phase boundary");
  ring.start(PART == PART16_POST ? a.stream + kPostStreamFloats : a.stream, ring_lds16,
             __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane);
  // Small parameter vectors that are read INSIDE the streaming loops go through LDS: an ordinary global load consumed while
  // ring DMAs are in flight makes hipcc's waitcnt pass emit s_waitcnt vmcnt(0) (it cannot count the DMAs of earlier loop
  // iterations), which drains the whole prefetch ring once per phase -- measured: 2 800 cycles per 512-cycle phase.
  float* const params = reinterpret_cast<float*>(ring_lds16 + C::NS * C::PH * 64);
  for (int i = threadIdx.x; i < 32 * a.ffn_tiles; i += C::THREADS) params[i] = a.up_b[i];  // 2 * ffn_mult * H
  if (TAIL != TAIL_QKV)
    for (int i = threadIdx.x; i < C::MEL; i += C::THREADS) params[C::UPB_FLOATS + i] = a.outp_b[i];
  __syncthreads();

  // residual tile (fp32).  Unlike the fp32 kernel, the branches accumulate straight into it: the rounding of the partial sums at
  // the residual's magnitude (~1e-6) is three orders below the bf16 operand rounding, and a separate branch tile would cost 128
  // more registers (measured: spills, whose scratch reloads force s_waitcnt vmcnt(0) and drain the weight ring).
  constexpr int NF = C::NF;
  constexpr bool PIN = NF > 2;  // (64-frame waves: the tile lives in the AGPRs, see acc_get)
  f4 h[C::HT][NF];
  const size_t tile0 = ((size_t)b * a.Tp + m0) * C::H;  // the tile's span of the h / q buffers (private images: h_at, q_at)
#pragma unroll
  for (int nt = 0; nt < C::HT; ++nt) {
    const f4 pb = PART == PART16_POST ? splat(0.f) : ldg4(a.proj_b + 16 * nt + 4 * g);
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) acc_put<PIN>(h[nt][ft], ldg4(a.h + h_at<C>(tile0, lane, nt, ft)) + pb);
  }
  // split layer: this wave's attention output rows (all heads), fetched and WAITED FOR before the streaming loop (a global load
  // consumed inside it would make hipcc drain the ring with vmcnt(0) every phase), then projected head by head
  auto project_attn_rows = [&]() {
    bf8 oa[C::HEADS][NF];
    const __bf16* orow = reinterpret_cast<const __bf16*>(a.attn_o) + rowbase * C::H + 8 * g;
#pragma unroll
    for (int hd = 0; hd < C::HEADS; ++hd)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        oa[hd][ft] = *reinterpret_cast<const bf8*>(orow + (size_t)ft * 16 * C::H + hd * C::DH);
      }
#pragma unroll
    for (int hd = 0; hd < C::HEADS; ++hd)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) asm volatile("" : "+v"(oa[hd][ft]));
#pragma unroll
    for (int hd = 0; hd < C::HEADS; ++hd) ktile16<C::HT>(ring, oa[hd], h);
  };
  STAMP16(0);
  // ---- x = x + attn(norm1(x, cond))   (transformer.py:142-146; q / k / v^T were produced by the previous kernel) ----
  if (PART == PART16_MID) project_attn_rows();
  if (PART == PART16_ALL && (EDTTS16_PHASES & 1)) {
    const __bf16* qbase = reinterpret_cast<const __bf16*>(a.q);
    // (64-frame waves: descriptor + scalar offset -- four row pointers held across the head loop were spilled and reloaded per head)
    const __amdgpu_buffer_rsrc_t rsq = make_rsrc(qbase + tile0);
    const unsigned qvoff = q_voff<C>(lane);
    auto qf = [&](int hd, int ft) {
      if constexpr (PIN) return as_bf8(bufld4(rsq, qvoff, q_soff<C>(hd, ft)));
      else return *reinterpret_cast<const bf8*>(qbase + q_at<C>(tile0, lane, hd, ft));
    };
    attention16<C, true>(qf, reinterpret_cast<const __bf16*>(a.k) + (size_t)b * a.Tp * C::H,
                         reinterpret_cast<const __bf16*>(a.vT) + (size_t)b * C::H * a.Tp, a.Tp, a.T, a.window, m0, lane,
                         [&](int, const bf8 (&ob)[NF]) { ktile16<C::HT>(ring, ob, h); }  // h += Wo[:, head] . O
#ifdef EDTTS_STAMPS
                         , a.stamps ? a.stamps + 8 : nullptr
#endif
                         );
  }
  STAMP16(1);
  // ---- x = x + cross_attn(norm2(x), context)   (transformer.py:151, mla.py:118-194) ----
  if (PART == PART16_POST) project_attn_rows();
  if (PART == PART16_MID) {
    // cross q of this wave's rows -> memory, residual -> memory; the cross-attention kernel and PART16_POST take over
    bf8 hn[C::KT][NF];
    rms_norm_pack<C>(h, a.n2w, nullptr, g, hn);
    __bf16* qcb = reinterpret_cast<__bf16*>(a.qc_out);
    for (int p = 0; p < C::KT; ++p) {
      f4 acc[2][NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = splat(0.f);
      gemm16_pair<C::KT, false>(ring, hn, acc[0], acc[1]);
      if (valid)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) *reinterpret_cast<f4*>(qcb + q_at<C>(tile0, lane, p, ft)) = as_f4(pack8(acc[0][ft], acc[1][ft]));
    }
    if (valid) {
#pragma unroll
      for (int nt = 0; nt < C::HT; ++nt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) stg4(a.h + h_at<C>(tile0, lane, nt, ft), h[nt][ft]);
    }
    ring.drain();
    return;
  }
  if (PART == PART16_ALL && (EDTTS16_PHASES & 2)) {
    // cross-attention q of all heads: written once after the q projection, read back head by head (64 / 128 registers less during
    // the cross-attention).  QLDS: in this wave's LDS slice; else (NF = 4) in this wave's own rows of the layer's q buffer, which
    // its self-attention has finished with (no other wave reads q rows; a padding wave stores nothing and reads what it finds).
    f4* const qlds = ring_lds16 + C::NS * C::PH * 64 + (C::PARAM_FLOATS + 3) / 4 +
                     (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * C::HEADS * NF) * 64 + lane;
    const __amdgpu_buffer_rsrc_t rsp = make_rsrc(reinterpret_cast<const __bf16*>(a.q) + tile0);
    const unsigned pvoff = q_voff<C>(lane);
    {
      bf8 hn[C::KT][NF];
      rms_norm_pack<C>(h, a.n2w, nullptr, g, hn);
      for (int p = 0; p < C::KT; ++p) {
        f4 acc[2][NF];
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) acc[0][ft] = acc[1][ft] = splat(0.f);
        gemm16_pair<C::KT, false>(ring, hn, acc[0], acc[1]);
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          if constexpr (C::QLDS) qlds[(p * NF + ft) * 64] = as_f4(pack8(acc[0][ft], acc[1][ft]));
          else if (valid)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, pack8(acc[0][ft], acc[1][ft])),
                                                   rsp, pvoff, q_soff<C>(p, ft), 0);
        }
      }
    }
    // (the parked rows are read back by the lanes that wrote them; the stores have left the wave before the first read is issued)
    if constexpr (!C::QLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP16(2);
    auto qf = [&](int hd, int ft) {
      if constexpr (C::QLDS) return as_bf8(qlds[(hd * NF + ft) * 64]);
      else return as_bf8(bufld4(rsp, pvoff, q_soff<C>(hd, ft)));
    };
    attention16<C, false>(qf, reinterpret_cast<const __bf16*>(a.kc) + (size_t)b * a.Sp * C::H,
                          reinterpret_cast<const __bf16*>(a.vcT) + (size_t)b * C::H * a.Sp, a.Sp, a.S, -1, m0, lane,
                          [&](int, const bf8 (&ob)[NF]) { ktile16<C::HT>(ring, ob, h); }
#ifdef EDTTS_STAMPS
                          , a.stamps ? a.stamps + 40 : nullptr
#endif
                          );
  }
  STAMP16(3);
  // ---- x = x + ffn(norm3(x, cond))   (transformer.py:154-158, :13-49) ----
  if (EDTTS16_PHASES & 4) {
    bf8 hn[C::KT][NF];
    rms_norm_pack<C>(h, a.n3w, a.cond + (size_t)b * a.cond_bstride + ((size_t)a.layer * 2 + 1) * 2 * C::H, g, hn);
#pragma unroll
    for (int nt = 0; nt < C::HT; ++nt) {
      const f4 db = ldg4(a.down_b + 16 * nt + 4 * g);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) acc_put<PIN>(h[nt][ft], acc_get<PIN>(h[nt][ft]) + db);
    }
    for (int jp = 0; jp < a.ffn_tiles / 2; ++jp) {  // ffn_mult * H hidden features = k-tiles (32 wide) of the down projection
      f4 act[2][NF];
      f2s half[2][NF];  // (NF = 4: each n-tile's activations are packed as soon as they exist -- 16 registers instead of 32)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = 2 * jp + u;
        f4 v[NF], gt[NF];
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) v[ft] = gt[ft] = splat(0.f);
        gemm16_pair<C::KT, false>(ring, hn, v, gt);
        const f4 vb = *reinterpret_cast<const f4*>(params + 32 * j + 4 * g), gb = *reinterpret_cast<const f4*>(params + 32 * j + 16 + 4 * g);
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          v[ft] += vb;
          gt[ft] += gb;
          const f4 e = {fast_exp2(gt[ft][0] * -1.4426950408889634f), fast_exp2(gt[ft][1] * -1.4426950408889634f),
                        fast_exp2(gt[ft][2] * -1.4426950408889634f), fast_exp2(gt[ft][3] * -1.4426950408889634f)};
          const f4 d = e + 1.0f;
          const f4 rc = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
          act[u][ft] = (v[ft] * gt[ft]) * rc;  // SwiGLU: value * silu(gate), transformer.py:21-23
          if constexpr (NF > 2) half[u][ft] = pack4(act[u][ft]);
        }
      }
      bf8 ab[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        if constexpr (NF > 2) ab[ft] = as_bf8(f4{half[0][ft][0], half[0][ft][1], half[1][ft][0], half[1][ft][1]});
        else ab[ft] = pack8(act[0][ft], act[1][ft]);
      }
      ktile16<C::HT>(ring, ab, h);
    }
  }
  STAMP16(4);
  // ---- tail ----
  // (64-frame waves: the tail's addresses are formed from fresh copies of the tile coordinates -- hipcc otherwise forms them at the top
  // of the kernel and carries them through every phase in registers it does not have)
  int lane_t = lane, b_t = b, m0_t = m0;
  if constexpr (PIN) asm volatile("" : "+v"(lane_t), "+s"(b_t), "+s"(m0_t));
  const int fq_t = lane_t & 15, g_t = lane_t >> 4;
  const size_t tile0_t = ((size_t)b_t * a.Tp + m0_t) * C::H;
  if (!(EDTTS16_PHASES & 8)) {
    f4 t = splat(0.f);
#pragma unroll
    for (int nt = 0; nt < C::HT; ++nt)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) t += acc_get<PIN>(h[nt][ft]);
    if (valid) stg4(a.h + h_at<C>(tile0_t, lane_t, 0, 0), t);
  } else if (TAIL == TAIL_QKV) {
    if (valid) {
#pragma unroll
      for (int nt = 0; nt < C::HT; ++nt)
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) stg4(a.h + h_at<C>(tile0_t, lane_t, nt, ft), acc_get<PIN>(h[nt][ft]));
    }
    bf8 hn[C::KT][NF];
    rms_norm_pack<C>(h, a.n1w, a.cond + (size_t)b_t * a.cond_bstride + ((size_t)(a.layer + 1) * 2) * 2 * C::H, g_t, hn);
    qkv_tail16<C>(ring, hn, a, b_t, m0_t, lane_t, valid);
  } else {
    bf8 hn[C::KT][NF];
    layer_norm_pack<C>(h, a.fnw, a.fnb, g_t, hn);
    for (int p = 0; p < C::MTP; ++p) {
      f4 e[2][NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) e[0][ft] = e[1][ft] = splat(0.f);
      gemm16_pair<C::KT, false>(ring, hn, e[0], e[1]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int nt = 2 * p + u;
        if (nt >= C::MT) continue;  // the padding half of the last pair
        const f4 ob = *reinterpret_cast<const f4*>(params + C::UPB_FLOATS + 16 * nt + 4 * g_t);
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) {
          const int f = m0_t + 16 * ft + fq_t;
          if (f >= a.T || !valid) continue;
          tail_apply<TAIL>(a, ((size_t)b_t * a.T + f) * C::MEL + 16 * nt + 4 * g_t, e[u][ft] + ob);
        }
      }
    }
  }
  STAMP16(5);
  ring.drain();
  STAMP16(6);
}

}  // namespace edtts16
