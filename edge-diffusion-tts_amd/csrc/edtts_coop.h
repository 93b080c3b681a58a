// edtts_coop.h -- the COOPERATIVE transformer-layer kernel: W waves share one frame tile.
//
// Why.  k_layer gives every wave a whole 16*NF-frame tile and a serial chain of ~7 000 * NF MFMAs per layer: right for large
// batches (thousands of tiles, one wave per SIMD, no synchronisation at all), wrong for small ones -- the reference's real call
// shape is B = 1 (inference.py:55-62, generate_sample.py:94): at T = 256 that is 16 waves on a chip with 1024 SIMDs and 0.104 ms
// per layer launch, all of it one wave's dependent instruction stream.  Here the W waves of a tile split the work that splits
// without changing any sum:
//   * attention: by HEADS (wave w takes heads w, w + W, ...; a head's chunk walk is attention_fused's, untouched); the normalised
//     O^T tiles go to LDS in MFMA C/D layout, which is the projection's B operand as it stands;
//   * every GEMM: by OUTPUT tiles (wave w computes n-tiles w, w + W, ... over ALL k-tiles, in the k order of the one-wave kernel),
//     so every output element sees exactly the products, in exactly the order, of k_layer: results are BITWISE those of the
//     one-wave instances (tests/test_gpu_parity.py::test_cooperative_instances_equal_the_one_wave_ones);
//   * the residual tile and the normalised tile are REPLICATED: each wave holds all of h (and recomputes the norms: a few hundred
//     VALU instructions), so a GEMM's B operand never has to be gathered -- only GEMM OUTPUTS cross waves (through LDS, one block
//     barrier per phase: ~10 per layer).
// A block is the W waves of one tile.  (The two-wave blocks of W = 2 need > 256 registers per wave and 60 KiB of LDS: two blocks
// per CU, on different SIMDs.)  Weight fragments are read straight from the packed stream of k_layer (edtts_pack_weights) at
// computed offsets -- no second weight layout.
#pragma once
#include "edtts_device.h"

namespace edtts {

template <class C, int W>
struct Coop {
  static constexpr int NF = C::NF, HT = C::HT;
  static constexpr int TILES = 1;                                      // frame tiles per block (a block = the W waves of ONE tile: no barrier couples two tiles)
  static constexpr int THREADS = 64 * W;
  static constexpr int O_BYTES = C::HEADS * C::OHEAD_BYTES;            // attention outputs of all heads (B-operand layout)
  static constexpr int Q_BYTES = C::WF * C::H * 4;                     // cross-attention q rows [WF][H]
  static constexpr int ACT_TILES = 2 * HT;                             // hidden 16-feature tiles per FFN group
  static constexpr int ACT_BYTES = ACT_TILES * NF * 1024;              // (shares the attention buffers: they are dead by then)
  static constexpr int A_BYTES = O_BYTES + Q_BYTES > ACT_BYTES ? O_BYTES + Q_BYTES : ACT_BYTES;
  static constexpr int X_BYTES = HT * NF * 1024;                       // one GEMM output tile in register layout
  static constexpr int TILE_BYTES = A_BYTES + X_BYTES;
  static constexpr int LDS_BYTES = TILES * TILE_BYTES;
  static constexpr bool FITS = LDS_BYTES * (4 / W) <= 160 * 1024;  // four waves per CU: 4 / W blocks (hidden 256: 32-frame tiles x 2 waves would need two 96-KiB blocks: not offered)
  static_assert(W == 2 || W == 4, "waves per tile");
};

// acc_a (and acc_b) += sum over KT k-steps of frag(k) * bop(k): the n-split building block.  Fragments come straight from the
// packed stream (byte offsets fa(k) / fb(k) relative to the descriptor base) through a register ring of LK k-steps that the
// CALLER owns: on entry it holds k-steps 0 .. LK-1 (co_preload), and every slot is re-requested right after its MFMAs -- with
// k-step k + LK of this sequence, or, behind its end, with k-step k + LK - KT of the NEXT sequence (na / nb), so that a chain of
// calls never starts on an exposed load.  nr(k) MFMA steps of k-tile k are issued (2 for the 8-row remainder k-tile of a head,
// else 4).  Accumulation order per output element: k ascending, r ascending -- k_layer's.
template <int LK, bool TWO, class FA, class FB>
EDTTS_DEV void co_preload(__amdgpu_buffer_rsrc_t rs, unsigned voff, FA&& fa, FB&& fb, f4 (&ra)[LK], f4 (&rb)[LK]) {
#pragma unroll
  for (int i = 0; i < LK; ++i) {
    ra[i] = bufld4(rs, voff, fa(i));
    if (TWO) rb[i] = bufld4(rs, voff, fb(i));
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int KT, int NF, int LK, bool TWO, bool TWO_NEXT, class FA, class FB, class NA, class NB, class BOP, class NR>
EDTTS_DEV void co_ktiles(__amdgpu_buffer_rsrc_t rs, unsigned voff, f4 (&ra)[LK], f4 (&rb)[LK], FA&& fa, FB&& fb, NA&& na, NB&& nb, BOP&& bop,
                         NR&& nr, f4 (&a)[NF], f4 (&b)[NF]) {
  static_assert(LK <= KT && KT % LK == 0, "the ring length must divide the sequence length (slot k % LK holds k-step k of every chained sequence)");
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    f4 in[NF];
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) in[ft] = bop(kt, ft);
    const f4 wa = ra[kt % LK], wb = TWO ? rb[kt % LK] : splat(0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r >= nr(kt)) continue;
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) a[ft] = EDTTS_MFMA(wa[r], in[ft][r], a[ft]);
      if (TWO) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) b[ft] = EDTTS_MFMA(wb[r], in[ft][r], b[ft]);
      }
    }
    if (kt + LK < KT) {
      ra[kt % LK] = bufld4(rs, voff, fa(kt + LK));
      if (TWO) rb[kt % LK] = bufld4(rs, voff, fb(kt + LK));
    } else {
      ra[kt % LK] = bufld4(rs, voff, na(kt + LK - KT));
      if (TWO_NEXT) rb[kt % LK] = bufld4(rs, voff, nb(kt + LK - KT));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Ring length: the smallest divisor of the sequence length KT that covers the L2 latency (3 k-steps of 16 MFMAs at NF = 2, 4 of 8
// at NF = 1).  A DIVISOR, because a chain of sequences shares the ring: slot k % LK must hold k-step k of EVERY sequence.
constexpr int co_look(int NF, int KT) {
  int lk = NF >= 2 ? 3 : 4;
  while (lk < KT && KT % lk != 0) ++lk;
  return lk < KT ? lk : KT;
}

// One n-split GEMM phase: this wave's output tiles wv, wv + W, ... of `ntiles`, two at a time (four independent accumulator chains
// at NF = 2, two at NF = 1: 64 cycles between dependent MFMAs >= the 40-cycle latency), a last odd one alone; one continuous
// fragment stream over the tiles (see co_ktiles).  preload() requests the first fragments and may run BEFORE the block barrier
// that publishes the phase's B operand: weights do not depend on it.
//   frag(nt, k) -> byte offset of the fragment of output tile nt, k-step k;  init(nt) -> f4 start value (bias);  out(nt, acc)
template <int KT, int NF, int W>
struct CoGemm {
  static constexpr int LK = co_look(NF, KT);
  f4 ra[LK], rb[LK];
  template <class FRAG>
  EDTTS_DEV void preload(__amdgpu_buffer_rsrc_t rs, unsigned voff, int wv, int ntiles, FRAG&& frag) {
    if (wv >= ntiles) return;
    const int tb = wv + W < ntiles ? wv + W : wv;
    co_preload<LK, true>(rs, voff, [&](int k) { return frag(wv, k); }, [&](int k) { return frag(tb, k); }, ra, rb);
  }
  template <class FRAG, class BOP, class NR, class INIT, class OUT>
  EDTTS_DEV void run(__amdgpu_buffer_rsrc_t rs, unsigned voff, int wv, int ntiles, FRAG&& frag, BOP&& bop, NR&& nr, INIT&& init, OUT&& out) {
    int nt = wv;
    for (; nt + W < ntiles; nt += 2 * W) {
      f4 a[NF], b[NF];
      const f4 ia = init(nt), ib = init(nt + W);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) { a[ft] = ia; b[ft] = ib; }
      // the tiles behind this pair (none left: re-request this pair's own fragments -- harmless, and the request count stays uniform)
      const int na_ = nt + 2 * W < ntiles ? nt + 2 * W : nt, nb_ = nt + 3 * W < ntiles ? nt + 3 * W : na_;
      co_ktiles<KT, NF, LK, true, true>(rs, voff, ra, rb, [&](int k) { return frag(nt, k); }, [&](int k) { return frag(nt + W, k); },
                                        [&](int k) { return frag(na_, k); }, [&](int k) { return frag(nb_, k); }, bop, nr, a, b);
      out(nt, a);
      out(nt + W, b);
    }
    if (nt < ntiles) {
      f4 a[NF], b[NF];
      const f4 ia = init(nt);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) { a[ft] = ia; b[ft] = splat(0.f); }
      co_ktiles<KT, NF, LK, false, false>(rs, voff, ra, rb, [&](int k) { return frag(nt, k); }, [&](int k) { return frag(nt, k); },
                                          [&](int k) { return frag(nt, k); }, [&](int k) { return frag(nt, k); }, bop, nr, a, b);
      out(nt, a);
    }
  }
};

struct NrFull {
  EDTTS_DEV constexpr int operator()(int) const { return 4; }
};
template <class C>
struct NrHeads {  // k-steps = (head, d-tile): the 8-row remainder tile of a head carries valid k in MFMA steps 0, 1 only
  EDTTS_DEV constexpr int operator()(int kt) const { return (C::DREM && kt % C::DT == C::DT - 1) ? 2 : 4; }
};

// q | k | v^T of a layer from the normalised tile hn: the 3 HT output tiles of the pair-packed stream, split over the waves; stores
// as qkv_tail's (layers/attention.py:91-93).  gm has been preloaded with fr.
template <class C, int W, class GM, class FR>
EDTTS_DEV void coop_qkv(const KArgs& a, __amdgpu_buffer_rsrc_t rs, unsigned voff, GM& gm, FR&& fr, const f4 (&hn)[C::HT][C::NF], int wv, int lane,
                        int b, int m0, bool valid) {
  constexpr int NF = C::NF, HT = C::HT, H = C::H;
  const int fq = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * a.Tp + m0 + fq;
  gm.run(rs, voff, wv, 3 * HT, fr, [&hn](int kt, int ft) { return hn[kt][ft]; }, NrFull{}, [](int) { return splat(0.f); },
         [&](int t, const f4 (&acc)[NF]) {
           if (!valid) return;
           const int which = t / HT, nt = t - which * HT;
           if (which < 2) {
             float* dst = (which == 0 ? a.q_out : a.k_out) + rowbase * H + 16 * nt + 4 * g;
#pragma unroll
             for (int ft = 0; ft < NF; ++ft) __builtin_nontemporal_store(acc[ft], reinterpret_cast<f4*>(dst + (size_t)ft * 16 * H));
           } else {
             float* dst = a.vT_out + ((size_t)b * C::VR + 16 * nt + 4 * g) * a.Tp + m0 + fq;
#pragma unroll
             for (int r = 0; r < 4; ++r)
#pragma unroll
               for (int ft = 0; ft < NF; ++ft) __builtin_nontemporal_store(acc[ft][r], dst + (size_t)r * a.Tp + 16 * ft);
           }
         });
}

template <class C, int TAIL, int W>
EDTTS_DEV void coop_layer_tile(const KArgs& a, char* lds, int wv, int lane, int b, int m0, bool valid) {
  using CO = Coop<C, W>;
  constexpr int NF = C::NF, HT = C::HT, H = C::H, KPT = C::HEADS * C::DT;
  const int fq = lane & 15, g = lane >> 4;
  char* const obuf = lds;                                  // [head][O^T tiles]
  float* const qtile = reinterpret_cast<float*>(lds + CO::O_BYTES);  // [WF][H]
  f4* const act = reinterpret_cast<f4*>(lds) + lane;        // [hidden tile j][ft] register layout (FFN; the attention buffers are dead)
  f4* const xb = reinterpret_cast<f4*>(lds + CO::A_BYTES) + lane;  // [nt][ft] register layout: the phase's output tile
  const size_t rowbase = (size_t)b * a.Tp + m0 + fq;
  float* const hp = a.h + rowbase * H + 4 * g;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.stream);
  const unsigned voff = (unsigned)lane * 16u;
  // fragment offsets of the layer's stream (edtts_pack_weights: proj | q_proj | out_proj | ffn | tail), in fragments
  constexpr unsigned F_PROJ = 0, F_Q = KPT * HT, F_O = F_Q + HT * HT, F_FFN = F_O + KPT * HT;
  const unsigned f_tail = F_FFN + (unsigned)a.ffn_tiles * 3u * HT;
  auto hn_bop = [](const f4 (&hn)[HT][NF]) { return [&hn](int kt, int ft) { return hn[kt][ft]; }; };
  auto x_put = [&](int nt, const f4 (&acc)[NF]) {
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) xb[(nt * NF + ft) * 64] = acc[ft];
  };
  auto o_bop = [&](int kt, int ft) {  // attention output k-tile kt = (head, d-tile) from LDS
    const int hd = kt / C::DT, dt = kt % C::DT;
    const char* ob = obuf + hd * C::OHEAD_BYTES;
    if (C::DREM && dt == C::DT - 1) {
      const f2 t = reinterpret_cast<const f2*>(ob + C::DFULL * NF * 1024 + ft * 512)[lane];
      return f4{t[0], t[1], 0.f, 0.f};
    }
    return reinterpret_cast<const f4*>(ob + (dt * NF + ft) * 1024)[lane];
  };

  // the residual tile: every wave of the tile holds all of it
  f4 h[HT][NF];
#pragma unroll
  for (int nt = 0; nt < HT; ++nt)
#pragma unroll
    for (int ft = 0; ft < NF; ++ft) h[nt][ft] = ldg4(hp + 16 * nt + (size_t)ft * 16 * H);
  WStream<C> no_ring;  // (attention_fused's projection is not used here)
  auto nop = []() {};
  auto add_branch = [&]() {  // h = branch (from LDS) + residual, as k_layer adds them
#pragma unroll
    for (int nt = 0; nt < HT; ++nt)
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) h[nt][ft] = xb[(nt * NF + ft) * 64] + h[nt][ft];
  };

  // ---- x = x + attn(norm1(x, cond))   (transformer.py:142-146) ---------------------------------------------------------------
  {
    QGlobal ql(a.q + ((size_t)b * a.Tp + m0) * H, H, fq, g);
    attention_fused<C, true, O_LDS>(ql, a.k + (size_t)b * a.Tp * H, a.vT + (size_t)b * C::VR * a.Tp, a.Tp, a.T, a.window, m0, lane,
                                    no_ring, h, obuf, nop, nullptr, 0, wv, W);
  }
  {
    auto fr = [&](int nt, int k) { return (F_PROJ + (unsigned)(k * HT + nt)) * 1024u; };
    CoGemm<KPT, NF, W> gm;
    gm.preload(rs, voff, wv, HT, fr);
    __syncthreads();
    gm.run(rs, voff, wv, HT, fr, o_bop, NrHeads<C>{}, [&](int nt) { return ldg4(a.proj_b + 16 * nt + 4 * g); }, x_put);
  }
  // ---- x = x + cross_attn(norm2(x), context)   (transformer.py:151, mla.py:118-194) ------------------------------------------
  {
    // q rows (pair-packed stream: fragment of (n-tile nt, k) at pair nt/2, position 2k + (nt & 1))
    auto fr = [&](int nt, int k) { return (F_Q + (unsigned)((nt >> 1) * 2 * HT + 2 * k + (nt & 1))) * 1024u; };
    CoGemm<HT, NF, W> gm;
    gm.preload(rs, voff, wv, HT, fr);
    __syncthreads();
    add_branch();
    f4 hn[HT][NF];
    rms_norm_tile<HT, NF>(h, a.n2w, nullptr, g, hn);
    gm.run(rs, voff, wv, HT, fr, hn_bop(hn), NrFull{}, [](int) { return splat(0.f); },
                       [&](int nt, const f4 (&acc)[NF]) {
#pragma unroll
                         for (int ft = 0; ft < NF; ++ft) stg4(qtile + (16 * ft + fq) * H + 16 * nt + 4 * g, acc[ft]);
                       });
  }
  __syncthreads();
  {
    QLds ql(qtile, H, fq, g);
    attention_fused<C, false, O_LDS>(ql, a.kc + (size_t)b * a.Sp * H, a.vcT + (size_t)b * C::VR * a.Sp, a.Sp, a.S, -1, m0, lane, no_ring, h,
                                     obuf, nop, nullptr, 0, wv, W);
  }
  {
    auto fr = [&](int nt, int k) { return (F_O + (unsigned)(k * HT + nt)) * 1024u; };
    CoGemm<KPT, NF, W> gm;
    gm.preload(rs, voff, wv, HT, fr);
    __syncthreads();
    gm.run(rs, voff, wv, HT, fr, o_bop, NrHeads<C>{}, [](int) { return splat(0.f); }, x_put);
  }
  // ---- x = x + ffn(norm3(x, cond))   (transformer.py:154-158, :13-49) --------------------------------------------------------
  {
    constexpr int LKU = co_look(NF, HT);
    f4 ua[LKU], ub[LKU];  // fragment ring of the up projections: its first requests fly across the barrier and the norm
    auto up_a = [&](int j) { return [j](int k) { return (F_FFN + (unsigned)j * 3u * HT + 2u * k) * 1024u; }; };
    auto up_b = [&](int j) { return [j](int k) { return (F_FFN + (unsigned)j * 3u * HT + 2u * k + 1u) * 1024u; }; };
    {
      const int j = wv < a.ffn_tiles ? wv : 0;
      co_preload<LKU, true>(rs, voff, up_a(j), up_b(j), ua, ub);
    }
    __syncthreads();
    add_branch();
    f4 hn[HT][NF];
    const float* mod = a.cond + (size_t)b * a.cond_bstride + ((size_t)a.layer * 2 + 1) * 2 * H;
    rms_norm_tile<HT, NF>(h, a.n3w, mod, g, hn);
    // this wave's output tiles of the down projection persist over the groups of hidden tiles: at most ceil(HT / W) of them
    constexpr int MINE = (HT + W - 1) / W;
    f4 dacc[MINE][NF];
#pragma unroll
    for (int i = 0; i < MINE; ++i) {
      const int nt = wv + i * W;
      const f4 db = ldg4(a.down_b + 16 * (nt < HT ? nt : HT - 1) + 4 * g);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) dacc[i][ft] = db;
    }
    for (int j0 = 0; j0 < a.ffn_tiles; j0 += CO::ACT_TILES) {
      const int nj = a.ffn_tiles - j0 < CO::ACT_TILES ? a.ffn_tiles - j0 : CO::ACT_TILES;
      if (j0 > 0) __syncthreads();  // (the previous group's activations have been read by every wave; group 0: the barrier above)
      // up: hidden tile j = value tile j and gate tile j of the interleaved stream, activation -> LDS
      {
        for (int j = j0 + wv; j < j0 + nj; j += W) {
          f4 v[NF], gt[NF];
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) v[ft] = gt[ft] = splat(0.f);
          // behind this tile: the wave's next tile of the group, else its first tile of the next group, else itself (harmless)
          const int jn = j + W < j0 + nj ? j + W : (j0 + CO::ACT_TILES + wv < a.ffn_tiles ? j0 + CO::ACT_TILES + wv : j);
          co_ktiles<HT, NF, LKU, true, true>(rs, voff, ua, ub, up_a(j), up_b(j), up_a(jn), up_b(jn), hn_bop(hn), NrFull{}, v, gt);
          const f4 vb = ldg4(a.up_b + 32 * j + 4 * g), gb = ldg4(a.up_b + 32 * j + 16 + 4 * g);
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) act[((j - j0) * NF + ft) * 64] = swiglu_tile(v[ft], gt[ft], vb, gb);
        }
      }
      // down: h[nt] += W_down[nt, hidden tile j] . act_j for this group's j, ascending (k_layer's order)
      {
        auto dn = [&](int nt) { return [nt, j0](int k) { return (F_FFN + (unsigned)(j0 + k) * 3u * HT + 2u * HT + (unsigned)nt) * 1024u; }; };
        auto act_bop = [&](int k, int ft) { return act[(k * NF + ft) * 64]; };
        f4 dummy[NF];
        if (nj == CO::ACT_TILES) {
          constexpr int LK = co_look(NF, CO::ACT_TILES);
          f4 ra[LK], rb[LK];
          co_preload<LK, false>(rs, voff, dn(wv < HT ? wv : HT - 1), dn(0), ra, rb);  // (weights: requested before the barrier that publishes the activations)
          __syncthreads();
#pragma unroll
          for (int i = 0; i < MINE; ++i) {
            const int nt = wv + i * W;
            if (nt >= HT) continue;
            const int nn = nt + W < HT ? nt + W : nt;
            co_ktiles<CO::ACT_TILES, NF, LK, false, false>(rs, voff, ra, rb, dn(nt), dn(nt), dn(nn), dn(nn), act_bop, NrFull{}, dacc[i], dummy);
          }
        } else {  // a short last group (ffn_mult * hidden / 16 not a multiple of the group): one k-step at a time
          __syncthreads();
#pragma unroll
          for (int i = 0; i < MINE; ++i) {
            const int nt = wv + i * W;
            if (nt >= HT) continue;
            for (int k = 0; k < nj; ++k) {
              f4 ra[1], rb[1];
              auto one = [&](int) { return (F_FFN + (unsigned)(j0 + k) * 3u * HT + 2u * HT + (unsigned)nt) * 1024u; };
              co_preload<1, false>(rs, voff, one, one, ra, rb);
              co_ktiles<1, NF, 1, false, false>(rs, voff, ra, rb, one, one, one, one, [&](int, int ft) { return act[(k * NF + ft) * 64]; }, NrFull{}, dacc[i], dummy);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MINE; ++i)
      if (wv + i * W < HT) x_put(wv + i * W, dacc[i]);
  }
  // ---- tail ----------------------------------------------------------------------------------------------------------------------
  if (TAIL == TAIL_QKV) {
    auto fr = [&](int t, int k) { return (f_tail + (unsigned)((t >> 1) * 2 * HT + 2 * k + (t & 1))) * 1024u; };
    CoGemm<HT, NF, W> gm;
    gm.preload(rs, voff, wv, 3 * HT, fr);
    __syncthreads();
    add_branch();
    if (valid) {  // the residual stream of the next layer: each wave stores its share of the (replicated) tile
#pragma unroll
      for (int nt = 0; nt < HT; ++nt)  // (compile-time register indices: a run-time nt would put the tile into scratch memory)
        if (nt % W == wv) {
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) stg4(hp + 16 * nt + (size_t)ft * 16 * H, h[nt][ft]);
        }
    }
    const float* mod = a.cond + (size_t)b * a.cond_bstride + ((size_t)(a.layer + 1) * 2) * 2 * H;
    rms_norm_tile<HT, NF>(h, a.n1w, mod, g, h);
    coop_qkv<C, W>(a, rs, voff, gm, fr, h, wv, lane, b, m0, valid);
  } else {
    __syncthreads();
    add_branch();
    f4 hn[HT][NF];
    layer_norm_tile<HT, NF>(h, a.fnw, a.fnb, g, hn);
    // final out_proj (n-major stream [nt][k]); even / odd k-tiles in separate accumulators, summed at the end (gemm_phase at NF < 4)
    for (int nt = wv; nt < C::MT; nt += W) {
      f4 e[NF], o[NF];
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) e[ft] = o[ft] = splat(0.f);
      f4 fr[HT];
#pragma unroll
      for (int k = 0; k < HT; ++k) fr[k] = bufld4(rs, voff, (f_tail + (unsigned)(nt * HT + k)) * 1024u);
#pragma unroll
      for (int k = 0; k < HT; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ft = 0; ft < NF; ++ft) {
            if (k & 1) o[ft] = EDTTS_MFMA(fr[k][r], hn[k][ft][r], o[ft]);
            else e[ft] = EDTTS_MFMA(fr[k][r], hn[k][ft][r], e[ft]);
          }
      const f4 ob = ldg4(a.outp_b + 16 * nt + 4 * g);
#pragma unroll
      for (int ft = 0; ft < NF; ++ft) {
        e[ft] += o[ft];
        const int f = m0 + 16 * ft + fq;
        if (!valid || f >= a.T) continue;
        const size_t idx = ((size_t)b * a.T + f) * C::MEL + 16 * nt + 4 * g;
        tail_apply<TAIL>(a, idx, e[ft] + ob);
      }
    }
  }
}

template <class C, int TAIL, int W>
__global__ __launch_bounds__(64 * W) void k_layer_co(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_co[];
  using CO = Coop<C, W>;
  static_assert(CO::FITS, "cooperative tile state must fit the CU's LDS");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tpu = a.Tp / C::WF, ntiles = a.B * tpu;
  int tile = remap_block(blockIdx.x, gridDim.x) * CO::TILES + wave / W;
  // a block's surplus tile (odd tile count) recomputes the last tile and stores nothing: its waves must keep the block's barrier count
  const bool valid = tile < ntiles;
  tile = valid ? tile : ntiles - 1;
  const int b = tile / tpu;
  coop_layer_tile<C, TAIL, W>(a, smem_co + (wave / W) * CO::TILE_BYTES, wave % W, lane, b, (tile - b * tpu) * C::WF, valid);
}

// The prologue the same way: h = in_proj(x) + pe (decoder.py:96-97) by output tiles, exchanged through LDS; AdaRMSNorm(layer 0)
// replicated; QKV(layer 0) by output tiles.  Stream: inp (n-tile pairs over the MT k-tiles) | qkv(0).
template <class C, int W>
__global__ __launch_bounds__(64 * W) void k_prologue_co(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_co[];
  constexpr int NF = C::NF, HT = C::HT, MT = C::MT, H = C::H;
  const int lane = threadIdx.x & 63, fq = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tpu = a.Tp / C::WF, ntiles = a.B * tpu;
  int tile = remap_block(blockIdx.x, gridDim.x);
  const bool valid = tile < ntiles;
  tile = valid ? tile : ntiles - 1;
  const int b = tile / tpu, m0 = (tile - b * tpu) * C::WF;
  f4* const xb = reinterpret_cast<f4*>(smem_co) + lane;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.stream);
  const unsigned voff = (unsigned)lane * 16u;
  auto fin = [&](int nt, int k) { return (unsigned)((nt >> 1) * 2 * MT + 2 * k + (nt & 1)) * 1024u; };
  CoGemm<MT, NF, W> gi;
  gi.preload(rs, voff, wv, HT, fin);
  f4 xin[MT][NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    const int f = m0 + 16 * ft + fq;
#pragma unroll
    for (int t = 0; t < MT; ++t) xin[t][ft] = f < a.T ? ldg4(a.x + ((size_t)b * a.T + f) * C::MEL + 16 * t + 4 * g) : splat(0.f);
  }
  gi.run(rs, voff, wv, HT, fin, [&xin](int kt, int ft) { return xin[kt][ft]; }, NrFull{}, [&](int nt) { return ldg4(a.inp_b + 16 * nt + 4 * g); },
         [&](int nt, const f4 (&acc)[NF]) {
#pragma unroll
           for (int ft = 0; ft < NF; ++ft) xb[(nt * NF + ft) * 64] = acc[ft];
         });
  auto fq_ = [&](int t, int k) { return (unsigned)(HT * MT + (t >> 1) * 2 * HT + 2 * k + (t & 1)) * 1024u; };
  CoGemm<HT, NF, W> gm;
  gm.preload(rs, voff, wv, 3 * HT, fq_);
  __syncthreads();
  f4 h[HT][NF];
#pragma unroll
  for (int ft = 0; ft < NF; ++ft) {
    int f = m0 + 16 * ft + fq;
    f = f < a.max_pos ? f : a.max_pos - 1;
#pragma unroll
    for (int nt = 0; nt < HT; ++nt) h[nt][ft] = xb[(nt * NF + ft) * 64] + ldg4(a.pe + (size_t)f * H + 16 * nt + 4 * g);  // embeddings.py:142
  }
  if (valid) {
    float* hp = a.h + ((size_t)b * a.Tp + m0 + fq) * H + 4 * g;
#pragma unroll
    for (int nt = 0; nt < HT; ++nt)
      if (nt % W == wv) {
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) stg4(hp + 16 * nt + (size_t)ft * 16 * H, h[nt][ft]);
      }
  }
  const float* mod = a.cond + (size_t)b * a.cond_bstride;  // layer 0, norm1
  rms_norm_tile<HT, NF>(h, a.n1w, mod, g, h);
  coop_qkv<C, W>(a, rs, voff, gm, fq_, h, wv, lane, b, m0, valid);
}

}  // namespace edtts
