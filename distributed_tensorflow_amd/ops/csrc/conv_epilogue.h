// Epilogue of the implicit-GEMM convolution kernels (shared by conv_igemm.hip and conv_pipe.hip).
//
// A lane of a wave that owns a WM x WN sub-tile holds, for each of its MR pixels (MFMA column
// fr = lane & 15 of pixel block mr), CPL = 4*NR consecutive output channels starting at
// co0 = n0 + wn*WN + (lane >> 4)*CPL: accumulator acc[nr][mr][e] is channel co0 + 4*nr + e.
// The epilogue adds bias / addend (optionally gated by a 1-bit mask), applies the folded
// BatchNorm of the inference flavour, stores 16 bytes per lane and accumulates the fused
// BatchNorm statistics (forward: sum, sum of squares; backward: sum g, sum g*y).
#pragma once
#include "common.h"
#include "conv_params.h"

namespace conv_epi {

// EPI_ADD: the plain epilogue for launches with an addend (no statistics): addend loads run one
// pixel block ahead of the stores
constexpr int EPI_PLAIN = 0, EPI_BSTATS = 1, EPI_INFER = 2, EPI_ADD = 3;

// Value of lane ^ (1 << T) inside a row of 16 lanes, without a trip through the LDS crossbar
// where the data-parallel-primitive modifiers can express the exchange: quad permutes for the
// neighbours at distance 1 and 2, a rotation by 8 for distance 8; distance 4 is a swizzle.
// Every lane of the wave must be active.
template <int T>
__device__ __forceinline__ float lane_xor16(float v) {
  const int i = __builtin_bit_cast(int, v);
  int r;
  if constexpr (T == 0) r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
  else if constexpr (T == 1) r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  else if constexpr (T == 2) r = __builtin_amdgcn_ds_swizzle(i, 0x101F);                     // lane ^ 4
  else r = __builtin_amdgcn_update_dpp(i, i, 0x128, 0xF, 0xF, true);                         // row_ror:8
  return __builtin_bit_cast(float, r);
}

// Sum over the 16 lanes that share lane/16 (the 16 pixels of an MFMA column block), leaving
// the totals scattered over those lanes: a halving butterfly - at step T a lane keeps one
// half of its values and hands the other half to lane ^ (1 << T) - so NV values cost NV - 1
// exchanges in total instead of 4 * NV.  Afterwards v[0 .. (NV >> steps) - 1] of a lane hold the
// totals of the original indices  base + j,  base = sum_T bit_T(lane) * (NV >> (T + 1)).
// Steps T .. TEND - 1 are run; N is the number of values still held when step T starts.
// DPP selects how values cross lanes: the modifiers / swizzle of lane_xor16, or the LDS crossbar
// (`__shfl_xor`).  The forward epilogue keeps the crossbar: with the modifiers the compiler's
// schedule of its 31 exchanges next to 64 live accumulators overflows the 128-register budget.
template <int NV, int N, int T, int TEND = 4, bool DPP = false>
__device__ __forceinline__ void butterfly16(float (&v)[NV], int fr) {
  if constexpr (T < TEND) {
    if constexpr (N > 1) {
      constexpr int H = N / 2;
      const bool up = (fr >> T) & 1;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const float send = up ? v[i] : v[i + H];
        const float keep = up ? v[i + H] : v[i];
        v[i] = keep + (DPP ? lane_xor16<T>(send) : __shfl_xor(send, 1 << T));
      }
      butterfly16<NV, H, T + 1, TEND, DPP>(v, fr);
    } else {
      v[0] += DPP ? lane_xor16<T>(v[0]) : __shfl_xor(v[0], 1 << T);
      butterfly16<NV, 1, T + 1, TEND, DPP>(v, fr);
    }
  }
}


// Epilogue of a data gradient that also accumulates the backward reductions of the BatchNorm
// whose output gradient it writes (sum g, sum g*y per channel; g = the stored value gated by the
// layer's ReLU mask, y = the layer's input).  Bias-free (host check).
//
// A software pipeline over the wave's MR pixel blocks, one block deep: the loads of block mr + 1
// (y and its mask bits, addend and its mask bits) are issued before block mr is stored - written
// behind the stores they could not be hoisted over them (the tensors may alias for all the
// compiler knows) and every wave would pay a memory latency per block - and nothing of a block
// stays live once it is done:
//   * a pixel's channels of a lane are stored back to back in 16-byte pieces (splitting a pixel's
//     bytes over two passes costs ~28 % on the bandwidth-bound layers: partial lines);
//   * the rounded, gated values of a piece and their products with y go through the 16-lane
//     butterfly at once (data-parallel-primitive adds, no LDS crossbar), so a lane carries one
//     running sum per piece from block to block instead of 2 * CPL.
// Registers: the accumulators, two sets of prefetched operands and those sums fit the
// 128-register budget of the tiles that run four waves per SIMD.
template <int MR, int NR, bool POINTWISE, int DT>
__device__ __forceinline__ void conv_epilogue_bstats(const ConvParams& p, f32x4 (&acc)[NR][MR],
                                                     const int m_base, const int co0, const int mt,
                                                     const int lane) {
  typedef typename Elem<DT>::T E;
  const int fr = lane & 15;
  constexpr int CPL = 4 * NR;
  constexpr int W = CPL >= 8 ? 8 : 4;              // channels per piece (one store)
  constexpr int HV = CPL / W;
  constexpr int TPG = W / 4;                       // MFMA tiles per piece
  constexpr int NV = 2 * W;                        // sums of a piece: [0, W) sum g, [W, 2 W) sum g*y
  constexpr int HALVINGS = NV >= 16 ? 4 : 3;       // 16 values: one total left per lane; 8: per two lanes
  struct Operands {
    uint32_t opix;                                 // output pixel of this lane; ~0: past the last row
    u32x4 y[HV], a[HV];
    uint32_t bits;                                 // per piece h: ReLU bits in byte h, addend bits in byte 2 + h
  };
  auto fetch = [&](const int mr, Operands& q) {
    const int m = m_base + mr * 16 + fr;
    uint32_t opix = ~0u;
    if (m < p.M) {
      if (POINTWISE) {
        opix = (uint32_t)m;
      } else {
        const uint32_t t = fd_div((uint32_t)m, p.divOW);
        const int ow = m - (int)t * p.OW;
        const uint32_t n = fd_div(t, p.divOH);
        const int oh = (int)t - (int)n * p.OH;
        opix = (uint32_t)(((int)n * p.OHfull + oh * p.OSH + p.OOH) * p.OWfull + (ow * p.OSW + p.OOW));
      }
    }
    q.opix = opix;
    const size_t e = (size_t)opix * p.ldc + co0;
    const size_t mb = (size_t)opix * (p.ldc >> 3) + (co0 >> 3);
    uint32_t bits = 0;
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      q.y[h] = u32x4{0u, 0u, 0u, 0u};
      q.a[h] = u32x4{0u, 0u, 0u, 0u};
      if (opix == ~0u || co0 + h * W >= p.Cout) continue;
      const E* yp = reinterpret_cast<const E*>(p.stats_y) + e + h * W;
      if constexpr (W == 8) {
        q.y[h] = *reinterpret_cast<const u32x4*>(yp);
      } else {
        const u32x2 t = *reinterpret_cast<const u32x2*>(yp);
        q.y[h].x = t.x; q.y[h].y = t.y;
      }
      // W == 4: the lane's four channels are one nibble of the mask byte
      uint32_t sb = 0xffu, ab = 0xffu;
      if (p.stats_mask) sb = ((uint32_t)p.stats_mask[mb + h] >> (W == 4 ? (co0 & 4) : 0)) & 0xffu;
      if (p.addend) {
        const E* ap = reinterpret_cast<const E*>(p.addend) + e + h * W;
        if constexpr (W == 8) {
          q.a[h] = *reinterpret_cast<const u32x4*>(ap);
        } else {
          const u32x2 t = *reinterpret_cast<const u32x2*>(ap);
          q.a[h].x = t.x; q.a[h].y = t.y;
        }
        if (p.addend_mask) ab = ((uint32_t)p.addend_mask[mb + h] >> (W == 4 ? (co0 & 4) : 0)) & 0xffu;
      }
      bits |= (sb << (8 * h)) | (ab << (16 + 8 * h));
    }
    q.bits = bits;
  };
  float st[HV];                                    // this lane's share of the piece's 2 W sums
#pragma unroll
  for (int h = 0; h < HV; ++h) st[h] = 0.f;
  Operands q[2];
  fetch(0, q[0]);
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    if (mr + 1 < MR) fetch(mr + 1, q[(mr + 1) & 1]);
    const Operands& c = q[mr & 1];
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      float x[W];
#pragma unroll
      for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[t * 4 + e] = acc[h * TPG + t][mr][e];
      const uint32_t aw[4] = {c.a[h].x, c.a[h].y, c.a[h].z, c.a[h].w};
      const uint32_t yw[4] = {c.y[h].x, c.y[h].y, c.y[h].z, c.y[h].w};
      const uint32_t bm = c.bits >> (16 + 8 * h); // addend: only the elements whose bit is set
      const uint32_t sb = c.bits >> (8 * h);      // g: the rounded value where the ReLU was open
      uint32_t w[W / 2];
#pragma unroll
      for (int j = 0; j < W / 2; ++j) {
        const uint32_t a = aw[j] & (((bm >> (2 * j)) & 1u ? 0xffffu : 0u) |
                                    ((bm >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u));
        w[j] = pack2<DT>(x[2 * j] + lo16<DT>(a), x[2 * j + 1] + hi16<DT>(a));
      }
      const bool live = c.opix != ~0u && co0 + h * W < p.Cout;
      if (live) {
        E* optr = reinterpret_cast<E*>(p.out) + ((size_t)c.opix * p.ldc + co0 + h * W);
        if constexpr (W == 8) {
          *reinterpret_cast<u32x4*>(optr) = u32x4{w[0], w[1], w[2], w[3]};
        } else {
          u32x2 o;
          o.x = w[0]; o.y = w[1];
          *reinterpret_cast<u32x2*>(optr) = o;
        }
      }
      float v[NV];                                 // [0, W): g, [W, 2 W): g * y of this piece
#pragma unroll
      for (int j = 0; j < W / 2; ++j) {
        const float lo = (live && ((sb >> (2 * j)) & 1u)) ? lo16<DT>(w[j]) : 0.f;
        const float hi = (live && ((sb >> (2 * j + 1)) & 1u)) ? hi16<DT>(w[j]) : 0.f;
        v[2 * j] = lo;
        v[2 * j + 1] = hi;
        v[W + 2 * j] = lo * lo16<DT>(yw[j]);
        v[W + 2 * j + 1] = hi * hi16<DT>(yw[j]);
      }
      butterfly16<NV, NV, 0, 4, true>(v, fr);      // every lane takes part (no early exit above)
      st[h] += v[0];
    }
  }
  if (p.stats == nullptr) return;
  int base = 0;
#pragma unroll
  for (int t = 0; t < HALVINGS; ++t) base += ((fr >> t) & 1) * (NV >> (t + 1));
  if (HALVINGS == 4 || (fr >> 3) == 0) {
    float* dst = p.stats + (size_t)(mt % p.stats_replicas) * 2 * p.Cout;
    const int qn = base / W, cn = base % W;        // quantity (sum g / sum g*y), channel in the piece
#pragma unroll
    for (int h = 0; h < HV; ++h)
      if (co0 + h * W + cn < p.Cout) atomicAdd(dst + qn * p.Cout + co0 + h * W + cn, st[h]);
  }
}

// m_base: first pixel row of this wave's sub-tile (m0 + wm*WM); co0: see above; mt: M-tile
// index of the block (selects the statistics replica).
template <bool POINTWISE>
__device__ __forceinline__ size_t out_pixel_index(const ConvParams& p, int m) {
  if (POINTWISE) return (size_t)m;
  const uint32_t t = fd_div((uint32_t)m, p.divOW);
  const int ow = m - (int)t * p.OW;
  const uint32_t n = fd_div(t, p.divOH);
  const int oh = (int)t - (int)n * p.OH;
  return (size_t)((int)n * p.OHfull + oh * p.OSH + p.OOH) * p.OWfull + (ow * p.OSW + p.OOW);
}

// Addend pieces and mask bytes of one pixel block of a lane, requested ahead of their use
// (EPI_ADD): the kernels ask for block 0 before their K loop - the addend does not depend on the
// product - and the epilogue for block mr + 1 before it stores block mr.
template <int NR, int EPI>
struct AddendAhead {
  static constexpr bool ON = EPI == EPI_ADD && NR >= 2;
  static constexpr int HV = ON ? NR / 2 : 1;       // 16-byte pieces of a lane's 4 NR channels
  u32x4 a[HV];
  uint32_t b[HV];
};

template <int NR, bool POINTWISE, int EPI, int DT>
__device__ __forceinline__ void addend_fetch(const ConvParams& p, AddendAhead<NR, EPI>& pre, const int m,
                                             const int co0) {
  typedef typename Elem<DT>::T E;
  constexpr int HV = AddendAhead<NR, EPI>::HV;
  if constexpr (AddendAhead<NR, EPI>::ON) {
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      pre.a[h] = u32x4{0u, 0u, 0u, 0u};
      pre.b[h] = 0xffu;
    }
    if (p.addend != nullptr && p.ldc % 8 == 0 && m < p.M) {
      const size_t opix = out_pixel_index<POINTWISE>(p, m);
      const E* aptr = reinterpret_cast<const E*>(p.addend) + opix * p.ldc + co0;
#pragma unroll
      for (int h = 0; h < HV; ++h)
        if (co0 + h * 8 + 8 <= p.Cout) pre.a[h] = *reinterpret_cast<const u32x4*>(aptr + h * 8);
      if (p.addend_mask && co0 + 8 <= p.Cout) {
        // the lane's HV mask bytes in one load (co0 is a multiple of 8 HV channels)
        const uint8_t* mp = p.addend_mask + opix * (p.ldc >> 3) + (co0 >> 3);
        uint32_t bits;
        if constexpr (HV == 4) {
          bits = co0 + 32 <= p.Cout ? *reinterpret_cast<const uint32_t*>(mp)
                                    : (uint32_t)mp[0] | (co0 + 16 <= p.Cout ? (uint32_t)mp[1] << 8 : 0u) |
                                          (co0 + 24 <= p.Cout ? (uint32_t)mp[2] << 16 : 0u);
        } else if constexpr (HV == 2) {
          bits = co0 + 16 <= p.Cout ? (uint32_t)*reinterpret_cast<const uint16_t*>(mp) : (uint32_t)mp[0];
        } else {
          bits = mp[0];
        }
#pragma unroll
        for (int h = 0; h < HV; ++h) pre.b[h] = (bits >> (8 * h)) & 0xffu;
      }
    }
  }
}

// `first`: the addend of the lane's pixel block 0, requested by the caller before its K loop
// (EPI_ADD; null: requested here).
template <int MR, int NR, bool POINTWISE, int EPI, int DT = DT_BF16>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[NR][MR],
                                              const int m_base, const int co0, const int mt,
                                              const int lane, const AddendAhead<NR, EPI>* first = nullptr) {
  static_assert(DT == DT_BF16 || DT == DT_F16, "the fused epilogue packs two 16-bit elements per word");
  if constexpr (EPI == EPI_BSTATS) {
    conv_epilogue_bstats<MR, NR, POINTWISE, DT>(p, acc, m_base, co0, mt, lane);
    return;
  }
  typedef typename Elem<DT>::T E;
  constexpr bool BSTATS = EPI == EPI_BSTATS;
  const int fr = lane & 15;
  // ---- epilogue: lane holds 4*NR consecutive channels of pixel (mr, fr) ---------------------
  constexpr int CPL = 4 * NR;                      // channels per lane
  float bias[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c)
    bias[c] = (p.bias != nullptr && co0 + c < p.Cout) ? p.bias[co0 + c] : 0.f;
  if constexpr (EPI == EPI_INFER) {
    // tile nr of a lane holds channels co0 + 4*nr .. co0 + 4*nr + 3
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      float sc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        sc[e] = (co0 + nr * 4 + e < p.Cout) ? p.scale[co0 + nr * 4 + e] : 0.f;
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nr][mr][e] *= sc[e];
    }
  }
  const bool do_relu = EPI == EPI_INFER && p.relu != 0;
  const bool vec_ok = (p.ldc % 8 == 0);
  // fused batch-norm statistics: [0, CPL) sums, [CPL, 2*CPL) sums of squares of this lane's
  // channels over its MR pixels (the host guarantees the vector store path in that case)
  float st[2 * CPL];
#pragma unroll
  for (int c = 0; c < 2 * CPL; ++c) st[c] = 0.f;
  auto out_pixel = [&](int m) -> size_t { return out_pixel_index<POINTWISE>(p, m); };
  // The addend (and its mask bits) of pixel block mr + 1 is requested before block mr is stored:
  // written inside the store loop the loads cannot be hoisted over the stores (an accumulating
  // data gradient adds to the very tensor it writes) and a wave pays a memory latency per block.
  constexpr bool AHEAD = AddendAhead<NR, EPI>::ON;
  constexpr int HV = AddendAhead<NR, EPI>::HV;
  AddendAhead<NR, EPI> pre;
  auto fetch_addend = [&](int mr) {
    addend_fetch<NR, POINTWISE, EPI, DT>(p, pre, m_base + mr * 16 + fr, co0);
  };
  if (first != nullptr) pre = *first;
  else fetch_addend(0);
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    u32x4 acur[HV];
    uint32_t bcur[HV];
    if constexpr (AHEAD) {
#pragma unroll
      for (int h = 0; h < HV; ++h) {
        acur[h] = pre.a[h];
        bcur[h] = pre.b[h];
      }
      if (mr + 1 < MR) fetch_addend(mr + 1);
    }
    const int m = m_base + mr * 16 + fr;
    if (m >= p.M) continue;
    const size_t opix = out_pixel(m);
    const size_t obase = opix * p.ldc;
    E* optr = reinterpret_cast<E*>(p.out) + obase + co0;
    const E* aptr = p.addend ? reinterpret_cast<const E*>(p.addend) + obase + co0 : nullptr;
    const uint8_t* mptr = p.addend_mask ? p.addend_mask + opix * (p.ldc >> 3) + (co0 >> 3) : nullptr;
    if constexpr (CPL >= 8) {
#pragma unroll
      for (int h = 0; h < CPL / 8; ++h) {
        // channels h*8 .. h*8+7 of this lane = MFMA tiles nr = 2h, 2h+1
        float v0 = acc[2 * h][mr][0] + bias[h * 8 + 0], v1 = acc[2 * h][mr][1] + bias[h * 8 + 1];
        float v2 = acc[2 * h][mr][2] + bias[h * 8 + 2], v3 = acc[2 * h][mr][3] + bias[h * 8 + 3];
        float v4 = acc[2 * h + 1][mr][0] + bias[h * 8 + 4], v5 = acc[2 * h + 1][mr][1] + bias[h * 8 + 5];
        float v6 = acc[2 * h + 1][mr][2] + bias[h * 8 + 6], v7 = acc[2 * h + 1][mr][3] + bias[h * 8 + 7];
        if (vec_ok && co0 + h * 8 + 8 <= p.Cout) {
          if (aptr) {
            u32x4 a;
            if constexpr (AHEAD) a = acur[h];
            else a = *reinterpret_cast<const u32x4*>(aptr + h * 8);
            if (mptr) {                            // keep only the elements whose mask bit is set
              const uint32_t b = AHEAD ? bcur[h] : (uint32_t)mptr[h];
              a.x &= ((b & 1u) ? 0xffffu : 0u) | ((b & 2u) ? 0xffff0000u : 0u);
              a.y &= ((b & 4u) ? 0xffffu : 0u) | ((b & 8u) ? 0xffff0000u : 0u);
              a.z &= ((b & 16u) ? 0xffffu : 0u) | ((b & 32u) ? 0xffff0000u : 0u);
              a.w &= ((b & 64u) ? 0xffffu : 0u) | ((b & 128u) ? 0xffff0000u : 0u);
            }
            v0 += lo16<DT>(a.x); v1 += hi16<DT>(a.x);
            v2 += lo16<DT>(a.y); v3 += hi16<DT>(a.y);
            v4 += lo16<DT>(a.z); v5 += hi16<DT>(a.z);
            v6 += lo16<DT>(a.w); v7 += hi16<DT>(a.w);
            // The eight sums stay eight separate registers up to the conversions.  Left alone the
            // compiler pairs them as v_pk_add_f32 with operand selects and regroups the pairs with
            // v_pk_mov_b32 one wait state later; in the whole-rows kernels that sequence lost the
            // addend of a lane's last channel in lanes 48 - 63 - ~2 500 of 51 million elements at
            // batch 256, different ones from launch to launch (scripts/debug/rows_stress.py).
            asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
          }
          if (do_relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
            v4 = fmaxf(v4, 0.f); v5 = fmaxf(v5, 0.f); v6 = fmaxf(v6, 0.f); v7 = fmaxf(v7, 0.f);
          }
          u32x4 o;
          o.x = pack2<DT>(v0, v1); o.y = pack2<DT>(v2, v3); o.z = pack2<DT>(v4, v5); o.w = pack2<DT>(v6, v7);
          *reinterpret_cast<u32x4*>(optr + h * 8) = o;
          if (EPI != EPI_INFER && EPI != EPI_ADD && p.stats) {
            const uint32_t w[4] = {o.x, o.y, o.z, o.w};
            if constexpr (BSTATS) {                  // sum g, sum g * y  (g = masked output)
              const u32x4 yq = *reinterpret_cast<const u32x4*>(reinterpret_cast<const E*>(p.stats_y) + obase + co0 + h * 8);
              const uint32_t yw[4] = {yq.x, yq.y, yq.z, yq.w};
              const uint32_t sb = p.stats_mask
                  ? p.stats_mask[opix * (p.ldc >> 3) + (co0 >> 3) + h] : 0xffu;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float lo = (sb >> (2 * j)) & 1u ? lo16<DT>(w[j]) : 0.f;
                const float hi = (sb >> (2 * j + 1)) & 1u ? hi16<DT>(w[j]) : 0.f;
                st[h * 8 + 2 * j] += lo;
                st[h * 8 + 2 * j + 1] += hi;
                st[CPL + h * 8 + 2 * j] += lo * lo16<DT>(yw[j]);
                st[CPL + h * 8 + 2 * j + 1] += hi * hi16<DT>(yw[j]);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float lo = lo16<DT>(w[j]), hi = hi16<DT>(w[j]);
                st[h * 8 + 2 * j] += lo;
                st[h * 8 + 2 * j + 1] += hi;
                st[CPL + h * 8 + 2 * j] += lo * lo;
                st[CPL + h * 8 + 2 * j + 1] += hi * hi;
              }
            }
          }
        } else {
          const float vv[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            if (co0 + h * 8 + c < p.Cout) {
              float x = vv[c];
              if (aptr) x += to_f<DT>(aptr[h * 8 + c]);
              optr[h * 8 + c] = from_f<DT>(do_relu ? fmaxf(x, 0.f) : x);
            }
          }
        }
      }
    } else {
      // CPL == 4: one 8-byte store
      float v0 = acc[0][mr][0] + bias[0], v1 = acc[0][mr][1] + bias[1];
      float v2 = acc[0][mr][2] + bias[2], v3 = acc[0][mr][3] + bias[3];
      if (p.ldc % 4 == 0 && co0 + 4 <= p.Cout) {
        if (aptr) {
          u32x2 a = *reinterpret_cast<const u32x2*>(aptr);
          if (mptr) {                              // this lane's 4 channels: one nibble of the byte
            const uint32_t b = (uint32_t)mptr[0] >> (co0 & 4);
            a.x &= ((b & 1u) ? 0xffffu : 0u) | ((b & 2u) ? 0xffff0000u : 0u);
            a.y &= ((b & 4u) ? 0xffffu : 0u) | ((b & 8u) ? 0xffff0000u : 0u);
          }
          v0 += lo16<DT>(a.x); v1 += hi16<DT>(a.x);
          v2 += lo16<DT>(a.y); v3 += hi16<DT>(a.y);
        }
        if (do_relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        u32x2 o;
        o.x = pack2<DT>(v0, v1);
        o.y = pack2<DT>(v2, v3);
        *reinterpret_cast<u32x2*>(optr) = o;
        if (EPI != EPI_INFER && EPI != EPI_ADD && p.stats) {
          const uint32_t w[2] = {o.x, o.y};
          if constexpr (BSTATS) {
            const u32x2 yq = *reinterpret_cast<const u32x2*>(reinterpret_cast<const E*>(p.stats_y) + obase + co0);
            const uint32_t yw[2] = {yq.x, yq.y};
            const uint32_t sb = p.stats_mask
                ? (uint32_t)p.stats_mask[opix * (p.ldc >> 3) + (co0 >> 3)] >> (co0 & 4) : 0xffu;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float lo = (sb >> (2 * j)) & 1u ? lo16<DT>(w[j]) : 0.f;
              const float hi = (sb >> (2 * j + 1)) & 1u ? hi16<DT>(w[j]) : 0.f;
              st[2 * j] += lo;
              st[2 * j + 1] += hi;
              st[CPL + 2 * j] += lo * lo16<DT>(yw[j]);
              st[CPL + 2 * j + 1] += hi * hi16<DT>(yw[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float lo = lo16<DT>(w[j]), hi = hi16<DT>(w[j]);
              st[2 * j] += lo;
              st[2 * j + 1] += hi;
              st[CPL + 2 * j] += lo * lo;
              st[CPL + 2 * j + 1] += hi * hi;
            }
          }
        }
      } else {
        const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (co0 + c < p.Cout) {
            float x = vv[c];
            if (aptr) x += to_f<DT>(aptr[c]);
            optr[c] = from_f<DT>(do_relu ? fmaxf(x, 0.f) : x);
          }
        }
      }
    }
  }
  if (EPI != EPI_INFER && EPI != EPI_ADD && p.stats) {
    constexpr int NV = 2 * CPL;
    constexpr int HALVINGS = NV >= 16 ? 4 : 3;     // NV is 8, 16 or 32
    constexpr int LEFT = NV >> HALVINGS;           // totals per lane afterwards
    butterfly16<NV, NV, 0>(st, fr);
    int base = 0;
#pragma unroll
    for (int t = 0; t < HALVINGS; ++t) base += ((fr >> t) & 1) * (NV >> (t + 1));
    if (HALVINGS == 4 || (fr >> 3) == 0) {
      float* dst = p.stats + (size_t)(mt % p.stats_replicas) * 2 * p.Cout;
#pragma unroll
      for (int j = 0; j < LEFT; ++j) {
        const int idx = base + j;
        const int q = idx / CPL, c = idx % CPL;    // quantity (sum / sum of squares), channel
        if (co0 + c < p.Cout) atomicAdd(dst + q * p.Cout + co0 + c, st[j]);
      }
    }
  }
}

// Epilogue for the element types without fused extras (fp16, fp32; bf16 takes conv_epilogue):
// out = acc (+ bias) (+ addend, optionally gated by the 1-bit mask), four channels per store.
// With `p.stats` it also accumulates the BatchNorm statistics of the stored values (sum and sum of
// squares per channel into replica row `mt % stats_replicas`), as conv_epilogue does.
template <int DT, int MR, int NR, bool POINTWISE>
__device__ __forceinline__ void conv_epilogue_plain(const ConvParams& p, f32x4 (&acc)[NR][MR],
                                                    const int m_base, const int co0, const int mt,
                                                    const int lane) {
  typedef typename Elem<DT>::T T;
  const int fr = lane & 15;
  T* out = reinterpret_cast<T*>(p.out);
  const T* addend = reinterpret_cast<const T*>(p.addend);
  const bool vec_ok = (p.ldc % 4 == 0);
  const bool do_stats = p.stats != nullptr;
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int m = m_base + mr * 16 + fr;
    if (m >= p.M) continue;
    size_t opix;
    if (POINTWISE) {
      opix = (size_t)m;
    } else {
      const uint32_t t = fd_div((uint32_t)m, p.divOW);
      const int ow = m - (int)t * p.OW;
      const uint32_t n = fd_div(t, p.divOH);
      const int oh = (int)t - (int)n * p.OH;
      opix = (size_t)((int)n * p.OHfull + oh * p.OSH + p.OOH) * p.OWfull + (ow * p.OSW + p.OOW);
    }
    const size_t obase = opix * p.ldc;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int c0 = co0 + nr * 4;                 // this tile's four channels
      if (c0 >= p.Cout) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        v[e] = acc[nr][mr][e] + ((p.bias && c0 + e < p.Cout) ? p.bias[c0 + e] : 0.f);
      const uint32_t mbits = p.addend_mask
          ? ((uint32_t)p.addend_mask[opix * (p.ldc >> 3) + (c0 >> 3)] >> (c0 & 4)) : 0xfu;
      if (vec_ok && c0 + 4 <= p.Cout) {
        if (addend) {
          float a[4];
          if constexpr (DT == DT_F32) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(addend + obase + c0);
            a[0] = q[0]; a[1] = q[1]; a[2] = q[2]; a[3] = q[3];
          } else {
            const u32x2 q = *reinterpret_cast<const u32x2*>(addend + obase + c0);
            a[0] = to_f<DT>((uint16_t)(q.x & 0xffffu)); a[1] = to_f<DT>((uint16_t)(q.x >> 16));
            a[2] = to_f<DT>((uint16_t)(q.y & 0xffffu)); a[3] = to_f<DT>((uint16_t)(q.y >> 16));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (mbits >> e) & 1u ? a[e] : 0.f;
        }
        if constexpr (DT == DT_F32) {
          *reinterpret_cast<f32x4*>(out + obase + c0) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          u32x2 o;
          o.x = pack2<DT>(v[0], v[1]);
          o.y = pack2<DT>(v[2], v[3]);
          *reinterpret_cast<u32x2*>(out + obase + c0) = o;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (c0 + e < p.Cout) {
            float x = v[e];
            if (addend && ((mbits >> e) & 1u)) x += to_f<DT>(addend[obase + c0 + e]);
            out[obase + c0 + e] = from_f<DT>(x);
          }
        }
      }
    }
  }
  // Statistics in a pass of their own over the accumulators, one MFMA tile of four channels at a
  // time: eight running sums instead of 8 * NR next to the accumulators (the fp32 tiles sit at
  // their register budget; carrying all sums through the store loop spilled thousands of them).
  // Forward statistics go with a plain convolution (no addend: host check), so the stored value is
  // the accumulator plus bias, rounded to the element type.
  if (do_stats) {
    float* dst = p.stats + (size_t)(mt % p.stats_replicas) * 2 * p.Cout;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const int c0 = co0 + nr * 4;
      float s8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) s8[e] = 0.f;
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        if (m_base + mr * 16 + fr < p.M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[nr][mr][e] + ((p.bias && c0 + e < p.Cout) ? p.bias[c0 + e] : 0.f);
            if constexpr (DT != DT_F32) x = to_f<DT>(from_f<DT>(x));
            s8[e] += x;
            s8[4 + e] += x * x;
          }
        }
      }
      butterfly16<8, 8, 0>(s8, fr);      // three halvings, then the two halves of the 16 lanes are added
      const int idx = ((fr & 1) << 2) | (((fr >> 1) & 1) << 1) | ((fr >> 2) & 1);
      const int q = idx >> 2, c = idx & 3;           // quantity (sum / sum of squares), channel
      if ((fr >> 3) == 0 && c0 + c < p.Cout) atomicAdd(dst + q * p.Cout + c0 + c, s8[0]);
    }
  }
}

}  // namespace conv_epi
