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

constexpr int EPI_PLAIN = 0, EPI_BSTATS = 1, EPI_INFER = 2;

// Sum over the 16 lanes that share lane/16 (the 16 pixels of an MFMA column block), leaving
// the totals scattered over those lanes: a halving butterfly - at step T a lane keeps one
// half of its values and hands the other half to lane ^ (1 << T) - so NV values cost NV - 1
// shuffles in total instead of 4 * NV.  Afterwards v[0 .. (NV >> steps) - 1] of a lane hold the
// totals of the original indices  base + j,  base = sum_T bit_T(lane) * (NV >> (T + 1)).
template <int NV, int N, int T>
__device__ __forceinline__ void butterfly16(float (&v)[NV], int fr) {
  if constexpr (T < 4) {
    if constexpr (N > 1) {
      constexpr int H = N / 2;
      const bool up = (fr >> T) & 1;
#pragma unroll
      for (int i = 0; i < H; ++i) {
        const float send = up ? v[i] : v[i + H];
        const float keep = up ? v[i + H] : v[i];
        v[i] = keep + __shfl_xor(send, 1 << T);
      }
      butterfly16<NV, H, T + 1>(v, fr);
    } else {
      v[0] += __shfl_xor(v[0], 1 << T);
      butterfly16<NV, 1, T + 1>(v, fr);
    }
  }
}


// Epilogue of a data gradient that also accumulates the backward reductions of the BatchNorm
// whose output gradient it writes (sum g, sum g*y per channel; g = the stored value gated by the
// layer's ReLU mask, y = the layer's input).  Three phases:
//   1. every load of the epilogue (y and its mask, addend and its mask, for all MR pixels) is
//      issued before anything is stored: written in the store loop they cannot be hoisted over
//      the stores (the tensors may alias for all the compiler knows), which put 2 * MR memory
//      latencies back to back in every wave and made this fusion lose against the separate
//      reduction sweep on large tensors;
//   2. results are stored pixel by pixel, a lane's channels in consecutive 16-byte stores (splitting
//      a pixel's bytes over two passes costs ~28 % on the bandwidth-bound layers: partial lines);
//      the rounded, gated values replace the accumulators;
//   3. statistics per group of 8 channels from the registers, reduced over the 16 pixel lanes and
//      added to the replica row.
template <int MR, int NR, bool POINTWISE, int DT>
__device__ __forceinline__ void conv_epilogue_bstats(const ConvParams& p, f32x4 (&acc)[NR][MR],
                                                     const int m_base, const int co0, const int mt,
                                                     const int lane) {
  typedef typename Elem<DT>::T E;
  const int fr = lane & 15;
  constexpr int CPL = 4 * NR;
  constexpr int W = CPL >= 8 ? 8 : 4;              // channels per group (one store)
  constexpr int HV = CPL / W;
  constexpr int TPG = W / 4;                       // MFMA tiles per group
  const bool vec_ok = p.ldc % W == 0;
  size_t opixv[MR];
  u32x4 ypre[MR][HV], apre[MR][HV];
  uint32_t sbits[MR][HV], abits[MR][HV];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int m = m_base + mr * 16 + fr;
    size_t opix = 0;
    if (m < p.M) {
      if (POINTWISE) {
        opix = (size_t)m;
      } else {
        const uint32_t t = fd_div((uint32_t)m, p.divOW);
        const int ow = m - (int)t * p.OW;
        const uint32_t n = fd_div(t, p.divOH);
        const int oh = (int)t - (int)n * p.OH;
        opix = (size_t)((int)n * p.OHfull + oh * p.OSH + p.OOH) * p.OWfull + (ow * p.OSW + p.OOW);
      }
    }
    opixv[mr] = opix;
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      const int cg = co0 + h * W;
      ypre[mr][h] = u32x4{0u, 0u, 0u, 0u};
      apre[mr][h] = u32x4{0u, 0u, 0u, 0u};
      sbits[mr][h] = 0xffu;
      abits[mr][h] = 0xffu;
      if (m >= p.M || !vec_ok || cg + W > p.Cout) continue;
      const size_t e = opix * p.ldc + cg;
      const size_t mb = opix * (p.ldc >> 3) + (cg >> 3);
      if constexpr (W == 8) {
        ypre[mr][h] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const E*>(p.stats_y) + e);
      } else {
        const u32x2 q = *reinterpret_cast<const u32x2*>(reinterpret_cast<const E*>(p.stats_y) + e);
        ypre[mr][h].x = q.x; ypre[mr][h].y = q.y;
      }
      // W == 4: the lane's four channels are one nibble of the mask byte
      if (p.stats_mask) sbits[mr][h] = (uint32_t)p.stats_mask[mb] >> (W == 4 ? (cg & 4) : 0);
      if (p.addend) {
        if constexpr (W == 8) {
          apre[mr][h] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const E*>(p.addend) + e);
        } else {
          const u32x2 q = *reinterpret_cast<const u32x2*>(reinterpret_cast<const E*>(p.addend) + e);
          apre[mr][h].x = q.x; apre[mr][h].y = q.y;
        }
        if (p.addend_mask) abits[mr][h] = (uint32_t)p.addend_mask[mb] >> (W == 4 ? (cg & 4) : 0);
      }
    }
  }
  float bias[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c)
    bias[c] = (p.bias != nullptr && co0 + c < p.Cout) ? p.bias[co0 + c] : 0.f;
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    if (m_base + mr * 16 + fr >= p.M) continue;
    const size_t obase = opixv[mr] * p.ldc;
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      const int cg = co0 + h * W;
      if (cg >= p.Cout) continue;
      float v[W];
#pragma unroll
      for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t * 4 + e] = acc[h * TPG + t][mr][e] + bias[h * W + t * 4 + e];
      E* optr = reinterpret_cast<E*>(p.out) + obase + cg;
      if (vec_ok && cg + W <= p.Cout) {
        if (p.addend) {
          const uint32_t aw[4] = {apre[mr][h].x, apre[mr][h].y, apre[mr][h].z, apre[mr][h].w};
          const uint32_t bm = abits[mr][h];       // keep only the elements whose mask bit is set
#pragma unroll
          for (int j = 0; j < W / 2; ++j) {
            const uint32_t a = aw[j] & (((bm >> (2 * j)) & 1u ? 0xffffu : 0u) |
                                        ((bm >> (2 * j + 1)) & 1u ? 0xffff0000u : 0u));
            v[2 * j] += lo16<DT>(a);
            v[2 * j + 1] += hi16<DT>(a);
          }
        }
        uint32_t w[W / 2];
#pragma unroll
        for (int j = 0; j < W / 2; ++j) w[j] = pack2<DT>(v[2 * j], v[2 * j + 1]);
        if constexpr (W == 8) {
          *reinterpret_cast<u32x4*>(optr) = u32x4{w[0], w[1], w[2], w[3]};
        } else {
          u32x2 o;
          o.x = w[0]; o.y = w[1];
          *reinterpret_cast<u32x2*>(optr) = o;
        }
        const uint32_t sb = sbits[mr][h];         // g: the rounded value where the ReLU was open
#pragma unroll
        for (int j = 0; j < W / 2; ++j) {
          acc[h * TPG + (2 * j) / 4][mr][(2 * j) % 4] = (sb >> (2 * j)) & 1u ? lo16<DT>(w[j]) : 0.f;
          acc[h * TPG + (2 * j + 1) / 4][mr][(2 * j + 1) % 4] = (sb >> (2 * j + 1)) & 1u ? hi16<DT>(w[j]) : 0.f;
        }
      } else {                                     // never with statistics (host check); plain store
        const E* aptr = p.addend ? reinterpret_cast<const E*>(p.addend) + obase + cg : nullptr;
#pragma unroll
        for (int c = 0; c < W; ++c) {
          if (cg + c < p.Cout) {
            float x = v[c];
            if (aptr) x += to_f<DT>(aptr[c]);
            optr[c] = from_f<DT>(x);
          }
          acc[h * TPG + c / 4][mr][c % 4] = 0.f;
        }
      }
    }
  }
  if (p.stats == nullptr) return;
#pragma unroll
  for (int h = 0; h < HV; ++h) {
    const int cg = co0 + h * W;
    float st[2 * W];
#pragma unroll
    for (int c = 0; c < 2 * W; ++c) st[c] = 0.f;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      if (m_base + mr * 16 + fr >= p.M) continue;
      const uint32_t yw[4] = {ypre[mr][h].x, ypre[mr][h].y, ypre[mr][h].z, ypre[mr][h].w};
#pragma unroll
      for (int j = 0; j < W / 2; ++j) {
        const float lo = acc[h * TPG + (2 * j) / 4][mr][(2 * j) % 4];
        const float hi = acc[h * TPG + (2 * j + 1) / 4][mr][(2 * j + 1) % 4];
        st[2 * j] += lo;
        st[2 * j + 1] += hi;
        st[W + 2 * j] += lo * lo16<DT>(yw[j]);
        st[W + 2 * j + 1] += hi * hi16<DT>(yw[j]);
      }
    }
    constexpr int NV = 2 * W;                      // 16 or 8
    constexpr int HALVINGS = NV >= 16 ? 4 : 3;
    butterfly16<NV, NV, 0>(st, fr);                // one total left per lane
    int base = 0;
#pragma unroll
    for (int t = 0; t < HALVINGS; ++t) base += ((fr >> t) & 1) * (NV >> (t + 1));
    if (HALVINGS == 4 || (fr >> 3) == 0) {
      float* dst = p.stats + (size_t)(mt % p.stats_replicas) * 2 * p.Cout;
      const int q = base / W, c = base % W;        // quantity (sum g / sum g*y), channel
      if (cg + c < p.Cout) atomicAdd(dst + q * p.Cout + cg + c, st[0]);
    }
  }
}

// m_base: first pixel row of this wave's sub-tile (m0 + wm*WM); co0: see above; mt: M-tile
// index of the block (selects the statistics replica).
template <int MR, int NR, bool POINTWISE, int EPI, int DT = DT_BF16>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[NR][MR],
                                              const int m_base, const int co0, const int mt,
                                              const int lane) {
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
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int m = m_base + mr * 16 + fr;
    if (m >= p.M) continue;
    size_t opix;                                   // linear pixel index in the output tensor
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
            u32x4 a = *reinterpret_cast<const u32x4*>(aptr + h * 8);
            if (mptr) {                            // keep only the elements whose mask bit is set
              const uint32_t b = mptr[h];
              a.x &= ((b & 1u) ? 0xffffu : 0u) | ((b & 2u) ? 0xffff0000u : 0u);
              a.y &= ((b & 4u) ? 0xffffu : 0u) | ((b & 8u) ? 0xffff0000u : 0u);
              a.z &= ((b & 16u) ? 0xffffu : 0u) | ((b & 32u) ? 0xffff0000u : 0u);
              a.w &= ((b & 64u) ? 0xffffu : 0u) | ((b & 128u) ? 0xffff0000u : 0u);
            }
            v0 += lo16<DT>(a.x); v1 += hi16<DT>(a.x);
            v2 += lo16<DT>(a.y); v3 += hi16<DT>(a.y);
            v4 += lo16<DT>(a.z); v5 += hi16<DT>(a.z);
            v6 += lo16<DT>(a.w); v7 += hi16<DT>(a.w);
          }
          if (do_relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
            v4 = fmaxf(v4, 0.f); v5 = fmaxf(v5, 0.f); v6 = fmaxf(v6, 0.f); v7 = fmaxf(v7, 0.f);
          }
          u32x4 o;
          o.x = pack2<DT>(v0, v1); o.y = pack2<DT>(v2, v3); o.z = pack2<DT>(v4, v5); o.w = pack2<DT>(v6, v7);
          *reinterpret_cast<u32x4*>(optr + h * 8) = o;
          if (EPI != EPI_INFER && p.stats) {
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
        if (EPI != EPI_INFER && p.stats) {
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
  if (EPI != EPI_INFER && p.stats) {
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
template <int DT, int MR, int NR, bool POINTWISE>
__device__ __forceinline__ void conv_epilogue_plain(const ConvParams& p, f32x4 (&acc)[NR][MR],
                                                    const int m_base, const int co0,
                                                    const int lane) {
  typedef typename Elem<DT>::T T;
  const int fr = lane & 15;
  T* out = reinterpret_cast<T*>(p.out);
  const T* addend = reinterpret_cast<const T*>(p.addend);
  const bool vec_ok = (p.ldc % 4 == 0);
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
}

}  // namespace conv_epi
