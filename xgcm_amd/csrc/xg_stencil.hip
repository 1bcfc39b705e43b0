// xg_stencil.hip -- two-point stencils (diff / interp / min / max) along one axis (K1, K1g, K2, K2S) and along two axes at once (K8)
// Part of libxgcm_hip.so; compiled twice (real = double / -DXG_F32), see xg_common.hpp.

#include "xg_common.hpp"

#ifndef XG_FUSED_SEG
#define XG_FUSED_SEG 2  // rows per wave-task of the two-axis kernel (K8); 2 and 4 measure alike, 1 loses 4 points
#endif

namespace {

// ------------------------------------------------------------------------------------------
// K2: stencil along a STRIDED axis.  View (outer, n, inner); lane <-> V consecutive elements of
// `inner`; wave-task = (o, segment of `seg` output rows, x-tile of 64*V elements).  Each lane
// marches along the axis holding the previous (metric-weighted) value in registers: exactly
// one 16-B load and one 16-B store per lane per row (+1 halo row per segment).
// MET bit0: m_out present, bit1: m_in present.
// ------------------------------------------------------------------------------------------
template <int OP, int V, int MET, bool NTL, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_stencil_strided(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int seg, u32 nseg, u32 ntile,
    int pad_lo, int bc, real fill, const real* __restrict__ halo, const real* __restrict__ m_in, MIdx mi,
    const real* __restrict__ m_out, MIdx mo) {
  typedef typename VecT<V>::type T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  constexpr int U = 4;

  const u64 w = wave_id();
  const u32 tile = (u32)(w % ntile);
  const u64 r = w / ntile;
  const u32 sg = (u32)(r % nseg);
  const int64_t o = (int64_t)(r / nseg);
  if (o >= g.outer) return;
  const int lane = threadIdx.x & 63;
  const int64_t x = ((int64_t)tile * WAVE + lane) * V;
  if (x >= g.inner) return;

  const int64_t j0 = (int64_t)sg * seg;
  const int64_t j1 = (j0 + seg < g.n_out) ? j0 + seg : g.n_out;
  const int64_t inner = g.inner;
  const real* pin = in + (o * g.n_in) * inner + x;
  real* pout = out + (o * g.n_out) * inner + x;

  int64_t mi_base = 0, mo_base = 0, mi_step = 0, mo_step = 0;
  if (HAS_MI) {
    mi_base = outer_off(g, mi, o) + inner_off(g, mi, x);
    mi_step = (V > 1) ? inner_off(g, mi, x + 1) - inner_off(g, mi, x) : 0;
  }
  if (HAS_MO) {
    mo_base = outer_off(g, mo, o) + inner_off(g, mo, x);
    mo_step = (V > 1) ? inner_off(g, mo, x + 1) - inner_off(g, mo, x) : 0;
  }

  // P(k): value of the padded, metric-weighted input at padded index k (q = k - pad_lo)
  auto loadq = [&](int64_t q) -> T {
    T v = ldg<T, NTL>(pin + q * inner);
    if (HAS_MI) v = v * ldm<T>(m_in, mi_base + q * mi.axis, mi_step);
    return v;
  };
  auto loadP = [&](int64_t k) -> T {
    int64_t q = k - pad_lo;
    if (q < 0 || q >= g.n_in) {
      if (bc == XG_BC_FILL) return splat<T>(fill);
      if (bc == XG_BC_HALO)  // halo values gathered beforehand: layout (outer, pad_lo + pad_hi, inner)
        return *reinterpret_cast<const T*>(halo + ((o * (g.n_out - g.n_in + 1) + (q < 0 ? 0 : pad_lo)) * inner + x));
      q = (q < 0) ? ((bc == XG_BC_PERIODIC) ? g.n_in - 1 : 0) : ((bc == XG_BC_PERIODIC) ? 0 : g.n_in - 1);
    }
    return loadq(q);
  };
  auto emit = [&](int64_t j, T l, T rr) {
    T res = op2<OP>(l, rr);
    if (HAS_MO) res = res / ldm<T>(m_out, mo_base + j * mo.axis, mo_step);
    stg<T, NTS>(pout + j * inner, res);
  };

  T prev = loadP(j0);
  int64_t k = j0 + 1;
  // interior: q = k - pad_lo in [0, n_in) guaranteed for k <= kend
  const int64_t kend = (j1 < g.n_in - 1 + pad_lo) ? j1 : g.n_in - 1 + pad_lo;
  for (; k + (U - 1) <= kend; k += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ldg<T, NTL>(pin + (k + u - pad_lo) * inner);
    if (HAS_MI) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = v[u] * ldm<T>(m_in, mi_base + (k + u - pad_lo) * mi.axis, mi_step);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      emit(k + u - 1, prev, v[u]);
      prev = v[u];
    }
  }
  for (; k <= kend; ++k) {
    T cur = loadq(k - pad_lo);
    emit(k - 1, prev, cur);
    prev = cur;
  }
  for (; k <= j1; ++k) {  // at most one step: the high halo
    T cur = loadP(k);
    emit(k - 1, prev, cur);
    prev = cur;
  }
}

// ------------------------------------------------------------------------------------------
// Linear-order stencil kernels.  Measured on MI355X (profiles/history/r01_streambench_*.txt): a kernel
// whose threads each move ONE 16-byte vector, with thread id == linear memory order, streams at
// the copy ceiling (~79 % of 8 TB/s); giving a thread several rows/tiles costs 10-25 %.  So the
// output is walked as a flat list of V-wide items: item -> (row, position) by one 32-bit
// division (the host splits launches so that item counts stay below 2^31).
//
// K1: stencil along the CONTIGUOUS (last) axis, view (rows, L).
//   V == 2: L_in == L_out even, pads (1,0) or (0,1): one aligned 16-B load + one 8-B neighbour
//           load that hits the same cache lines (L1-served), one 16-B store.
//   V == 1: general path (any pads, odd lengths, N+1 / N-1 outputs): two 8-B loads.
// ------------------------------------------------------------------------------------------
template <int OP, int V, int MET, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_stencil_contig(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int64_t row0, u32 nrows, u32 nblk, FastDiv per,
    ZBand zb, int pad_lo, int bc, real fill, const real* __restrict__ halo, const real* __restrict__ m_in,
    MIdx mi, const real* __restrict__ m_out, MIdx mo, int ntl) {
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  // XCD banding (see K2S): neighbouring workgroups share an L2, so the cache line holding a
  // workgroup's left neighbour is not fetched a second time by another XCD (-3 % HBM reads)
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 gid = lb * BLOCK + threadIdx.x;
  u32 r = fdiv(gid, per);  // per.d = V-wide items per output row
  if (r >= nrows) return;
  const u32 i0 = (gid - r * per.d) * V;
  u32 zz = 0, zy = 0;
  if (MET != 0 && zb.on) {
    if (!zband_map(zb, r, zz, zy)) return;
    r = zz * zb.Y + zy;
  }
  const u32 Li = (u32)g.n_in, Lo = (u32)g.n_out;  // host guarantees row lengths < 2^31
  const real* prow = in + (row0 * (int64_t)Li + (u64)r * Li);
  real* orow = out + (row0 * (int64_t)Lo + (u64)r * Lo);
  // metric row offsets: with z-banding (z, y) are already known, else one FastDiv per outer dim
  // (the host only selects this kernel with metrics when g.idx32 holds)
  int64_t mib = 0, mob = 0;
  if (MET != 0 && zb.on) {  // outer dims are exactly (Z, Y)
    if (HAS_MI) mib = (int64_t)zz * mi.outer[0] + (int64_t)zy * mi.outer[1];
    if (HAS_MO) mob = (int64_t)zz * mo.outer[0] + (int64_t)zy * mo.outer[1];
  } else {
    if (HAS_MI) mib = outer_off32(g, mi, (u32)(row0 + r));
    if (HAS_MO) mob = outer_off32(g, mo, (u32)(row0 + r));
  }

  if (V > 1) {
    u32 nidx;
    bool edge;
    if (pad_lo) { edge = (i0 == 0); nidx = edge ? ((bc == XG_BC_PERIODIC) ? Li - 1 : 0) : i0 - 1; }
    else { edge = (i0 + NV == Li); nidx = edge ? ((bc == XG_BC_PERIODIC) ? 0 : Li - 1) : i0 + NV; }
    dv a = (ntl & 1) ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(prow + i0)) : *reinterpret_cast<const dv*>(prow + i0);
    real n;
    // ntl & 2, no metrics, a wave that lies inside ONE row (27 of 28 waves on 3600-cell rows): the neighbour comes from the
    // next lane's registers (DPP), and the one lane that has no such lane -- the wave's first / last -- takes its value from a
    // SCALAR load (its row and index are wave-uniform): one vector-memory instruction per wave instead of two.  (With a
    // per-lane load for that lane the instruction count stays at two and the kernel loses 0.4 points:
    // profiles/history/r03ab_ab_k1dpp.jsonl.)
    const u32 gfirst = __builtin_amdgcn_readfirstlane(gid - (threadIdx.x & 63));
    const u32 rfirst = fdiv(gfirst, per), rlast = fdiv(gfirst + (WAVE - 1), per);
    if (MET == 0 && (ntl & 2) && rfirst == rlast) {
      const int lane = threadIdx.x & 63;
      const u32 ru = rfirst;
      const real* prow_u = in + (row0 * (int64_t)Li + (u64)ru * Li);
      const u32 ib = (gfirst + (pad_lo ? 0 : WAVE - 1) - ru * per.d) * V;  // first cell of the boundary lane's vector
      u32 nb_idx;
      bool eb;
      if (pad_lo) { eb = (ib == 0); nb_idx = eb ? ((bc == XG_BC_PERIODIC) ? Li - 1 : 0) : ib - 1; }
      else { eb = (ib + NV == Li); nb_idx = eb ? ((bc == XG_BC_PERIODIC) ? 0 : Li - 1) : ib + NV; }
      real nb = prow_u[nb_idx];
      if (eb && bc == XG_BC_FILL) nb = fill;
      if (eb && bc == XG_BC_HALO) nb = halo[(row0 + ru) * (int64_t)(Lo - Li + 1)];
      n = pad_lo ? from_lane_below(a[NV - 1]) : from_lane_above(a[0]);
      if (lane == (pad_lo ? 0 : WAVE - 1)) n = nb;
    } else {
    n = prow[nidx];
    if (HAS_MI) {
      a = a * ldm<dv>(m_in, mib + (int64_t)i0 * mi.axis, mi.axis);
      n = n * m_in[mib + (int64_t)nidx * mi.axis];
    }
    if (edge && bc == XG_BC_FILL) n = fill;
    if (edge && bc == XG_BC_HALO) n = halo[(row0 + r) * (int64_t)(Lo - Li + 1)];  // one halo cell per row here
    }
    dv res;
    if (pad_lo) {
      res[0] = op2<OP>(n, a[0]);
#pragma unroll
      for (int k = 1; k < NV; ++k) res[k] = op2<OP>(a[k - 1], a[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NV - 1; ++k) res[k] = op2<OP>(a[k], a[k + 1]);
      res[NV - 1] = op2<OP>(a[NV - 1], n);
    }
    if (HAS_MO) res = res / ldm<dv>(m_out, mob + (int64_t)i0 * mo.axis, mo.axis);
    // plain operators: the output line is dropped from the L2 as it is written (`sc1 nt`, see stg_drop): +0.6-1.0 points in
    // three alternating-process rounds, the bench's X operators -1 % (profiles/history/r03ba_*, r03bb_*); with metrics no clear gain
    if (NTS && MET == 0) stg_drop<dv>(orow + i0, res);
    else stg<dv, NTS>(orow + i0, res);
  } else {
    int64_t ql = (int64_t)i0 - pad_lo, qr = (int64_t)i0 + 1 - pad_lo;
    bool fl = false, fr = false;
    if (ql < 0) { fl = (bc == XG_BC_FILL); ql = (bc == XG_BC_PERIODIC) ? (int64_t)Li - 1 : 0; }
    if (qr >= (int64_t)Li) { fr = (bc == XG_BC_FILL); qr = (bc == XG_BC_PERIODIC) ? 0 : (int64_t)Li - 1; }
    real l = prow[ql], rr = prow[qr];
    if (HAS_MI) {
      l = l * m_in[mib + ql * mi.axis];
      rr = rr * m_in[mib + qr * mi.axis];
    }
    if (fl) l = fill;
    if (fr) rr = fill;
    if (bc == XG_BC_HALO) {
      const int64_t hb = (row0 + r) * (int64_t)(Lo - Li + 1);
      if ((int64_t)i0 - pad_lo < 0) l = halo[hb];
      if ((int64_t)i0 + 1 - pad_lo >= (int64_t)Li) rr = halo[hb + pad_lo];
    }
    real res = op2<OP>(l, rr);
    if (HAS_MO) res = res / m_out[mob + (int64_t)i0 * mo.axis];
    stg<real, NTS>(orow + i0, res);
  }
}

// ------------------------------------------------------------------------------------------
// K1r: contiguous axis WITH metrics (derivative / metric_weighted along X), wave-task = (R consecutive
// rows, x-tile of 64 lane vectors).  Why a second form of K1: a flat one-vector-per-thread kernel keeps
// (waves) x (one 1-KB load) in flight, so whatever a wave computes between its load and its store adds to
// the memory latency it has to hide.  K1 with a divisor spends ~100 VALU instructions per lane vector,
// 2/3 of them per-lane row decoding (rows of 1800 vectors do not align with waves, so the row index, the
// z-band map and every row offset are vector work).  Here the row group and tile of a wave are uniform:
// the decode and all row bases live on the scalar unit, a lane computes one 32-bit offset, and the loads
// of all R rows (field, neighbour, metrics) are issued before the first operation -- R KB in flight per
// wave instead of one.  Rows are 16-B aligned (L % NV == 0); `mal`: every metric row is too, and
// contiguous along X (host-proven) => one 16-B metric load per lane vector, no per-lane alignment test.
// Outer dims are (Z, Y) or (Y); with broadcast metrics the row groups are visited band-major (z-banding).
// ------------------------------------------------------------------------------------------
// ZS ("z-share", needs z-banding): the R rows of a wave are the SAME row y of R consecutive levels, so one
// metric vector (and one neighbour metric) serves all R rows -- metric loads, although L2 hits, compete with
// the field loads for the CU's outstanding-request capacity.
template <int OP, int MET, int R, bool ZS>
__global__ __launch_bounds__(BLOCK) void k_stencil_contig_rw(
    const real* __restrict__ in, real* __restrict__ out, u32 L, u32 Z, u32 Y, u32 nblk, FastDiv ntile, FastDiv fYG,
    ZBand zb, int pad_lo, int bc, real fill, const real* __restrict__ halo, const real* __restrict__ m_in,
    int64_t mi_z, int64_t mi_y, int64_t mi_x, const real* __restrict__ m_out, int64_t mo_z, int64_t mo_y,
    int64_t mo_x, int mal, int ntl) {
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  constexpr int RM = ZS ? 1 : R;  // metric vectors a wave loads
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 grp = fdiv(w, ntile);
  const u32 tile = w - grp * ntile.d;
  u32 z0, y0;  // first row of the wave: rows (z0, y0 + u) or, z-shared, (z0 + u, y0)
  if (ZS) {    // band-major over (band of B rows, level group, row in band)
    u32 zg, y;
    if (!zband_map(zb, grp, zg, y)) return;
    z0 = zg * R;
    y0 = y;
  } else if (zb.on) {
    u32 z, yg;
    if (!zband_map(zb, grp, z, yg)) return;
    z0 = z;
    y0 = yg * R;
  } else {
    const u32 z = fdiv(grp, fYG);
    if (z >= Z) return;
    z0 = z;
    y0 = (grp - z * fYG.d) * R;
  }
  const u32 nvalid = ZS ? ((Z - z0 < (u32)R) ? Z - z0 : (u32)R) : ((Y - y0 < (u32)R) ? Y - y0 : (u32)R);
  const u32 i0 = (tile * WAVE + (threadIdx.x & 63)) * NV;
  if (i0 >= L) return;
  u32 nidx;
  bool edge;
  if (pad_lo) { edge = (i0 == 0); nidx = edge ? ((bc == XG_BC_PERIODIC) ? L - 1 : 0) : i0 - 1; }
  else { edge = (i0 + NV == L); nidx = edge ? ((bc == XG_BC_PERIODIC) ? 0 : L - 1) : i0 + NV; }

  const bool nb_dpp = (ntl & 2) != 0;
  const int lane = threadIdx.x & 63;
  const bool own_nb = edge || (pad_lo ? lane == 0 : lane == WAVE - 1);  // no source lane: this lane loads the value itself
  dv a[R], wi[RM], wo[RM];
  real n[R], wn[RM];
  u64 rows[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const u32 uu = ((u32)u < nvalid) ? (u32)u : nvalid - 1;  // a short last group repeats its last row (not stored)
    rows[u] = ZS ? (u64)(z0 + uu) * Y + y0 : (u64)z0 * Y + (y0 + uu);
    const real* prow = in + rows[u] * L;
    a[u] = (ntl & 1) ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(prow + i0)) : *reinterpret_cast<const dv*>(prow + i0);
    if (!nb_dpp) n[u] = prow[nidx];
  }
  if (nb_dpp) {  // the value beside a lane's vector from the neighbouring lane's registers (see from_lane_below)
#pragma unroll
    for (int u = 0; u < R; ++u) n[u] = pad_lo ? from_lane_below(a[u][NV - 1]) : from_lane_above(a[u][0]);
    if (own_nb) {  // (scalar loads for this one lane, as in the flat kernel: no change here, 0.749 / 0.722 either way --
                   // the block's R loads issue together; profiles/history/r03ar_ab_k1r_scalar.jsonl)
#pragma unroll
      for (int u = 0; u < R; ++u) n[u] = in[rows[u] * L + nidx];
    }
  }
  if (edge && bc == XG_BC_HALO) {
#pragma unroll
    for (int u = 0; u < R; ++u) n[u] = halo[rows[u]];  // one halo cell per row (never weighted: no m_in with halos)
  }
#pragma unroll
  for (int u = 0; u < RM; ++u) {
    const u32 uu = ((u32)u < nvalid) ? (u32)u : nvalid - 1;
    const int64_t zz = ZS ? 0 : (int64_t)z0, yy = ZS ? (int64_t)y0 : (int64_t)(y0 + uu);
    if (HAS_MI) {
      const real* mrow = m_in + (zz * mi_z + yy * mi_y);
      if (mal) wi[u] = *reinterpret_cast<const dv*>(mrow + i0);
      else {
#pragma unroll
        for (int k = 0; k < NV; ++k) wi[u][k] = mrow[(int64_t)(i0 + k) * mi_x];
      }
      if (nb_dpp && mal) {
        wn[u] = pad_lo ? from_lane_below(wi[u][NV - 1]) : from_lane_above(wi[u][0]);
        if (own_nb) wn[u] = mrow[(int64_t)nidx * mi_x];
      } else {
        wn[u] = mrow[(int64_t)nidx * mi_x];
      }
    }
    if (HAS_MO) {
      const real* mrow = m_out + (zz * mo_z + yy * mo_y);
      if (mal) wo[u] = *reinterpret_cast<const dv*>(mrow + i0);
      else {
#pragma unroll
        for (int k = 0; k < NV; ++k) wo[u][k] = mrow[(int64_t)(i0 + k) * mo_x];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if ((u32)u >= nvalid) break;
    constexpr int um_mask = ZS ? 0 : ~0;
    const int um = u & um_mask;
    dv av = a[u];
    real nv = n[u];
    if (HAS_MI) {
      av = av * wi[um];
      if (!(edge && bc == XG_BC_HALO)) nv = nv * wn[um];
    }
    if (edge && bc == XG_BC_FILL) nv = fill;
    dv res;
    if (pad_lo) {
      res[0] = op2<OP>(nv, av[0]);
#pragma unroll
      for (int k = 1; k < NV; ++k) res[k] = op2<OP>(av[k - 1], av[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NV - 1; ++k) res[k] = op2<OP>(av[k], av[k + 1]);
      res[NV - 1] = op2<OP>(av[NV - 1], nv);
    }
    // (measured with the divisor load removed: 78.6 %; with a product in place of the IEEE division: no change --
    // profiles/history/r02b_ab_bounds.jsonl: the L2-resident metric LOAD is the cost, not the division)
    if (HAS_MO) res = res / wo[um];
    stg_s<dv, true>(out + rows[u] * L + i0, res);  // (`sc1 nt`: derivative X +0.3 / +0.8 points on two boxes, two metrics +-0)
  }
}

// ------------------------------------------------------------------------------------------
// K1g: contiguous axis, GENERAL lengths (odd rows, N+1 / N-1 outputs: outer/inner positions).
// Rows of the output are then not 16-B aligned, but the output ARRAY is: the array is walked as
// a flat list of NV-element groups (which may straddle two rows), each element is computed like
// the V == 1 path of K1 (two narrow loads served by L1) and the group leaves as one aligned
// 16-B store.  1/NV of the threads, index math and store instructions of the one-element form.
// ------------------------------------------------------------------------------------------
template <int OP, int MET, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_stencil_contig_gen(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int64_t row0, u32 nelem, u32 nblk,
    FastDiv fLo, int pad_lo, int bc, real fill, const real* __restrict__ halo, const real* __restrict__ m_in,
    MIdx mi, const real* __restrict__ m_out, MIdx mo) {
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 gid = lb * BLOCK + threadIdx.x;
  if (gid >= (nelem + NV - 1) / NV) return;
  const u32 e0 = NV * gid;
  const u32 Li = (u32)g.n_in, Lo = (u32)g.n_out;
  auto one = [&](u32 r, u32 i) -> real {
    const real* prow = in + (row0 * (int64_t)Li + (u64)r * Li);
    int64_t ql = (int64_t)i - pad_lo, qr = (int64_t)i + 1 - pad_lo;
    bool fl = false, fr = false;
    if (ql < 0) { fl = (bc == XG_BC_FILL); ql = (bc == XG_BC_PERIODIC) ? (int64_t)Li - 1 : 0; }
    if (qr >= (int64_t)Li) { fr = (bc == XG_BC_FILL); qr = (bc == XG_BC_PERIODIC) ? 0 : (int64_t)Li - 1; }
    real l = prow[ql], rr = prow[qr];
    if (HAS_MI) {
      const int64_t mib = outer_off32(g, mi, (u32)(row0 + r));
      l = l * m_in[mib + ql * mi.axis];
      rr = rr * m_in[mib + qr * mi.axis];
    }
    if (fl) l = fill;
    if (fr) rr = fill;
    if (bc == XG_BC_HALO) {
      const int64_t hb = (row0 + r) * (int64_t)(Lo - Li + 1);
      if ((int64_t)i - pad_lo < 0) l = halo[hb];
      if ((int64_t)i + 1 - pad_lo >= (int64_t)Li) rr = halo[hb + pad_lo];
    }
    real res = op2<OP>(l, rr);
    if (HAS_MO) res = res / m_out[outer_off32(g, mo, (u32)(row0 + r)) + (int64_t)i * mo.axis];
    return res;
  };
  u32 r = fdiv(e0, fLo), i = e0 - r * Lo;
  real* po = out + (row0 * (int64_t)Lo + (u64)e0);  // row0 * Lo is a multiple of NV (host) => 16-B aligned
  dv res;
  const int64_t q0 = (int64_t)i - pad_lo;
  if (i + NV <= Lo && q0 >= 0 && q0 + NV < (int64_t)Li && e0 + NV <= nelem) {
    // interior group inside one row: the NV outputs share NV + 1 consecutive inputs
    const real* prow = in + (row0 * (int64_t)Li + (u64)r * Li) + q0;
    real v[NV + 1];
#pragma unroll
    for (int k = 0; k <= NV; ++k) v[k] = prow[k];
    if (HAS_MI) {
      const int64_t mib = outer_off32(g, mi, (u32)(row0 + r)) + q0 * mi.axis;
#pragma unroll
      for (int k = 0; k <= NV; ++k) v[k] = v[k] * m_in[mib + k * mi.axis];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) res[k] = op2<OP>(v[k], v[k + 1]);
    if (HAS_MO) {
      const int64_t mob = outer_off32(g, mo, (u32)(row0 + r)) + (int64_t)i * mo.axis;
#pragma unroll
      for (int k = 0; k < NV; ++k) res[k] = res[k] / m_out[mob + k * mo.axis];
    }
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (e0 + k < nelem) res[k] = one(r, i);
      if (++i == Lo) { i = 0; ++r; }
    }
  }
  if (e0 + NV <= nelem) {
    stg_s<dv, NTS>(po, res);  // (`sc1 nt`, rule 16: +0.8 points in 3 of 3 rounds; the strided general kernel K2g loses 10 and keeps `nt`)
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (e0 + k < nelem) po[k] = res[k];
  }
}

// ------------------------------------------------------------------------------------------
// K2S: stencil along a STRIDED axis, the small-row case (few x-tiles per row, e.g. Y of a
// (Z,Y,X) field).  Measured on MI355X with random data (profiles/history/r01_streambench_d_*.txt):
//   * the set of rows in flight must stay compact: each wave register-marches only SEG (= 4)
//     rows -- SEG+1 independent 16-B loads, then SEG stores -- instead of a long segment;
//   * the halo row a segment re-reads must come from the SAME XCD's L2: workgroup b runs on XCD
//     b % 8 (observed dispatch rule, used for speed only), so the linear wave sequence is cut
//     into 8 contiguous bands, one per XCD ("banding").  Each XCD then streams one compact
//     address range and its re-reads never cross the fabric.  6.3 TB/s vs 5.1 TB/s without.
// One wave = one x-tile of one SEG-row segment; the (outer, segment, tile) split and all row
// bases are wave-uniform (scalar unit, FastDiv).
// ------------------------------------------------------------------------------------------
// ZK > 1 ("z-share", z-banded launches only): the wave carries the same rows of ZK consecutive outer levels;
// the metric rows, broadcast along that dim, are loaded once for all of them (see K1r).
template <int OP, int V, int MET, bool NTS, int SEG, int ZK = 1>
__global__ __launch_bounds__(BLOCK) void k_stencil_strided_seg(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int64_t o0, u32 nouter, u32 nblk,
    FastDiv ntile, FastDiv nseg, ZBand zb, Chunk ck, int pad_lo, int bc, real fill,
    const real* __restrict__ halo, const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo,
    int mal) {
  typedef typename VecT<V>::type T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  // `mal` (host-proven): every metric is contiguous along the lanes and each of its rows 16-B aligned, so a
  // lane vector's metric is ONE 16-B load with no per-lane alignment test
  auto ldmv = [&](const real* m, int64_t off, int64_t step) -> T {
    if (V > 1 && (mal & 1)) return *reinterpret_cast<const T*>(m + off);
    return ldm<T>(m, off, step);
  };
  // banding: XCD (b % 8) owns logical blocks [xcd * pb, (xcd + 1) * pb)
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  u32 oo, sg, tile, zl = 0, ostep = 1;
  if (ck.on) {  // (outer, chunk, segment, tile in chunk)
    const u32 cg = fdiv(w, ck.per_group);
    const u32 rem = w - cg * ck.per_group.d;
    sg = fdiv(rem, ck.ch);
    oo = fdiv(cg, ck.fnchunk);
    tile = (cg - oo * ck.fnchunk.d) * ck.ch.d + (rem - sg * ck.ch.d);
    if (oo >= nouter || tile >= ntile.d) return;
    zl = oo;
  } else {
    const u32 r = fdiv(w, ntile);
    tile = w - r * ntile.d;
    if (MET != 0 && zb.on) {  // band-major order over (segment band, outer [group], segment)
      if (!zband_map(zb, r, oo, sg)) return;
      zl = oo * ZK;
      zband_face(zb, zl, sg, oo, ostep);  // (faces under the levels: `nouter` counts LEVELS, the levels of a task lie `ostep` apart)
    } else {
      oo = fdiv(r, nseg);
      if (oo >= nouter) return;
      sg = r - oo * nseg.d;
      zl = oo;
    }
  }
  const int64_t o = o0 + oo;
  const int nk = (ZK > 1 && (int64_t)nouter - (int64_t)zl < ZK) ? (int)(nouter - zl) : ZK;  // levels this wave really has
  const int64_t inner = g.inner;
  const int64_t x = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (x >= inner) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (g.n_out - j0 < SEG) ? g.n_out - j0 : SEG;  // rows this segment really has

  int64_t mib = 0, mob = 0, mis = 0, mos = 0;  // (host guarantees g.idx32 when metrics are present)
  const bool ui = HAS_MI && (mal & 2), uo = HAS_MO && (mal & 4);  // row-uniform metrics: scalar loads
  int64_t mibu = 0, mobu = 0;                                     // their wave-uniform offsets
  if (HAS_MI) {
    if (ui) mibu = outer_off32(g, mi, (u32)o);
    else {
      inner_off_step32(g, mi, (u32)x, V > 1, mib, mis);
      mib += outer_off32(g, mi, (u32)o);
    }
  }
  if (HAS_MO) {
    if (uo) mobu = outer_off32(g, mo, (u32)o) + j0 * mo.axis;
    else {
      inner_off_step32(g, mo, (u32)x, V > 1, mob, mos);
      mob += outer_off32(g, mo, (u32)o) + j0 * mo.axis;
    }
  }

  // padded index k = j0 + u  ->  input row q (wave-uniform), fill flag
  T v[ZK][SEG + 1];
  T wm[SEG + 1];  // input-metric rows, shared by the ZK levels
  bool ff[SEG + 1], hh[SEG + 1];
  int64_t qq[SEG + 1];
#pragma unroll
  for (int u = 0; u <= SEG; ++u) {
    int64_t k = j0 + ((u <= nrow) ? u : nrow);  // clamp inside the padded range for short tails
    int64_t q = k - pad_lo;
    ff[u] = false;
    hh[u] = false;
    if (q < 0 || q >= g.n_in) {
      ff[u] = (bc == XG_BC_FILL);
      if (bc == XG_BC_HALO) {  // pre-gathered halo rows, layout (outer, pad_lo + pad_hi, inner)
        hh[u] = true;
        q = (q < 0) ? 0 : pad_lo;
      } else {
        q = (q < 0) ? ((bc == XG_BC_PERIODIC) ? g.n_in - 1 : 0) : ((bc == XG_BC_PERIODIC) ? 0 : g.n_in - 1);
      }
    }
    qq[u] = q;
  }
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    const int64_t ok = o + (int64_t)((kz < nk) ? kz : nk - 1) * ostep;  // a short last group repeats its last level (not stored)
    const real* pin = in + (ok * g.n_in) * inner + x;
    const real* phalo = halo + (ok * (g.n_out - g.n_in + 1)) * inner + x;
#pragma unroll
    for (int u = 0; u <= SEG; ++u) v[kz][u] = *reinterpret_cast<const T*>((hh[u] ? phalo : pin) + qq[u] * inner);
  }
  if (HAS_MI) {
#pragma unroll
    for (int u = 0; u <= SEG; ++u) wm[u] = ui ? splat<T>(m_in[mibu + qq[u] * mi.axis]) : ldmv(m_in, mib + qq[u] * mi.axis, mis);
  }
  T dm[SEG];  // divisors: loaded with the field rows, before the first operation
  if (HAS_MO) {
#pragma unroll
    for (int u = 0; u < SEG; ++u) dm[u] = uo ? splat<T>(m_out[mobu + ((u < nrow) ? u : 0) * mo.axis]) : ldmv(m_out, mob + ((u < nrow) ? u : 0) * mo.axis, mos);
  }
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    if (kz >= nk) break;
    real* pout = out + ((o + (int64_t)kz * ostep) * g.n_out + j0) * inner + x;
    T p[SEG + 1];
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      T t = v[kz][u];
      if (HAS_MI && !hh[u]) t = t * wm[u];  // (pre-gathered halo rows are those of the PRODUCT: not weighted again)
      p[u] = ff[u] ? splat<T>(fill) : t;
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      if (u < nrow) {
        T res = op2<OP>(p[u], p[u + 1]);
        if (HAS_MO) res = res / dm[u];
        stg_s<T, NTS>(pout + u * inner, res);  // (`sc1 nt`: diff Z +0.7, derivative Z +1.2, metric_weighted Z +0.3, Y +-0)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K2g: strided axis, rows NOT vector-aligned (odd inner extent: fields on `outer` positions, N + 1 points
// along X).  The view is (outer, n, inner); the output ARRAY is 16-B aligned even if its rows are not: a row
// starts `lead` cells before a 16-B boundary, and from there on it is cut into NV-element groups that leave
// as aligned 16-B stores.  The last group of a row runs over into the next row's lead cells, so every cell
// has exactly one owner.  Work order and XCD banding are those of K2S with one row per wave-task (column
// chunks for whole-plane rows), so the second source row of a task is an L2 hit; the row's coordinates and
// its halo / fill decisions are wave-uniform.  Inputs are narrow consecutive loads (their alignment differs
// from the output's); metrics are addressed per element through their strides.
// ------------------------------------------------------------------------------------------
template <int OP, int MET, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_stencil_strided_gen(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int64_t o0, u32 nouter, u32 nblk, FastDiv ntile,
    FastDiv nseg, Chunk ck, int pad_lo, int bc, real fill, const real* __restrict__ halo,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo) {
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  u32 oo, jj, tile;
  if (ck.on) {  // (outer, chunk, row, tile in chunk)
    const u32 cg = fdiv(w, ck.per_group);
    const u32 rem = w - cg * ck.per_group.d;
    jj = fdiv(rem, ck.ch);
    oo = fdiv(cg, ck.fnchunk);
    tile = (cg - oo * ck.fnchunk.d) * ck.ch.d + (rem - jj * ck.ch.d);
    if (oo >= nouter || tile >= ntile.d) return;
  } else {
    const u32 r = fdiv(w, ntile);
    tile = w - r * ntile.d;
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    jj = r - oo * nseg.d;
  }
  const int64_t inner = g.inner, n_in = g.n_in, n_out = g.n_out;
  const int nhalo = (int)(n_out - n_in + 1);  // pad_lo + pad_hi
  // one output row (o, j): its two source rows (base pointers, fill flags, input-metric row offsets) and
  // the output-metric row offset -- all wave-uniform
  struct Row { const real* src[2]; bool fl[2]; bool wm[2]; int64_t mib[2]; int64_t mob; };
  auto resolve = [&](int64_t o, int64_t j) -> Row {
    Row rw;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      int64_t q = j + side - pad_lo;
      rw.fl[side] = false;
      rw.wm[side] = HAS_MI;
      const real* base = in + o * n_in * inner;
      if (q < 0 || q >= n_in) {
        rw.fl[side] = (bc == XG_BC_FILL);
        if (bc == XG_BC_HALO) {  // pre-gathered halo rows, layout (outer, pad_lo + pad_hi, inner); never weighted
          base = halo + o * nhalo * inner;
          q = (q < 0) ? 0 : pad_lo;
          rw.wm[side] = false;
        } else {
          q = (q < 0) ? ((bc == XG_BC_PERIODIC) ? n_in - 1 : 0) : ((bc == XG_BC_PERIODIC) ? 0 : n_in - 1);
        }
      }
      rw.src[side] = base + q * inner;
      rw.mib[side] = HAS_MI ? outer_off32(g, mi, (u32)o) + q * mi.axis : 0;
    }
    rw.mob = HAS_MO ? outer_off32(g, mo, (u32)o) + j * mo.axis : 0;
    return rw;
  };
  // one output cell at inner index xi of a resolved row (one inner dim -- the usual case -- needs no division)
  const bool single_inner = g.n_inner == 1;
  auto cell = [&](const Row& rw, int64_t xi) -> real {
    real l = rw.src[0][xi], r = rw.src[1][xi];
    if (HAS_MI) {
      const int64_t io = single_inner ? xi * mi.inner[0] : inner_off32(g, mi, (u32)xi);
      if (rw.wm[0]) l = l * m_in[rw.mib[0] + io];
      if (rw.wm[1]) r = r * m_in[rw.mib[1] + io];
    }
    real v = op2<OP>(rw.fl[0] ? fill : l, rw.fl[1] ? fill : r);
    if (HAS_MO) v = v / m_out[rw.mob + (single_inner ? xi * mo.inner[0] : inner_off32(g, mo, (u32)xi))];
    return v;
  };
  const int64_t o = o0 + oo, j = jj;
  const int64_t rowstart = (o * n_out + j) * inner;             // flat index of the row's first cell
  const int64_t lead = (NV - rowstart % NV) % NV;               // those cells belong to the previous row's last group
  const int64_t x = lead + ((int64_t)tile * WAVE + (threadIdx.x & 63)) * NV;
  if (x >= inner) return;
  const Row rw = resolve(o, j);
  dv res;
  if (x + NV <= inner) {  // the group lies inside the row: consecutive narrow loads, no per-element logic
#pragma unroll
    for (int e = 0; e < NV; ++e) res[e] = cell(rw, x + e);
    stg<dv, NTS>(out + rowstart + x, res);
    return;
  }
  // the row's last group runs over into the next row (or past the end of the array)
  int64_t o2 = o, j2 = j + 1;
  if (j2 == n_out) { j2 = 0; ++o2; }
  const bool more = o2 < g.outer;
  const Row rw2 = more ? resolve(o2, j2) : rw;
  real vals[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int64_t xe = x + e;
    real v = real(0);
    if (xe < inner) v = cell(rw, xe);
    else if (more) v = cell(rw2, xe - inner);
    vals[e] = v;
  }
  if (more) {
#pragma unroll
    for (int e = 0; e < NV; ++e) res[e] = vals[e];
    stg<dv, NTS>(out + rowstart + x, res);
  } else {
    for (int e = 0; x + e < inner; ++e) out[rowstart + x + e] = vals[e];
  }
}

// ------------------------------------------------------------------------------------------
// K8: the same two-point operator along BOTH of the last two axes in one pass, e.g.
// Grid.interp(da, ["X", "Y"]) (tracer -> vorticity point).  The reference applies the axes one
// after the other (xgcm/grid.py:798-800 carries a TODO about fusing them): pad + op along the
// first, then pad + op along the second = 32 B/cell.  Here one wave loads SEG+1 rows of pairs
// plus the 8-B X neighbour (as K7), applies the first axis in registers and the second across
// rows: 16 B/cell, and bit-identical to the sequential form because the order of the
// floating-point operations is kept (`order` 0: X then Y, 1: Y then X).  The halo of the SECOND
// axis acts on the intermediate array, as in the reference: a fill halo is the constant itself,
// periodic/extend halos are the first-axis result of the wrapped/clamped row or column.
// Length-preserving position pairs only (pads (1,0)/(0,1)), nx even; other cases run sequentially.
// ------------------------------------------------------------------------------------------
// MET: `metric_weighted` on both axes with the same metric set (Grid.interp(da, ["X", "Y"], metric_weighted=("X", "Y")),
// area-weighted interpolation to the corner points): the reference multiplies by the metric at the current position,
// applies the axis, divides by the metric at the new position -- twice (xgcm/grid.py:804-828).  Three planes (ny, nx)
// shared by all outer indices: m1 at the input positions, m2 at the positions after the first axis (divisor of the
// first step AND factor of the second: `(t / m2) * m2` is kept as written, it is not the identity in floating point),
// m3 at the output positions.  A fill halo replaces the PRODUCT, as in the reference (the array is padded after the
// multiplication).
// ZK (MET only, band-major order): outer indices per wave-task sharing the metric rows in registers -- with one level
// per task the kernel moves 56 B per output cell from the L2 to the CUs (three metric planes + the field) and stops at
// 46 % of 8 TB/s, the L2 -> CU path saturated like K4c's; two passes of the metric-carrying 1-D kernels take 3.6 ms.
template <int OP, bool NTS, int SEG, bool MET, int ZK = 1>
__global__ __launch_bounds__(BLOCK) void k_stencil2d(
    const real* __restrict__ in, real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny,
    int64_t nx, FastDiv ntile, FastDiv nseg, int order, int plx, int bcx, real fillx, int ply, int bcy,
    real filly, const real* __restrict__ m1, const real* __restrict__ m2, const real* __restrict__ m3, ZBand zb) {
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  const u32 tile = w - r * ntile.d;
  u32 oo, sg;
  if (MET && zb.on) {  // band-major: every outer index of a band of rows before the next band (the three planes stay in L2)
    if (!zband_map(zb, r, oo, sg)) return;
    oo *= ZK;
    if (oo >= nouter) return;
  } else {
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    sg = r - oo * nseg.d;
  }
  const int nk = (ZK > 1 && (int64_t)nouter - (int64_t)oo < ZK) ? (int)(nouter - oo) : ZK;  // levels this wave really has
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * NV;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;

  int64_t nidx;
  bool edge;
  if (plx) { edge = (i0 == 0); nidx = edge ? ((bcx == XG_BC_PERIODIC) ? nx - 1 : 0) : i0 - 1; }
  else { edge = (i0 + NV == nx); nidx = edge ? ((bcx == XG_BC_PERIODIC) ? 0 : nx - 1) : i0 + NV; }
  const bool fill_edge = edge && (bcx == XG_BC_FILL);
  // X stencil on a lane vector `a` with the value `n` next to it (left of a[0] if plx, right of a[NV-1] otherwise)
  auto opx = [&](dv a, real n) -> dv {
    dv t;
    if (plx) {
      t[0] = op2<OP>(n, a[0]);
#pragma unroll
      for (int k = 1; k < NV; ++k) t[k] = op2<OP>(a[k - 1], a[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NV - 1; ++k) t[k] = op2<OP>(a[k], a[k + 1]);
      t[NV - 1] = op2<OP>(a[NV - 1], n);
    }
    return t;
  };

  bool rowfill[SEG + 1];
  int64_t qq[SEG + 1];
#pragma unroll
  for (int u = 0; u <= SEG; ++u) {
    int64_t k = j0 + ((u <= nrow) ? u : nrow);
    int64_t q = k - ply;
    bool f = false;
    if (q < 0) { f = (bcy == XG_BC_FILL); q = (bcy == XG_BC_PERIODIC) ? ny - 1 : 0; }
    else if (q >= ny) { f = (bcy == XG_BC_FILL); q = (bcy == XG_BC_PERIODIC) ? 0 : ny - 1; }
    rowfill[u] = f;
    qq[u] = q;
  }
  // The value next to a lane's vector (left of its first element if plx, right of its last otherwise) is the
  // neighbouring lane's last / first element: taken from that lane's registers (DPP) instead of an 8-byte load per row
  // and array -- 21 of this kernel's 45 loads with metrics, whose addresses walk over the cache lines of the vector loads
  // a second time.  The lane at the wave's end and the lane at the row's end load the value themselves.
  const int lane = threadIdx.x & 63;
  const bool nb_dpp = (order & 2) != 0;
  order &= 1;
  const bool own_nb = !nb_dpp || edge || (plx ? lane == 0 : lane == WAVE - 1);
  auto beside = [&](dv v) -> real { return plx ? from_lane_below(v[NV - 1]) : from_lane_above(v[0]); };
  // the field rows of every level of this task, then the metric rows (shared by the levels)
  dv prz[ZK][SEG + 1];
  real nbz[ZK][SEG + 1];
  dv a1[MET ? SEG + 1 : 1], mid[MET ? SEG + 1 : 1], d3[MET ? SEG : 1];
  real a1n[MET ? SEG + 1 : 1], midn[MET ? SEG + 1 : 1];
  int64_t mrow[SEG + 1];
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    const real* pin = in + (o0 + oo + ((kz < nk) ? kz : nk - 1)) * ny * nx;  // a short last group repeats its last level (not stored)
#pragma unroll
    for (int u = 0; u <= SEG; ++u) prz[kz][u] = *reinterpret_cast<const dv*>(pin + qq[u] * nx + i0);
  }
  if (MET) {
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      a1[u] = *reinterpret_cast<const dv*>(m1 + qq[u] * nx + i0);
      // between the axes: (Y as the input, X as the output) when X goes first, (Y as the output, X as the input) otherwise
      mrow[u] = (order == 0) ? qq[u] : j0 + ((u < nrow) ? u : 0);
      mid[u] = *reinterpret_cast<const dv*>(m2 + mrow[u] * nx + i0);
    }
  }
  if (nb_dpp) {  // every neighbour value from the lane beside ...
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz)
#pragma unroll
      for (int u = 0; u <= SEG; ++u) nbz[kz][u] = beside(prz[kz][u]);
    if (MET) {
#pragma unroll
      for (int u = 0; u <= SEG; ++u) {
        a1n[u] = beside(a1[u]);
        midn[u] = beside(mid[u]);
      }
    }
  }
  if (own_nb) {  // ... except in the lanes that have none (ONE divergent block: the wave's end lane and the row's end lane)
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz) {
      const real* pin = in + (o0 + oo + ((kz < nk) ? kz : nk - 1)) * ny * nx;
#pragma unroll
      for (int u = 0; u <= SEG; ++u) nbz[kz][u] = pin[qq[u] * nx + nidx];
    }
    if (MET) {
#pragma unroll
      for (int u = 0; u <= SEG; ++u) {
        a1n[u] = m1[qq[u] * nx + nidx];
        midn[u] = m2[mrow[u] * nx + nidx];
      }
    }
  }
  if (MET) {
#pragma unroll
    for (int u = 0; u < SEG; ++u) d3[u] = *reinterpret_cast<const dv*>(m3 + (j0 + ((u < nrow) ? u : 0)) * nx + i0);
  }
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
  if (kz >= nk) break;
  real* po = out + ((o0 + oo + kz) * ny + j0) * nx + i0;
  dv pr[SEG + 1];
  real nb[SEG + 1];
#pragma unroll
  for (int u = 0; u <= SEG; ++u) {  // the products at the input positions (a halo row / column repeats the product it copies)
    pr[u] = MET ? prz[kz][u] * a1[u] : prz[kz][u];
    nb[u] = MET ? nbz[kz][u] * a1n[u] : nbz[kz][u];
  }
  if (order == 0) {  // X first, then Y on the intermediate
    dv tx[SEG + 1];
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      dv t = opx(pr[u], fill_edge ? fillx : nb[u]);
      if (MET) t = (t / mid[u]) * mid[u];
      tx[u] = rowfill[u] ? splat<dv>(filly) : t;
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u)
      if (u < nrow) {
        dv res = op2<OP>(tx[u], tx[u + 1]);
        if (MET) res = res / d3[u];
        if (MET) stg<dv, NTS>(po + u * nx, res);
        else stg_s<dv, NTS>(po + u * nx, res);  // (`sc1 nt`: two-axis interp +1.3 points; with metrics -1.2: those keep `nt`)
      }
  } else {  // Y first, then X on the intermediate
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      if (rowfill[u]) { pr[u] = splat<dv>(filly); nb[u] = filly; }
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      if (u < nrow) {
        dv ty = op2<OP>(pr[u], pr[u + 1]);
        real tn = op2<OP>(nb[u], nb[u + 1]);
        if (MET) {
          ty = (ty / mid[u]) * mid[u];
          tn = (tn / midn[u]) * midn[u];
        }
        dv res = opx(ty, fill_edge ? fillx : tn);
        if (MET) res = res / d3[u];
        if (MET) stg<dv, NTS>(po + u * nx, res);
        else stg_s<dv, NTS>(po + u * nx, res);
      }
    }
  }
  }
}

// K8y: K8 with `metric_weighted` on both axes, X first, band-major -- the case of `Grid.interp(da, ["X", "Y"],
// metric_weighted=("X", "Y"))` -- with Y-STACKED workgroups: the WPB waves of a workgroup are WPB consecutive segments of
// ONE x-tile and level group, and the row a segment shares with its neighbour (its halo row) is not loaded and pushed
// through the X stencil and the `(t / m2) * m2` round trip a second time: the neighbour hands its finished intermediate row
// over through LDS.  K8 computes SEG + 1 intermediate rows for SEG output rows (3 for 2): a third of its X-stencil work,
// divisions, field and metric loads is that shared row.  Only the wave at the workgroup's edge (or at the array's edge,
// where the halo follows the boundary rule) computes its halo row itself.  Same operations on the same operands: same bits.
template <int OP, bool NTS, int SEG, int ZK>
__global__ __launch_bounds__(BLOCK) void k_stencil2d_ys(
    const real* __restrict__ in, real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny, int64_t nx,
    FastDiv ntile, int dpp, int plx, int bcx, real fillx, int ply, int bcy, real filly,
    const real* __restrict__ m1, const real* __restrict__ m2, const real* __restrict__ m3, ZBand zb) {
  __shared__ dv s_tx[WPB][ZK][WAVE];
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;  // (a whole workgroup)
  const u32 wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const u32 r = fdiv(lb, ntile);
  const u32 tile = lb - r * ntile.d;
  u32 oo = 0, sgrp = 0;
  bool active = zband_map(zb, r, oo, sgrp);  // zb counts groups of WPB segments
  oo *= ZK;
  active = active && oo < nouter;
  if (!active) oo = 0;
  const u32 sg = sgrp * WPB + wib;
  const int nk = ((int64_t)nouter - (int64_t)oo < ZK) ? (int)(nouter - oo) : ZK;  // levels this wave really has
  const int64_t i0 = ((int64_t)tile * WAVE + lane) * NV;
  const int64_t j0 = (int64_t)sg * SEG;
  active = active && i0 < nx && j0 < ny;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  int64_t nidx;
  bool edge;
  if (plx) { edge = (i0 == 0); nidx = edge ? ((bcx == XG_BC_PERIODIC) ? nx - 1 : 0) : i0 - 1; }
  else { edge = (i0 + NV == nx); nidx = edge ? ((bcx == XG_BC_PERIODIC) ? 0 : nx - 1) : i0 + NV; }
  const bool fill_edge = edge && (bcx == XG_BC_FILL);
  auto opx = [&](dv a, real n) -> dv {
    dv t;
    if (plx) {
      t[0] = op2<OP>(n, a[0]);
#pragma unroll
      for (int k = 1; k < NV; ++k) t[k] = op2<OP>(a[k - 1], a[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NV - 1; ++k) t[k] = op2<OP>(a[k], a[k + 1]);
      t[NV - 1] = op2<OP>(a[NV - 1], n);
    }
    return t;
  };
  // rows u = 0 .. SEG of the intermediate array that this segment's SEG outputs need (K8's numbering): row u is array row
  // j0 + u - ply; u = HALO is the one shared with the neighbouring segment, u = GIVE the one the neighbour needs from here
  const int HALO = ply ? 0 : SEG, GIVE = ply ? SEG : 0;
  // the halo row comes from the neighbouring wave when that wave exists in this workgroup and the row is an ordinary one
  const bool recv = active && nrow == SEG && (ply ? (wib > 0) : (wib + 1 < WPB && j0 + SEG < ny));
  bool rowfill[SEG + 1];
  int64_t qq[SEG + 1];
#pragma unroll
  for (int u = 0; u <= SEG; ++u) {
    int64_t k = j0 + ((u <= nrow) ? u : nrow);
    int64_t q = k - ply;
    bool f = false;
    if (q < 0) { f = (bcy == XG_BC_FILL); q = (bcy == XG_BC_PERIODIC) ? ny - 1 : 0; }
    else if (q >= ny) { f = (bcy == XG_BC_FILL); q = (bcy == XG_BC_PERIODIC) ? 0 : ny - 1; }
    rowfill[u] = f;
    qq[u] = q;
  }
  const bool nb_dpp = dpp != 0;
  const bool own_nb = !nb_dpp || edge || (plx ? lane == 0 : lane == WAVE - 1);
  auto beside = [&](dv v) -> real { return plx ? from_lane_below(v[NV - 1]) : from_lane_above(v[0]); };
  dv tx[ZK][SEG + 1];
  dv d3[SEG];
  if (active) {
    dv prz[ZK][SEG + 1], a1[SEG + 1], mid[SEG + 1];
    real nbz[ZK][SEG + 1], a1n[SEG + 1];
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      if (u == HALO && recv) continue;  // (wave-uniform)
#pragma unroll
      for (int kz = 0; kz < ZK; ++kz) {
        const real* pin = in + (o0 + oo + ((kz < nk) ? kz : nk - 1)) * ny * nx;  // a short last group repeats its last level (not stored)
        prz[kz][u] = *reinterpret_cast<const dv*>(pin + qq[u] * nx + i0);
      }
      a1[u] = *reinterpret_cast<const dv*>(m1 + qq[u] * nx + i0);
      mid[u] = *reinterpret_cast<const dv*>(m2 + qq[u] * nx + i0);  // X first: between the axes = (Y as the input, X as the output)
    }
    if (nb_dpp) {
#pragma unroll
      for (int u = 0; u <= SEG; ++u) {
        if (u == HALO && recv) continue;
#pragma unroll
        for (int kz = 0; kz < ZK; ++kz) nbz[kz][u] = beside(prz[kz][u]);
        a1n[u] = beside(a1[u]);
      }
    }
    if (own_nb) {  // ONE divergent block: the wave's end lane and the row's end lane
#pragma unroll
      for (int u = 0; u <= SEG; ++u) {
        if (u == HALO && recv) continue;
#pragma unroll
        for (int kz = 0; kz < ZK; ++kz) {
          const real* pin = in + (o0 + oo + ((kz < nk) ? kz : nk - 1)) * ny * nx;
          nbz[kz][u] = pin[qq[u] * nx + nidx];
        }
        a1n[u] = m1[qq[u] * nx + nidx];
      }
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) d3[u] = *reinterpret_cast<const dv*>(m3 + (j0 + ((u < nrow) ? u : 0)) * nx + i0);
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      if (u == HALO && recv) continue;
#pragma unroll
      for (int kz = 0; kz < ZK; ++kz) {
        const dv pr = prz[kz][u] * a1[u];      // the products at the input positions
        const real nb = nbz[kz][u] * a1n[u];
        dv t = opx(pr, fill_edge ? fillx : nb);
        t = (t / mid[u]) * mid[u];             // kept as written: not the identity in floating point
        tx[kz][u] = rowfill[u] ? splat<dv>(filly) : t;
      }
    }
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz) s_tx[wib][kz][lane] = tx[kz][GIVE];
  }
  __syncthreads();
  if (!active) return;
  if (recv) {
    const u32 from = ply ? wib - 1 : wib + 1;
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz) tx[kz][HALO] = s_tx[from][kz][lane];
  }
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    if (kz >= nk) break;
    real* po = out + ((o0 + oo + kz) * ny + j0) * nx + i0;
#pragma unroll
    for (int u = 0; u < SEG; ++u)
      if (u < nrow) {
        dv res = op2<OP>(tx[kz][u], tx[kz][u + 1]);
        res = res / d3[u];
        stg<dv, NTS>(po + u * nx, res);
      }
  }
}

// ------------------------------------------------------------------------------------------

struct StencilCall {
  const real* in; real* out; Geo g; int pad_lo, pad_hi, bc; real fill; const real* halo;
  const real* m_in; MIdx mi; const real* m_out; MIdx mo; hipStream_t st;
};

// marching kernel (one HBM read per cell whatever the plane size)
template <int OP, int V, int MET>
int launch_march(const StencilCall& c) {
  int seg = tune().seg < 1 ? 1 : tune().seg;
  const u32 nseg = (u32)((c.g.n_out + seg - 1) / seg);
  const u32 ntile = (u32)((c.g.inner + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 ntask = (u64)ntile * nseg * (u64)c.g.outer;
  const u64 nblocks = (ntask + WPB - 1) / WPB;
  if (nblocks == 0 || nblocks > 0x7fffffffull) return fail(XG_ERR_UNSUPPORTED, "launch of %llu blocks exceeds grid limits", nblocks);
  if (tune().nt_store)
    hipLaunchKernelGGL((k_stencil_strided<OP, V, MET, false, true>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), c.st, c.in, c.out, c.g, seg, nseg, ntile, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
  else
    hipLaunchKernelGGL((k_stencil_strided<OP, V, MET, false, false>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), c.st, c.in, c.out, c.g, seg, nseg, ntile, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
  return 0;
}

template <int OP, int MET>
int launch_contig_gen(const StencilCall& c) {
  const u64 Lo = (u64)c.g.n_out;
  u64 rows_per = 0xfffffff0ull / Lo;
  rows_per -= rows_per % NV;  // a multiple of NV rows per launch keeps every launch's first group aligned
  if (rows_per < (u64)NV) return -1;
  const FastDiv fLo = make_fastdiv(Lo);
  for (u64 row0 = 0; row0 < (u64)c.g.outer; row0 += rows_per) {
    const u64 nrows = ((u64)c.g.outer - row0 < rows_per) ? (u64)c.g.outer - row0 : rows_per;
    const u32 nelem = (u32)(nrows * Lo);
    const u32 nblk = (u32)((((u64)nelem + NV - 1) / NV + BLOCK - 1) / BLOCK);
    const u32 grid = ((nblk + 7) / 8) * 8;
    if (tune().nt_store)
      hipLaunchKernelGGL((k_stencil_contig_gen<OP, MET, true>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)row0, nelem, nblk, fLo, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
    else
      hipLaunchKernelGGL((k_stencil_contig_gen<OP, MET, false>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)row0, nelem, nblk, fLo, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
  }
  return 0;
}

// K1r launch: metrics present, aligned rows, outer dims (Z, Y) or (Y); returns 1 when it does not apply
template <int OP, int MET>
int launch_contig_rw(const StencilCall& c) {
  const Geo& g = c.g;
  const int R = (MET & 2) ? tune().contig_rw_mi : tune().contig_rw;
  if (MET == 0 || R <= 0 || tune().contig_rw <= 0 || !tune().nt_store || g.n_outer > 2 || g.n_outer < 1 || !g.idx32) return 1;
  if (g.n_in != g.n_out || g.n_in % NV || g.n_in >= (1ll << 28)) return 1;
  const u64 Z = g.n_outer == 2 ? (u64)g.outer_shape[0] : 1, Y = (u64)g.outer_shape[g.n_outer - 1];
  const int yd = g.n_outer - 1;  // index of the Y dim in the metric's outer strides
  const int64_t mi_z = (c.m_in && g.n_outer == 2) ? c.mi.outer[0] : 0, mi_y = c.m_in ? c.mi.outer[yd] : 0, mi_x = c.m_in ? c.mi.axis : 0;
  const int64_t mo_z = (c.m_out && g.n_outer == 2) ? c.mo.outer[0] : 0, mo_y = c.m_out ? c.mo.outer[yd] : 0, mo_x = c.m_out ? c.mo.axis : 0;
  auto vec_ok = [](const real* m, int64_t sz, int64_t sy, int64_t sx) { return !m || (aligned16(m) && sx == 1 && sz % NV == 0 && sy % NV == 0); };
  const int mal = vec_ok(c.m_in, mi_z, mi_y, mi_x) && vec_ok(c.m_out, mo_z, mo_y, mo_x);
  u32 RR = R >= 8 ? 8u : (R >= 4 ? 4u : (R >= 2 ? 2u : 1u));
  const bool bcast_z = tune().zband && Z >= 2 && mi_z == 0 && mo_z == 0;  // z-banding: all metrics broadcast along Z
  // (z-STACKED workgroups -- the 4 waves = 4 consecutive level groups of one row and x-tile, the metric vectors loaded by one
  // wave and handed on through LDS, as K4L does for weight rows -- built and measured: L1->L2 requests -22 %, but derivative
  // X 0.744 -> 0.689, metric_weighted X 0.687 -> 0.659, HBM reads 1.01 -> 1.12x: a workgroup then streams 8 level planes at
  // once; profiles/history/r03av_*)
  const bool zs = bcast_z && RR > 1 && tune().rw_zshare;
  if (RR == 8 && !zs) RR = 4;
  const u64 ntile = ((u64)g.n_in / NV + WAVE - 1) / WAVE;
  // band height (see launch_seg_n): one metric 32 rows (PMC reads 1.006x the algorithmic bytes), two metrics 8 rows
  // (16 rows: 1.17x -- the two metric bands no longer survive in the L2 next to the streams -- 8 rows: 1.017x; same speed)
  const u32 zb_base = (u32)(tune().zb_rows > 1 ? tune().zb_rows : 16);
  const u32 brows = (c.m_in && c.m_out) ? zb_base / 2 : zb_base * 2;
  ZBand zb = make_zband(false, 0, 0, 1);
  u64 YG = (Y + RR - 1) / RR, groups = Z * YG;
  if (zs) {  // band-major over (band of `brows` rows, level group, row)
    const u64 ZG = (Z + RR - 1) / RR;
    zb = make_zband(true, ZG, Y, brows);
    if (!zb.on) return 1;
    groups = ((Y + brows - 1) / brows) * brows * ZG;
  } else if (bcast_z) {
    const u32 band = (brows + RR - 1) / RR;  // row groups per band
    zb = make_zband(true, Z, YG, band);
    if (zb.on) groups = ((YG + band - 1) / band) * band * Z;
  }
  const u64 waves = groups * ntile;
  if (waves > MAX_ITEMS) return 1;
  const u32 nblk = (u32)((waves + WPB - 1) / WPB);
  const u32 grid = ((nblk + 7) / 8) * 8;
  const FastDiv fnt = make_fastdiv(ntile), fYG = make_fastdiv(YG);
#define XG_RW(R_, ZS_) hipLaunchKernelGGL((k_stencil_contig_rw<OP, MET, R_, ZS_>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, (u32)g.n_in, (u32)Z, (u32)Y, nblk, fnt, fYG, zb, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, mi_z, mi_y, mi_x, c.m_out, mo_z, mo_y, mo_x, mal, (tune().nt_load ? 1 : 0) | (tune().nb_dpp ? 2 : 0))
  if (zs) { if (RR == 8) XG_RW(8, true); else if (RR == 4) XG_RW(4, true); else XG_RW(2, true); }
  else { if (RR == 4) XG_RW(4, false); else if (RR == 2) XG_RW(2, false); else XG_RW(1, false); }
#undef XG_RW
  return 0;
}

template <int OP, int V, int MET>
int launch_contig(const StencilCall& c) {
  if constexpr (V == NV && MET != 0) {
    const int rc = launch_contig_rw<OP, MET>(c);
    if (rc != 1) return rc;
  }
  if (V == 1 && tune().contig_gen && aligned16(c.out) && c.g.n_in <= 0x7fffffffll && c.g.n_out <= 0x7fffffffll &&
      c.g.outer * c.g.n_out >= 2) {
    const int rc = launch_contig_gen<OP, MET>(c);
    if (rc >= 0) return rc;
  }
  const u64 per = (u64)((c.g.n_out + V - 1) / V);
  if (per > MAX_ITEMS || c.g.n_in > 0x7fffffffll) return fail(XG_ERR_UNSUPPORTED, "row of %llu items too long", per);
  const FastDiv fper = make_fastdiv(per);
  const u64 rows_per = MAX_ITEMS / per;
  // z-banding: outer dims (Z, Y) with every metric broadcast along Z, whole problem in one launch
  const u32 ZB_ROWS = (u32)(tune().zb_rows > 0 ? tune().zb_rows : 16);
  bool zb_ok = MET != 0 && tune().zband && c.g.n_outer == 2 && (!c.m_in || c.mi.outer[0] == 0) &&
               (!c.m_out || c.mo.outer[0] == 0);
  u64 work_rows = (u64)c.g.outer;
  ZBand zb = make_zband(false, 0, 0, 1);
  if (zb_ok) {
    const u64 Z = (u64)c.g.outer_shape[0], Y = (u64)c.g.outer_shape[1];
    const u64 padded = ((Y + ZB_ROWS - 1) / ZB_ROWS) * ZB_ROWS * Z;
    if (padded <= rows_per) { zb = make_zband(true, Z, Y, ZB_ROWS); if (zb.on) work_rows = padded; }
  }
  for (u64 row0 = 0; row0 < work_rows; row0 += rows_per) {
    const u32 nrows = (u32)((work_rows - row0 < rows_per) ? work_rows - row0 : rows_per);
    const u32 nblk = (u32)(((u64)nrows * per + BLOCK - 1) / BLOCK);
    const u32 grid = ((nblk + 7) / 8) * 8;
    if (tune().nt_store)
      hipLaunchKernelGGL((k_stencil_contig<OP, V, MET, true>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)row0, nrows, nblk, fper, zb, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, (tune().nt_load ? 1 : 0) | (tune().nb_dpp ? 2 : 0));
    else
      hipLaunchKernelGGL((k_stencil_contig<OP, V, MET, false>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)row0, nrows, nblk, fper, zb, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, (tune().nt_load ? 1 : 0) | (tune().nb_dpp ? 2 : 0));
  }
  return 0;
}

// rows of the strided axis per wave-task.  One row (+ its halo row, an L2 hit thanks to the banded order) is the
// most robust choice: measured through the raw ABI on a cool and on a warm MI355X (tools/abi_ab.py), Y stencils
// 78.9 / 78.0 % with SEG = 1, 79.7 / 76.2 % with 2, 78.7 / 73.9 % with 4; Z (column chunks) 79.1 / 77.0 %, 77.7 / 73.9 %,
// 76.1 / 70.5 % -- the fewer rows a thread carries, the less it loses when the device is warm.
constexpr int STENCIL_SEG = 1;
// With metrics the balance shifts: the division / products sit between a wave's loads and its store, and a
// wave that carries 2 rows (3 field loads, 2 divisor loads in flight at once) hides that arithmetic behind twice
// the bytes -- r01 measured derivative along Y at 74 % with four rows per task against 68 % with one.

inline bool metric_vec_ok(const Geo& g, const real* m, const MIdx& mm) {
  if (!m) return true;
  if (!aligned16(m) || g.n_inner != 1 || mm.inner[0] != 1 || mm.axis % NV) return false;
  for (int d = 0; d < g.n_outer; ++d)
    if (mm.outer[d] % NV) return false;
  return true;
}

// K2Sy: the plain two-point operator along a strided axis with short rows (diff / interp / min / max along Y of (Z, Y, X)),
// Y-STACKED: the WPB waves of a workgroup are WPB CONSECUTIVE ROWS of one x-tile.  K2S gives a wave one output row and two
// loads -- its row and the row below, which the neighbouring task loaded too (an L2 hit, but a second trip through the
// address and L1 pipeline for every output).  Here every wave loads ONE row, the upper one of its pair, and receives the
// lower one from the wave below through LDS; only the lowest wave of a workgroup loads both: 1.25 loads per output row.
template <int OP, bool NTS, int NW>
__global__ __launch_bounds__(NW * WAVE) void k_stencil_strided_ys(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int64_t o0, u32 nouter, u32 nblk, FastDiv ntile,
    FastDiv ngrp, int pad_lo, int bc, real fill, const real* __restrict__ halo) {
  typedef dv T;
  __shared__ T s_row[NW][WAVE];
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;  // (whole workgroups)
  const u32 wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const u32 r = fdiv(lb, ntile);
  const u32 tile = lb - r * ntile.d;
  const u32 oo = fdiv(r, ngrp);
  if (oo >= nouter) return;
  const int64_t j = (int64_t)(r - oo * ngrp.d) * NW + wib;  // this wave's output row
  const int64_t o = o0 + oo, inner = g.inner;
  const int64_t x = ((int64_t)tile * WAVE + lane) * NV;
  const bool active = j < g.n_out && x < inner;
  // padded rows j and j + 1 -> input rows (wave-uniform), fill flags, pre-gathered halo rows
  auto source = [&](int64_t k, bool& f, bool& h) -> int64_t {
    int64_t q = k - pad_lo;
    f = false;
    h = false;
    if (q < 0 || q >= g.n_in) {
      f = (bc == XG_BC_FILL);
      if (bc == XG_BC_HALO) { h = true; q = (q < 0) ? 0 : pad_lo; }
      else q = (q < 0) ? ((bc == XG_BC_PERIODIC) ? g.n_in - 1 : 0) : ((bc == XG_BC_PERIODIC) ? 0 : g.n_in - 1);
    }
    return q;
  };
  T lo = splat<T>(real(0)), hi = splat<T>(real(0));
  bool f0 = false, f1 = false, h0 = false, h1 = false;
  if (active) {
    const real* pin = in + (o * g.n_in) * inner + x;
    const real* phalo = halo + (o * (g.n_out - g.n_in + 1)) * inner + x;
    const int64_t q0 = source(j, f0, h0), q1 = source(j + 1, f1, h1);
    hi = *reinterpret_cast<const T*>((h1 ? phalo : pin) + q1 * inner);
    if (wib == 0) lo = *reinterpret_cast<const T*>((h0 ? phalo : pin) + q0 * inner);  // the row below the workgroup: an L2 hit
    s_row[wib][lane] = hi;
  }
  __syncthreads();
  if (!active) return;
  if (wib > 0) lo = s_row[wib - 1][lane];  // (the wave below is active: its row index is smaller)
  const T a = f0 ? splat<T>(fill) : lo, b = f1 ? splat<T>(fill) : hi;
  if (NTS) stg_drop<T>(out + (o * g.n_out + j) * inner + x, op2<OP>(a, b));  // (as in the flat X kernel: +0.5-0.9 points)
  else stg<T, false>(out + (o * g.n_out + j) * inner + x, op2<OP>(a, b));
}

// K2Sm: K2S WITH metrics, z-banded and z-shared, Y-STACKED (DESIGN rule 14): the WPB waves of a workgroup are WPB consecutive
// SEG-row segments of one x-tile and one group of ZK levels.  A wave loads only the SEG upper rows of its SEG + 1 (and their
// input-metric rows); the lowest one -- already multiplied by its metric, or replaced by the fill value -- comes from the
// wave below through LDS: the very product that wave formed for its own top row, so the same bits.  Only the lowest wave of a
// workgroup loads its row 0 itself.
template <int OP, int MET, int SEG, int ZK>
__global__ __launch_bounds__(BLOCK) void k_stencil_strided_ysm(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 nouter, u32 nblk, FastDiv ntile, u32 nseg, ZBand zb,
    int pad_lo, int bc, real fill, const real* __restrict__ halo, const real* __restrict__ m_in, MIdx mi,
    const real* __restrict__ m_out, MIdx mo, int mal) {
  typedef dv T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  __shared__ T s_p[WPB][ZK][WAVE];
  auto ldmv = [&](const real* m, int64_t off, int64_t step) -> T {
    if (mal & 1) return *reinterpret_cast<const T*>(m + off);
    return ldm<T>(m, off, step);
  };
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;  // (whole workgroups)
  const u32 wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const u32 r = fdiv(lb, ntile);
  const u32 tile = lb - r * ntile.d;
  u32 oo, ssg, ostep;
  if (!zband_map(zb, r, oo, ssg)) return;  // band-major order over (band of super-segments, level group, super-segment)
  const u32 zl = oo * ZK;
  zband_face(zb, zl, ssg, oo, ostep);  // (faces under the levels: `nouter` counts LEVELS, the levels of a task lie `ostep` apart)
  const u32 sg = ssg * WPB + wib;
  const int64_t o = oo;
  const int nk = ((int64_t)nouter - (int64_t)zl < ZK) ? (int)(nouter - zl) : ZK;  // levels this group really has
  const int64_t inner = g.inner;
  const int64_t x = ((int64_t)tile * WAVE + lane) * NV;
  const bool active = sg < nseg && x < inner;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (g.n_out - j0 < SEG) ? g.n_out - j0 : SEG;  // rows this segment really has (<= 0: none)
  T p[ZK][SEG + 1];
  T dm[SEG];
  if (active) {
    int64_t mib = 0, mob = 0, mis = 0, mos = 0;
    const bool ui = HAS_MI && (mal & 2), uo = HAS_MO && (mal & 4);  // row-uniform metrics: scalar loads
    int64_t mibu = 0, mobu = 0;
    if (HAS_MI) {
      if (ui) mibu = outer_off32(g, mi, (u32)o);
      else {
        inner_off_step32(g, mi, (u32)x, true, mib, mis);
        mib += outer_off32(g, mi, (u32)o);
      }
    }
    if (HAS_MO) {
      if (uo) mobu = outer_off32(g, mo, (u32)o) + j0 * mo.axis;
      else {
        inner_off_step32(g, mo, (u32)x, true, mob, mos);
        mob += outer_off32(g, mo, (u32)o) + j0 * mo.axis;
      }
    }
    // padded index k = j0 + u  ->  input row q (wave-uniform), fill flag, pre-gathered halo row
    bool ff[SEG + 1], hh[SEG + 1];
    int64_t qq[SEG + 1];
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      int64_t k = j0 + ((u <= nrow) ? u : nrow);  // clamp inside the padded range for short tails
      int64_t q = k - pad_lo;
      ff[u] = false;
      hh[u] = false;
      if (q < 0 || q >= g.n_in) {
        ff[u] = (bc == XG_BC_FILL);
        if (bc == XG_BC_HALO) { hh[u] = true; q = (q < 0) ? 0 : pad_lo; }
        else q = (q < 0) ? ((bc == XG_BC_PERIODIC) ? g.n_in - 1 : 0) : ((bc == XG_BC_PERIODIC) ? 0 : g.n_in - 1);
      }
      qq[u] = q;
    }
    const int ulo = (wib == 0) ? 0 : 1;  // the lowest wave of the workgroup loads its row 0 itself
    T v[ZK][SEG + 1], wm[SEG + 1];
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz) {
      const int64_t ok = o + (int64_t)((kz < nk) ? kz : nk - 1) * ostep;  // a short last group repeats its last level (not stored)
      const real* pin = in + (ok * g.n_in) * inner + x;
      const real* phalo = halo + (ok * (g.n_out - g.n_in + 1)) * inner + x;
#pragma unroll
      for (int u = 0; u <= SEG; ++u)
        if (u >= ulo) v[kz][u] = *reinterpret_cast<const T*>((hh[u] ? phalo : pin) + qq[u] * inner);
    }
    if (HAS_MI) {
#pragma unroll
      for (int u = 0; u <= SEG; ++u)
        if (u >= ulo) wm[u] = ui ? splat<T>(m_in[mibu + qq[u] * mi.axis]) : ldmv(m_in, mib + qq[u] * mi.axis, mis);
    }
    if (HAS_MO) {
#pragma unroll
      for (int u = 0; u < SEG; ++u) dm[u] = uo ? splat<T>(m_out[mobu + ((u < nrow) ? u : 0) * mo.axis]) : ldmv(m_out, mob + ((u < nrow) ? u : 0) * mo.axis, mos);
    }
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz) {
#pragma unroll
      for (int u = 0; u <= SEG; ++u) {
        if (u >= ulo) {
          T t = v[kz][u];
          if (HAS_MI && !hh[u]) t = t * wm[u];  // (pre-gathered halo rows are those of the PRODUCT: not weighted again)
          p[kz][u] = ff[u] ? splat<T>(fill) : t;
        }
      }
      s_p[wib][kz][lane] = p[kz][SEG];  // my top row = the row 0 of the wave above
    }
  }
  __syncthreads();
  if (!active) return;
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    if (kz >= nk) break;
    if (wib > 0) p[kz][0] = s_p[wib - 1][kz][lane];  // (the wave below is active: its segment index is smaller)
    real* pout = out + ((o + (int64_t)kz * ostep) * g.n_out + j0) * inner + x;
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      if (u < nrow) {
        T res = op2<OP>(p[kz][u], p[kz][u + 1]);
        if (HAS_MO) res = res / dm[u];
        stg_s<T, true>(pout + u * inner, res);  // (`sc1 nt`: +0.2 points)
      }
    }
  }
}

// (K2Sy / K2Sm with 32-bit index arithmetic and without the generic outer-offset peel -- 542 -> 317 scalar instructions per
// K2Sm wave, 203 -> 164 per K2Sy wave -- measured against the unchanged K2S in alternating processes of one session: K2Sy's
// lead over K2S stays at +1.7 points, K2Sm's SHRINKS from +1.7 to +0.7; the scalar unit is not what bounds these kernels, the
// 64-bit forms stay.  profiles/history/r03ax_ab_lean32_old_new.jsonl)
// K2Sm launch: the geometry tests are launch_seg_n's (z-banding, one outer dim, metrics broadcast along it, 16-B lane vectors)
// z-banding geometry of a strided-axis launch with metrics: ONE outer dim along which every metric is broadcast (levels), or
// TWO with the metrics broadcast along the slower one only -- (Z, face | Y | X) fields with per-face metrics (rule 17).
// Returns false when the metrics do not allow banding; else the number of levels and of faces under each level.
inline bool zband_levels(const StencilCall& c, u64* levels, u64* faces) {
  if (c.g.n_outer < 1 || c.g.n_outer > 2) return false;
  if ((c.m_in && c.mi.outer[0] != 0) || (c.m_out && c.mo.outer[0] != 0)) return false;
  *levels = (u64)c.g.outer_shape[0];
  *faces = c.g.n_outer == 2 ? (u64)c.g.outer_shape[1] : 1;
  return true;
}

template <int OP, int MET, int SEG, int ZK>
bool launch_ysm(const StencilCall& c) {
  const u64 ntile = (u64)((c.g.inner + (int64_t)WAVE * NV - 1) / ((int64_t)WAVE * NV));
  const u64 nseg = (u64)((c.g.n_out + SEG - 1) / SEG);
  if (nseg < 2 * WPB || nseg > 0x7fffffffull) return false;
  const u64 nsseg = (nseg + WPB - 1) / WPB;  // super-segments: WPB segments, one workgroup per x-tile and level group
  int mal = (metric_vec_ok(c.g, c.m_in, c.mi) && metric_vec_ok(c.g, c.m_out, c.mo)) ? 1 : 0;
  auto row_uniform = [&](const real* m, const MIdx& mm) {
    if (!m) return false;
    for (int d = 0; d < c.g.n_inner; ++d)
      if (mm.inner[d] != 0) return false;
    return true;
  };
  if (tune().met_scalar) mal |= (row_uniform(c.m_in, c.mi) ? 2 : 0) | (row_uniform(c.m_out, c.mo) ? 4 : 0);
  const u32 zb_base = (u32)(tune().zb_rows > 1 ? tune().zb_rows : 16);
  // one metric: bands of 2 x zb_rows = 32 rows hold here (reads 1.062 -> 1.032x, +0.6 points, profiles/history/r03bh_pmc_dy_bands.jsonl):
  // the output lines are dropped from the L2 as they are written (rule 16), the band's one metric plane and the halo rows stay
  // two metrics: 16-row bands hold in the y-stacked form, where K2S needs 8 (round 4: reads 6.00 -> 5.72 GB, traffic 1.064 ->
  // 1.038x at the same speed, nine placements paired; profiles/history/r04h_ab_iymw_*.log)
  const u32 zbr = (c.m_in && c.m_out) ? zb_base : zb_base * 2;
  const u32 per = (u32)(SEG * WPB);
  const u32 ZB_SS = (zbr + per - 1) / per;  // band height in super-segments (at least one)
  u64 levels = 0, faces = 1;
  if (!zband_levels(c, &levels, &faces)) return false;
  const u64 rows = nsseg * faces;           // super-segments under one level (whole faces: none straddles two)
  const u64 padded = ((rows + ZB_SS - 1) / ZB_SS) * ZB_SS;
  const u64 zgroups = (levels + ZK - 1) / ZK;
  const u64 nwg = padded * zgroups * ntile;
  ZBand zb = make_zband(true, zgroups, rows, ZB_SS, faces);
  if (!zb.on || nwg > MAX_ITEMS) return false;
  const u32 nblk = (u32)nwg;
  const u32 grid = ((nblk + 7) / 8) * 8;
  hipLaunchKernelGGL((k_stencil_strided_ysm<OP, MET, SEG, ZK>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (u32)levels, nblk, make_fastdiv(ntile), (u32)nseg, zb, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, mal);
  return true;
}

template <int OP, int V, int MET, int SEG, int ZK = 1>
int launch_seg_n(const StencilCall& c) {
  const u64 ntile = (u64)((c.g.inner + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 nseg = (u64)((c.g.n_out + SEG - 1) / SEG);
  const Chunk noch = make_chunk(0, 1, 0);
  const Chunk ck = make_chunk(ntile, nseg, ntile > (u64)tune().seg_max_tiles ? (u32)tune().zchunk : 0u);
  const u64 per_outer = ck.on ? (u64)ck.nchunk * ck.ch.d * nseg : ntile * nseg;  // waves per outer index
  if (per_outer > MAX_ITEMS) return launch_march<OP, V, MET>(c);  // (never the case below 2^31 cells per outer index)
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  int mal = (V > 1 && MET != 0 && metric_vec_ok(c.g, c.m_in, c.mi) && metric_vec_ok(c.g, c.m_out, c.mo)) ? 1 : 0;
  // a metric that does not vary along the lanes (drF(Z) under a (Z, Y, X) field, the usual vertical metric): one value
  // per row, wave-uniform -- a scalar load instead of a vector load of 64 equal addresses (bits 1 / 2: m_in / m_out)
  auto row_uniform = [&](const real* m, const MIdx& mm) {
    if (!m) return false;
    for (int d = 0; d < c.g.n_inner; ++d)
      if (mm.inner[d] != 0) return false;
    return true;
  };
  if (tune().met_scalar && MET != 0) mal |= (row_uniform(c.m_in, c.mi) ? 2 : 0) | (row_uniform(c.m_out, c.mo) ? 4 : 0);
  // z-banding: a single outer dim along which every metric is broadcast, one launch (bands of 16 rows)
  // band height: `zb_rows` rows when two metrics share the XCD's L2, twice that for one -- a band boundary costs one
  // halo-row re-read from HBM per level (PMC: +6 % reads at 16 rows), a band must stay L2-resident for all levels
  // (round 3, PMC per band height, profiles/history/r03g_*: one metric 32 rows 1.106x the algorithmic reads, 16 rows 1.064x =
  // the halo row; two metrics 16 rows 1.14x, 8 rows 1.127x = the halo row; same speed within 0.5 % => 16 / 8 rows)
  const u32 zb_base = (u32)(tune().zb_rows > 1 ? tune().zb_rows : 16);
  const u32 zbr = (c.m_in && c.m_out) ? zb_base / 2 : zb_base;
  const u32 ZB_SEGS = (zbr + SEG - 1) / SEG;
  u64 levels = 0, faces = 1;  // one outer dim (levels), or (levels, faces) with per-face metrics shared by the levels
  const bool zb_ok = !ck.on && MET != 0 && tune().zband && zband_levels(c, &levels, &faces);
  if (ZK > 1 && !tune().nt_store) return launch_seg_n<OP, V, MET, SEG, 1>(c);  // z-shared tasks exist with non-temporal stores only
  if (zb_ok) {
    const u64 rows = nseg * faces;  // segments under one level
    const u64 padded_segs = ((rows + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS;
    const u64 zgroups = (levels + ZK - 1) / ZK;  // ZK levels per wave share the metric rows
    const u64 waves = padded_segs * zgroups * ntile;
    ZBand zb = make_zband(true, zgroups, rows, ZB_SEGS, faces);
    if (zb.on && waves <= MAX_ITEMS) {
      const u32 nblk = (u32)((waves + WPB - 1) / WPB);
      const u32 grid = ((nblk + 7) / 8) * 8;
      if (tune().nt_store)
        hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, true, SEG, ZK>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)0, (u32)levels, nblk, fnt, fns, zb, noch, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, mal);
      else
        hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, false, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)0, (u32)levels, nblk, fnt, fns, zb, noch, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, mal);
      return 0;
    }
  }
  if (ZK > 1) return launch_seg_n<OP, V, MET, SEG, 1>(c);  // not z-banded: no shared metric rows
  // (K2Sy extended to metrics that are not z-banded and to whole-plane rows in column chunks, measured in one process:
  // 3-D divisor along Y +0.6 points, plain diff along Z 0.768 -> 0.740, derivative Z 0.749 -> 0.657, metric_weighted Z
  // 0.739 -> 0.612 -- four levels per workgroup make five plane streams per XCD; not kept, profiles/history/r03ah_ab_ys_ext.jsonl)
  if (MET == 0 && V == NV && SEG == 1 && !ck.on && tune().seg_ys && c.g.n_out >= 2 * WPB) {  // K2Sy: y-stacked workgroups
    // (8 waves per workgroup -- 1.125 loads per output row -- measured slower: 0.789 against 0.802, profiles/history/r03aa_*; two
    // x-tiles per wave, the scalar row logic paid once per 2 KB: 0.780 -> 0.753, profiles/history/r03au_ab_k2sy_xt.jsonl; non-temporal
    // loads for the rows nobody reads again: 0.785 / 0.784, profiles/history/r03az_ab_k2sy_nt.jsonl)
    const u64 nw = WPB;
    const u64 ngrp = ((u64)c.g.n_out + nw - 1) / nw, per = ngrp * ntile;  // workgroups per outer index
    if (per <= MAX_ITEMS) {
      const FastDiv fng = make_fastdiv(ngrp);
      const u64 ostep = MAX_ITEMS / per;
      for (int64_t o0 = 0; o0 < c.g.outer; o0 += (int64_t)ostep) {
        const u32 nouter = (u32)((c.g.outer - o0 < (int64_t)ostep) ? c.g.outer - o0 : (int64_t)ostep);
        const u32 nblk = (u32)((u64)nouter * per);
        const u32 grid = ((nblk + 7) / 8) * 8;
        if (tune().nt_store) hipLaunchKernelGGL((k_stencil_strided_ys<OP, true, WPB>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, o0, nouter, nblk, fnt, fng, c.pad_lo, c.bc, c.fill, c.halo);
        else hipLaunchKernelGGL((k_stencil_strided_ys<OP, false, WPB>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, o0, nouter, nblk, fnt, fng, c.pad_lo, c.bc, c.fill, c.halo);
      }
      return 0;
    }
  }
  const ZBand zoff = make_zband(false, 0, 0, 1);
  for (int64_t o0 = 0; o0 < c.g.outer; o0 += (int64_t)outer_per) {
    const u32 nouter = (u32)((c.g.outer - o0 < (int64_t)outer_per) ? c.g.outer - o0 : (int64_t)outer_per);
    const u32 nblk = (u32)(((u64)nouter * per_outer + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
    if (tune().nt_store)
      hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, true, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, o0, nouter, nblk, fnt, fns, zoff, ck, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, mal);
    else
      hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, false, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, o0, nouter, nblk, fnt, fns, zoff, ck, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo, mal);
  }
  return 0;
}

template <int OP, int V, int MET>
int launch_seg(const StencilCall& c) {
  if constexpr (MET != 0 && V > 1) {
    // levels per wave-task sharing the metric rows: 4 with two metrics, 2 with a divisor only (A/B on a slow and a fast
    // box: derivative Y 66.4 -> 69.5 % / 73.5 -> 74.3 %; with two metrics 4 stays ahead: 66.5 against 64.6 %)
    // rows per wave-task: 4 with two metrics (metric_weighted Y 71.7 -> 73.5 % / 72.4 -> 73.9 % on two boxes), 2 with one
    const int ms = (MET == 3 ? tune().met_seg : tune().met_seg1), zk = tune().nt_store ? (MET == 3 ? tune().met_zk : tune().met_zk1) : 1;
    // K2Sm (y-stacked workgroups) where launch_seg_n would z-band: `met_ys1` / `met_ys2` = 10 * rows + levels per wave.
    // One metric (derivative Y): 1 row x 2 levels 0.733 -> 0.750 in one process, every other shape +0.7..0.9; two metrics
    // (metric_weighted Y): 1 x 8 within +-1 point of K2S, everything else behind it => off in round 3; round 4, paired over nine
    // buffer placements: 1 x 4 is 0.9 % ahead of K2S with 8-row bands and level with it at 16 rows, where it re-reads half
    // as many halo rows => `met_ys2` = 14
    const int ys = (MET == 3) ? tune().met_ys2 : tune().met_ys1;
    const u64 ntile_ = (u64)((c.g.inner + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
    u64 lv_ = 0, fc_ = 1;
    if (ys && tune().nt_store && tune().zband && ntile_ <= (u64)tune().seg_max_tiles && zband_levels(c, &lv_, &fc_)) {
      bool done = false;
      switch (ys) {
        case 12: done = launch_ysm<OP, MET, 1, 2>(c); break;
        case 14: done = launch_ysm<OP, MET, 1, 4>(c); break;
        case 18: done = launch_ysm<OP, MET, 1, 8>(c); break;
        case 22: done = launch_ysm<OP, MET, 2, 2>(c); break;
        // (2 x 4, 2 x 8, 4 x 2, 4 x 4 measured too: no better than K2S, profiles/history/r03af_*, r03ag_*)
        default: break;
      }
      if (done) return 0;
    }
    if (zk >= 8 && ms >= 2) return launch_seg_n<OP, V, MET, 2, 8>(c);
    if (zk >= 4) {
      if (ms >= 4) return launch_seg_n<OP, V, MET, 4, 4>(c);
      if (ms >= 2) return launch_seg_n<OP, V, MET, 2, 4>(c);
      return launch_seg_n<OP, V, MET, 1, 4>(c);
    }
    if (zk >= 2) {
      if (ms >= 4) return launch_seg_n<OP, V, MET, 4, 2>(c);
      if (ms >= 2) return launch_seg_n<OP, V, MET, 2, 2>(c);
      return launch_seg_n<OP, V, MET, 1, 2>(c);
    }
    if (ms >= 4) return launch_seg_n<OP, V, MET, 4>(c);
    if (ms >= 2) return launch_seg_n<OP, V, MET, 2>(c);
  }
  return launch_seg_n<OP, V, MET, STENCIL_SEG>(c);
}

// flat NV-group walk for misaligned rows of a strided axis (no metrics); returns 1 when it does not apply
template <int OP, int MET>
int launch_strided_gen(const StencilCall& c) {
  const Geo& g = c.g;
  if (!g.idx32) return 1;
  const u64 gmax = (u64)((g.inner + NV - 1) / NV);     // groups a row can hold
  const u64 ntile = (gmax + WAVE - 1) / WAVE;
  const u64 nrow = (u64)g.n_out;
  const Chunk ck = make_chunk(ntile, nrow, ntile > (u64)tune().seg_max_tiles ? (u32)tune().zchunk : 0u);
  const u64 per_outer = ck.on ? (u64)ck.nchunk * ck.ch.d * nrow : ntile * nrow;  // waves per outer index
  if (per_outer > MAX_ITEMS) return 1;
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nrow);
  const u64 outer_per = MAX_ITEMS / per_outer;
  for (int64_t o0 = 0; o0 < g.outer; o0 += (int64_t)outer_per) {
    const u32 nouter = (u32)((g.outer - o0 < (int64_t)outer_per) ? g.outer - o0 : (int64_t)outer_per);
    const u32 nblk = (u32)(((u64)nouter * per_outer + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
    if (tune().nt_store)
      hipLaunchKernelGGL((k_stencil_strided_gen<OP, MET, true>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, g, o0, nouter, nblk, fnt, fns, ck, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
    else
      hipLaunchKernelGGL((k_stencil_strided_gen<OP, MET, false>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, g, o0, nouter, nblk, fnt, fns, ck, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
  }
  return 0;
}
template <int OP>
int strided_gen_met(int met, const StencilCall& c) {
#ifdef XG_INT  // integer build: no metrics (stencil1d_impl refuses them)
  return launch_strided_gen<OP, 0>(c);
#endif
  switch (met) {
    case 0: return launch_strided_gen<OP, 0>(c);
    case 1: return launch_strided_gen<OP, 1>(c);
    case 2: return launch_strided_gen<OP, 2>(c);
    default: return launch_strided_gen<OP, 3>(c);
  }
}
int strided_gen_dispatch(int op, int met, const StencilCall& c) {
  switch (op) {
    case XG_OP_DIFF: return strided_gen_met<XG_OP_DIFF>(met, c);
    case XG_OP_INTERP: return strided_gen_met<XG_OP_INTERP>(met, c);
    case XG_OP_MIN: return strided_gen_met<XG_OP_MIN>(met, c);
#ifdef XG_INT
    case XG_OP_MINU: return strided_gen_met<XG_OP_MINU>(met, c);
    case XG_OP_MAXU: return strided_gen_met<XG_OP_MAXU>(met, c);
#endif
    default: return strided_gen_met<XG_OP_MAX>(met, c);
  }
}

enum StencilKind { KIND_CONTIG = 0, KIND_LIN = 1, KIND_MARCH = 2 };

template <int OP, int V, int MET>
int stencil_kind(int kind, const StencilCall& c) {
  if (kind == KIND_CONTIG) return launch_contig<OP, V, MET>(c);
  if (kind == KIND_LIN) return launch_seg<OP, V, MET>(c);
  return launch_march<OP, V, MET>(c);
}
template <int OP, int V>
int stencil_met(int met, int kind, const StencilCall& c) {
#ifdef XG_INT
  return stencil_kind<OP, V, 0>(kind, c);
#endif
  switch (met) {
    case 0: return stencil_kind<OP, V, 0>(kind, c);
    case 1: return stencil_kind<OP, V, 1>(kind, c);
    case 2: return stencil_kind<OP, V, 2>(kind, c);
    default: return stencil_kind<OP, V, 3>(kind, c);
  }
}
template <int OP>
int stencil_vec(int V, int met, int kind, const StencilCall& c) {
  return V > 1 ? stencil_met<OP, NV>(met, kind, c) : stencil_met<OP, 1>(met, kind, c);
}
int stencil_dispatch(int op, int V, int met, int kind, const StencilCall& c) {
  switch (op) {
    case XG_OP_DIFF: return stencil_vec<XG_OP_DIFF>(V, met, kind, c);
    case XG_OP_INTERP: return stencil_vec<XG_OP_INTERP>(V, met, kind, c);
    case XG_OP_MIN: return stencil_vec<XG_OP_MIN>(V, met, kind, c);
#ifdef XG_INT
    case XG_OP_MINU: return stencil_vec<XG_OP_MINU>(V, met, kind, c);
    case XG_OP_MAXU: return stencil_vec<XG_OP_MAXU>(V, met, kind, c);
#endif
    default: return stencil_vec<XG_OP_MAX>(V, met, kind, c);
  }
}


}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
#ifndef XG_INT  // two axes in one pass: float builds only (integer interp leaves the integer domain between the axes)
static int stencil2d_impl(int op, const real* in, real* out, const int64_t* shape, int ndim, int order,
                          int padx_lo, int padx_hi, int bc_x, real fill_x, int pady_lo, int pady_hi, int bc_y,
                          real fill_y, const real* m1, const real* m2, const real* m3, void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if (op < XG_OP_DIFF || op > XG_OP_MAX) return fail(XG_ERR_INVALID, "unknown op %d", op);
  if (ndim < 2 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [2,%d]", ndim, XG_MAX_NDIM);
  if (order != 0 && order != 1) return fail(XG_ERR_INVALID, "order must be 0 (X then Y) or 1 (Y then X)");
  if (padx_lo + padx_hi != 1 || pady_lo + pady_hi != 1 || ((padx_lo | padx_hi | pady_lo | pady_hi) & ~1))
    return fail(XG_ERR_UNSUPPORTED, "fused 2-D stencil needs length-preserving pads (1,0) or (0,1) on both axes");
  if (bc_x < XG_BC_PERIODIC || bc_x > XG_BC_EXTEND || bc_y < XG_BC_PERIODIC || bc_y > XG_BC_EXTEND)
    return fail(XG_ERR_INVALID, "fused 2-D stencil needs a boundary mode on both axes");
  const int64_t ny = shape[ndim - 2], nx = shape[ndim - 1];
  int64_t outer = 1;
  for (int d = 0; d < ndim - 2; ++d) outer *= shape[d];
  if (outer == 0 || ny == 0 || nx == 0) return XG_OK;
  const bool met = m1 != nullptr;
  if (met && (!aligned16(m1) || !aligned16(m2) || !aligned16(m3))) return fail(XG_ERR_UNSUPPORTED, "fused 2-D stencil: metric planes must be 16-byte aligned");
  if (nx % NV || !aligned16(in) || !aligned16(out)) return fail(XG_ERR_UNSUPPORTED, "fused 2-D stencil needs an X extent that is a multiple of the 16-byte lane vector");
  constexpr int SEG = XG_FUSED_SEG;
  const u64 ntile = (u64)((nx + NV * WAVE - 1) / (NV * WAVE));
  const u64 nseg = (u64)((ny + SEG - 1) / SEG);
  const u64 per_outer = ntile * nseg;
  if (per_outer > MAX_ITEMS) return fail(XG_ERR_UNSUPPORTED, "extent too large for the 2-D stencil kernel");
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
  for (int64_t o0 = 0; o0 < outer; o0 += (int64_t)outer_per) {
    const u32 nouter = (u32)((outer - o0 < (int64_t)outer_per) ? outer - o0 : (int64_t)outer_per);
    u32 nblk = (u32)(((u64)nouter * per_outer + WPB - 1) / WPB);
    ZBand zb = make_zband(false, 0, 0, 1);
    const int ZK2 = tune().met_zk2 >= 4 ? 4 : 2;  // levels per wave-task sharing the metric rows
    bool ys = false;  // K8y: X first with metrics, y-stacked workgroups handing the shared intermediate row on through LDS
    if (met && tune().zband && nouter >= 2) {  // the metric planes are shared by the outer indices: band-major order
      const u32 B = ((u32)(tune().zb_rows > 0 ? tune().zb_rows : 16) + SEG - 1) / SEG;
      const u64 groups = ((u64)nouter + ZK2 - 1) / ZK2;
      if (order == 0 && nts && ZK2 == 4 && tune().met_ys && groups >= 2) {
        const u64 nsg = (nseg + WPB - 1) / WPB;                 // groups of WPB segments = workgroups per (level group, x-tile)
        const u32 Bg = (B + WPB - 1) / WPB;
        const u64 blocks = ((nsg + Bg - 1) / Bg) * Bg * groups * ntile;
        if (blocks < 0x7ffffff0ull) {
          zb = make_zband(true, groups, nsg, Bg);
          if (zb.on) { nblk = (u32)blocks; ys = true; }
        }
      }
      const u64 padded = ((nseg + B - 1) / B) * B * groups * ntile;
      if (!ys && groups >= 2 && padded <= MAX_ITEMS) {
        zb = make_zband(true, groups, nseg, B);
        if (zb.on) nblk = (u32)((padded + WPB - 1) / WPB);
      }
    }
    const u32 grid = ((nblk + 7) / 8) * 8;
    if (ys) {
#define XG_YS(O) hipLaunchKernelGGL((k_stencil2d_ys<O, true, SEG, 4>), dim3(grid), dim3(BLOCK), 0, st, in, out, o0, nouter, nblk, ny, nx, fnt, tune().nb_dpp, padx_lo, bc_x, fill_x, pady_lo, bc_y, fill_y, m1, m2, m3, zb)
      switch (op) { case XG_OP_DIFF: XG_YS(XG_OP_DIFF); break; case XG_OP_INTERP: XG_YS(XG_OP_INTERP); break; case XG_OP_MIN: XG_YS(XG_OP_MIN); break; default: XG_YS(XG_OP_MAX); }
#undef XG_YS
      continue;
    }
#define XG_GM(O, NTS, M) hipLaunchKernelGGL((k_stencil2d<O, NTS, SEG, M>), dim3(grid), dim3(BLOCK), 0, st, in, out, o0, nouter, nblk, ny, nx, fnt, fns, order | (tune().nb_dpp ? 2 : 0), padx_lo, bc_x, fill_x, pady_lo, bc_y, fill_y, m1, m2, m3, zb)
#define XG_GZK(O, NTS, ZK_) hipLaunchKernelGGL((k_stencil2d<O, NTS, SEG, true, ZK_>), dim3(grid), dim3(BLOCK), 0, st, in, out, o0, nouter, nblk, ny, nx, fnt, fns, order | (tune().nb_dpp ? 2 : 0), padx_lo, bc_x, fill_x, pady_lo, bc_y, fill_y, m1, m2, m3, zb)
#define XG_GZ(O, NTS) do { if (ZK2 == 4) XG_GZK(O, NTS, 4); else XG_GZK(O, NTS, 2); } while (0)
#define XG_GO(O, NTS) do { if (met && zb.on) XG_GZ(O, NTS); else if (met) XG_GM(O, NTS, true); else XG_GM(O, NTS, false); } while (0)
#define XG_O(O) do { if (nts) XG_GO(O, true); else XG_GO(O, false); } while (0)
    switch (op) { case XG_OP_DIFF: XG_O(XG_OP_DIFF); break; case XG_OP_INTERP: XG_O(XG_OP_INTERP); break; case XG_OP_MIN: XG_O(XG_OP_MIN); break; default: XG_O(XG_OP_MAX); }
#undef XG_O
#undef XG_GO
#undef XG_GZ
#undef XG_GZK
#undef XG_GM
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}
#endif  // !XG_INT


extern "C" {

static int stencil1d_impl(int op, const real* in, const real* halo, real* out, const int64_t* shape, int ndim,
                          int axis, int64_t n_out, int pad_lo, int pad_hi, int bc, real fill, const real* m_in,
                          const int64_t* m_in_strides, const real* m_out, const int64_t* m_out_strides,
                          void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
#ifdef XG_INT
  if (op < XG_OP_DIFF || op > XG_OP_MAXU) return fail(XG_ERR_INVALID, "unknown op %d", op);
#else
  if (op < XG_OP_DIFF || op > XG_OP_MAX) return fail(XG_ERR_INVALID, "unknown op %d", op);
#endif
  if ((pad_lo | pad_hi) & ~1) return fail(XG_ERR_INVALID, "pad widths must be 0 or 1, got (%d,%d)", pad_lo, pad_hi);
  if (bc < XG_BC_NONE || bc > XG_BC_HALO) return fail(XG_ERR_INVALID, "unknown boundary mode %d", bc);
  if (bc == XG_BC_HALO && !halo) return fail(XG_ERR_INVALID, "XG_BC_HALO without a halo buffer");
  if ((m_in && !m_in_strides) || (m_out && !m_out_strides)) return fail(XG_ERR_INVALID, "metric without strides");
#ifdef XG_INT
  if (m_in || m_out) return fail(XG_ERR_UNSUPPORTED, "integer stencils take no metrics: convert to float64 first (numpy promotes int * float)");
#endif
  Geo g; MIdx mi, mo;
  int rc = build_geo(shape, ndim, axis, n_out, m_in ? m_in_strides : nullptr, m_out ? m_out_strides : nullptr, &g, &mi, &mo);
  if (rc) return rc;
  if (n_out != g.n_in + pad_lo + pad_hi - 1) return fail(XG_ERR_INVALID, "n_out %lld != n_in %lld + %d + %d - 1", (long long)n_out, (long long)g.n_in, pad_lo, pad_hi);
  if ((pad_lo || pad_hi) && bc == XG_BC_NONE) return fail(XG_ERR_INVALID, "halo cells requested but no boundary mode given");
  if (g.n_in < 1) return fail(XG_ERR_INVALID, "empty stencil axis");
  if (g.outer == 0 || g.inner == 0 || n_out <= 0) return XG_OK;  // empty output
  const int met = (m_out ? 1 : 0) | (m_in ? 2 : 0);
  const bool al = aligned16(in) && aligned16(out) && (bc != XG_BC_HALO || aligned16(halo));
  StencilCall c = {in, out, g, pad_lo, pad_hi, bc, fill, halo, m_in, mi, m_out, mo, (hipStream_t)stream};
  int V, kind;
  if (g.inner == 1) {
    kind = KIND_CONTIG;
    V = (al && (g.n_in % NV == 0) && (n_out % NV == 0)) ? NV : 1;
  } else {
    V = (al && (g.inner % NV == 0) && vec_metric_ok(g, met != 0)) ? NV : 1;
    // few x-tiles per row (Y of a (Z,Y,X) field): short banded segments keep the rows in flight
    // compact.  Many tiles per row (Z: a whole plane per row): the same kernel over column chunks
    // of `zchunk` tiles (measured 5.2 -> 6.0 TB/s against marching the full column, which is
    // kept for XG_ZCHUNK=0 and for extents beyond the u32 index range).
    const int64_t ntile = (g.inner + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V);
    const bool chunked = tune().zchunk > 0 && (met == 0 || g.idx32);
    kind = (ntile <= (int64_t)tune().seg_max_tiles || chunked) ? KIND_LIN : KIND_MARCH;
  }
  if (met != 0 && kind != KIND_MARCH && !g.idx32)
    return fail(XG_ERR_UNSUPPORTED, "metric-weighted stencils need outer/inner extents below 2^32");
  if (g.inner > 1 && V == 1 && al && tune().strided_gen && g.inner >= 2 * NV) {
    // rows of the strided axis are not 16-B aligned (odd inner extent): flat NV-group walk instead of 8-B lanes
    rc = strided_gen_dispatch(op, met, c);
    if (rc == 0) { XG_LAUNCH_CHECK(); return XG_OK; }
    if (rc != 1) return rc;  // 1: not applicable here, take the scalar-lane path
  }
  rc = stencil_dispatch(op, V, met, kind, c);
  if (rc) return rc;
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_stencil1d)(int op, const real* in, real* out, const int64_t* shape, int ndim, int axis,
                     int64_t n_out, int pad_lo, int pad_hi, int bc, real fill, const real* m_in,
                     const int64_t* m_in_strides, const real* m_out, const int64_t* m_out_strides,
                     void* stream) {
  if (bc == XG_BC_HALO) return fail(XG_ERR_INVALID, "XG_BC_HALO needs xg_stencil1d_halo");
  return stencil1d_impl(op, in, nullptr, out, shape, ndim, axis, n_out, pad_lo, pad_hi, bc, fill, m_in, m_in_strides,
                        m_out, m_out_strides, stream);
}

int XG_FN(xg_stencil1d_halo)(int op, const real* in, const real* halo, real* out, const int64_t* shape, int ndim,
                          int axis, int64_t n_out, int pad_lo, int pad_hi, const real* m_out,
                          const int64_t* m_out_strides, void* stream) {
  if (!halo && (pad_lo || pad_hi)) return fail(XG_ERR_INVALID, "NULL halo buffer");
  return stencil1d_impl(op, in, halo, out, shape, ndim, axis, n_out, pad_lo, pad_hi,
                        (pad_lo || pad_hi) ? XG_BC_HALO : XG_BC_NONE, real(0), nullptr, nullptr, m_out, m_out_strides,
                        stream);
}

#ifndef XG_INT
int XG_FN(xg_stencil1d_halo_w)(int op, const real* in, const real* halo, real* out, const int64_t* shape, int ndim,
                            int axis, int64_t n_out, int pad_lo, int pad_hi, const real* m_in,
                            const int64_t* m_in_strides, const real* m_out, const int64_t* m_out_strides,
                            void* stream) {
  if (!halo && (pad_lo || pad_hi)) return fail(XG_ERR_INVALID, "NULL halo buffer");
  return stencil1d_impl(op, in, halo, out, shape, ndim, axis, n_out, pad_lo, pad_hi,
                        (pad_lo || pad_hi) ? XG_BC_HALO : XG_BC_NONE, real(0), m_in, m_in_strides, m_out, m_out_strides,
                        stream);
}
#endif

#ifndef XG_INT
int XG_FN(xg_stencil2d)(int op, const real* in, real* out, const int64_t* shape, int ndim, int order,
                     int padx_lo, int padx_hi, int bc_x, real fill_x, int pady_lo, int pady_hi, int bc_y,
                     real fill_y, void* stream) {
  return stencil2d_impl(op, in, out, shape, ndim, order, padx_lo, padx_hi, bc_x, fill_x, pady_lo, pady_hi, bc_y, fill_y,
                        nullptr, nullptr, nullptr, stream);
}

int XG_FN(xg_stencil2d_metric)(int op, const real* in, real* out, const int64_t* shape, int ndim, int order,
                            int padx_lo, int padx_hi, int bc_x, real fill_x, int pady_lo, int pady_hi, int bc_y,
                            real fill_y, const real* m_in, const real* m_mid, const real* m_out, void* stream) {
  if (!m_in || !m_mid || !m_out) return fail(XG_ERR_INVALID, "NULL metric plane");
  return stencil2d_impl(op, in, out, shape, ndim, order, padx_lo, padx_hi, bc_x, fill_x, pady_lo, pady_hi, bc_y, fill_y,
                        m_in, m_mid, m_out, stream);
}
#endif  // !XG_INT

}  // extern "C"
