// xg_transform.hip -- vertical coordinate transform: linear / log interpolation and conservative remapping (K9a, K9b)
// Part of libxgcm_hip.so; compiled twice (real = double / -DXG_F32), see xg_common.hpp.

#include "xg_common.hpp"

namespace {

// (Rows leaving the LDS ring / window as 16-B stores of lane pairs instead of 8-B stores per lane, with and without `sc1 nt`,
// measured in one process on both lean kernels: -0.4 ... +0.1 points -- the 512-B row stores are not what bounds them.
// profiles/history/r03bf_ab_row_store.jsonl)
// ------------------------------------------------------------------------------------------
// vertical coordinate transform (next-row f4; reference xgcm/transform.py:15-142, numba gufuncs)
//
// One thread = one column (all other dims); lanes run along the innermost dim, so every step of
// the column loops is a coalesced access when the transform axis is not the contiguous one.
//
// K9a linear: numpy.interp(target, theta, phi) per column, restated step for step because the
//   result on NaN-laden or duplicated theta depends on the search path: the guess carried from one
//   target level to the next, the +-1 probes, the 8-element window, then bisection
//   (numpy/_core/src/multiarray/compiled_base.c; numba's np.interp is a port of the same code).
//   Arithmetic is double whatever the storage type, as in numpy/numba.
// K9b conservative: the reference's O(n*m) accumulation; per output bin the contributions are
//   added in source-level order (the order of the reference's outer loop), JT bins per pass held
//   in registers so that the column is re-read m/JT times instead of m times; a cell that misses
//   the whole (sorted) tile of bins is rejected with two compares.
// Measured dead end: staging each column in LDS ([level][lane], 64-lane blocks) made every probe
//   an LDS hit but left 2 waves per CU -- linear 12.8 -> 18.9 ms, conservative 18.5 -> 61.9 ms on
//   the 75 x 2400 x 3600 case; the kernels therefore run at full occupancy on global memory.
// ------------------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ int64_t interp_search(double key, F xp, int64_t len, int64_t guess) {
  int64_t imin = 0, imax = len;
  if (key > xp(len - 1)) return len;
  else if (key < xp(0)) return -1;
  if (len <= 4) {
    int64_t i;
    for (i = 1; i < len && key >= xp(i); ++i) {}
    return i - 1;
  }
  if (guess > len - 3) guess = len - 3;
  if (guess < 1) guess = 1;
  if (key < xp(guess)) {
    if (key < xp(guess - 1)) {
      imax = guess - 1;
      if (guess > 8 && key >= xp(guess - 8)) imin = guess - 8;
    } else {
      return guess - 1;
    }
  } else {
    if (key < xp(guess + 1)) return guess;
    if (key < xp(guess + 2)) return guess + 1;
    imin = guess + 2;
    if (guess < len - 8 - 1 && key < xp(guess + 8)) imax = guess + 8;
  }
  while (imin < imax) {
    const int64_t imid = imin + ((imax - imin) >> 1);
    if (key >= xp(imid)) imin = imid + 1;
    else imax = imid;
  }
  return imin - 1;
}

// np.log in the storage type: evaluated in double and rounded once (float32: correctly rounded
// up to double-rounding ties; numpy's own float32 log is a few-ulp SIMD routine, so method="log"
// is a tolerance comparison in float32, not a bit-exact one)
__device__ __forceinline__ real xg_log_store(real v) { return (real)log((double)v); }

// one interpolated value from the bracketing pair, numpy's formula and NaN fall-backs
__device__ __forceinline__ double interp_pair(double xv, double xj, double xj1, double fj, double fj1) {
  const double slope = (fj1 - fj) / (xj1 - xj);
  double res = slope * (xv - xj) + fj;
  if (res != res) {
    res = slope * (xv - xj1) + fj1;
    if (res != res && fj == fj1) res = fj;
  }
  return res;
}

// STAGE: one wave per workgroup keeps its 64 columns' outputs in an LDS tile [m][64] and writes them as m complete
// 512-B rows at the end.  The lanes of a wave reach a target level at different source levels, so emitting straight
// to memory stores a few lanes at a time: 239 M partial write requests for 54 M lines, 9.1 GB written for 3.5 GB
// (profiles/history/r02g_transform_linear_bounds.txt).  STAGE & 2: the target levels are the same for every column (a 1-D
// `target`): they sit in LDS too, so the divergent emission loop carries no memory instruction at all.
template <bool LOG, int STAGE, int TWIN = 16>
__global__ __launch_bounds__((STAGE & 5) ? WAVE : BLOCK) void k_transform_linear(
    const real* __restrict__ phi, const real* __restrict__ theta, const real* __restrict__ target,
    real* __restrict__ out, Geo g, MIdx mt, MIdx mg, int mask_edges, int bypass_checks, int fast_path) {
  extern __shared__ real t_lds[];
  constexpr int TB = (STAGE & 5) ? WAVE : BLOCK;
  const int64_t c = (int64_t)blockIdx.x * TB + threadIdx.x;
  const int64_t inner = g.inner, n = g.n_in, m = g.n_out;
  real* tile = t_lds;                                  // [m][64]
  real* lds_lev = t_lds + ((STAGE & 1) ? m * WAVE : (STAGE & 4) ? TWIN * WAVE : 0);  // [m]
  if (STAGE & 2) {  // before any lane leaves: the whole workgroup fills the level table
    for (int64_t i = threadIdx.x; i < m; i += TB) {
      const real v = target[i * mg.axis];
      lds_lev[i] = LOG ? xg_log_store(v) : v;
    }
    if (!(STAGE & 5)) __syncthreads();
  }
  // STAGE & 8 (with the ring and the shared level table): the LEAN streaming loop below.  The targets are the same for
  // every column, so they are validated ONCE here instead of at every emission (a NaN or decreasing target sends every
  // column to the exact path), and the table ends in +inf: "the cursor's level lies in this interval" is then one compare,
  // false for ever once the cursor has run off the end.
  bool levels_bad = false;
  if ((STAGE & 8) != 0) {
    bool bad = false;
    for (int64_t i = threadIdx.x; i < m; i += TB) {
      const real v = lds_lev[i];
      if (v != v || (i > 0 && v < lds_lev[i - 1])) bad = true;
    }
    levels_bad = __ballot(bad) != 0;
    if (threadIdx.x == 0) lds_lev[m] = (real)INFINITY;
  }
  if (c >= g.outer * g.inner) return;
  const int lane = threadIdx.x & 63;
  const int64_t o = c / inner, x = c - o * inner;
  const real* pphi = phi + (o * n) * inner + x;
  const real* pth = theta + outer_off(g, mt, o) + inner_off(g, mt, x);
  const real* ptg = target + outer_off(g, mg, o) + inner_off(g, mg, x);
  real* pout = out + (o * m) * inner + x;
  // theta as the reference sees it: the storage type's log first (np.log in interp_1d_linear)
  auto TH = [&](int64_t k) -> real { real v = pth[k * mt.axis]; return LOG ? xg_log_store(v) : v; };
  auto LEV = [&](int64_t i) -> real {
    if (STAGE & 2) return lds_lev[i];
    real v = ptg[i * mg.axis];
    return LOG ? xg_log_store(v) : v;
  };
  // STAGE & 4: the tile is a RING of TWIN rows.  `flushed` (wave-uniform) = rows already written out: a row leaves as
  // one complete 512-B store as soon as every lane of the wave has emitted it; a lane that runs more than TWIN rows
  // ahead of the slowest one stores directly and remembers the row in `direct` (m <= 64), as does the exact path.
  int64_t flushed = 0;
  u64 direct = 0;
  bool dir_all = false;  // exact path: every output stored directly
  auto OUT = [&](int64_t i, real r) {
    if (STAGE & 4) {
      if (!dir_all && i < flushed + TWIN) tile[(i & (TWIN - 1)) * WAVE + lane] = r;
      else { pout[i * inner] = r; direct |= 1ull << i; }
    } else if (STAGE & 1) tile[i * WAVE + lane] = r;
    else pout[i * inner] = r;
  };

  // ---- fast path: a well-formed column (no NaN, monotonic theta) and non-decreasing, NaN-free
  // targets.  numpy's search then returns max{j: xp[j] <= key} whatever its probing path, so the
  // column is streamed ONCE level by level (coalesced across lanes) while a per-lane cursor walks
  // the targets; any violation met on the way sends the lane to the exact path below, which
  // rewrites every output of the column.
  bool exact = !(fast_path & 1) || n < 2;
  if (!exact) {
    const real a0 = TH(0), a1 = TH(n - 1);
    if (a0 != a0 || a1 != a1) exact = true;
    else {
      const bool flip = !bypass_checks && (a1 < a0);
      const real tmin = flip ? a1 : a0, tmax = flip ? a0 : a1;  // == nanmin / nanmax once monotonic
      if (bypass_checks && a1 < a0) exact = true;               // decreasing but not flipped: numpy's path decides
      double xk = (double)(flip ? a1 : a0);
      double fk = (double)pphi[(flip ? n - 1 : 0) * inner];
      const double lval = fk;
      if constexpr ((STAGE & 8) != 0) {
        // ---- lean streaming loop: 32-bit cursors, column pointers advanced by a signed per-lane stride (no 64-bit
        // multiply per load), targets left of the column emitted before the loop, the division skipped where no lane of
        // the wave emits, and an emission body without branches except numpy's NaN fall-back.  Same operations on the same
        // operands as the loop below, so the same bits.
        if (levels_bad) exact = true;
        const int mm = (int)m, nn = (int)n;
        int i = 0, fl = 0;
        real lev = lds_lev[0];
        const int64_t sT = flip ? -mt.axis : mt.axis, sF = flip ? -inner : inner;
        const real* pT = pth + (flip ? n - 2 : 1) * mt.axis;  // the column's second level in walking order (n >= 2)
        const real* pF = pphi + (flip ? n - 2 : 1) * inner;
        const u32 inner32 = (u32)inner;
        real* pfl = pout;  // row `fl` of this lane's column: a running pointer, no multiply per flushed row
        // (occupancy hints, amdgpu_waves_per_eu 6 / 7 against the 5 waves its 84 VGPRs allow: +-0.3 / -2 points,
        // profiles/history/r03ad_ab_linear_occupancy.jsonl -- the loop is no longer bound by what it issues)
        // (edge masking needs no test inside the column: the cursor's level is >= the column's first point after the
        // prologue and < its current point in the loop; only targets left / right of the column can be masked)
        auto emit = [&](real r) {
          if (i < fl + TWIN) tile[(i & (TWIN - 1)) * WAVE + lane] = r;
          else { pout[(u64)((u32)i * inner32)] = r; direct |= 1ull << i; }  // (host: m * inner < 2^32 for this loop)
          ++i;
          lev = lds_lev[i];
        };
        if (!exact) {
          const real x0 = flip ? a1 : a0;
          const real left = mask_edges ? (real)NAN : (real)lval;
          while (lev < x0) emit(left);  // left of the column (x0 == tmin)
        }
        constexpr int UT = 8;
        for (int k0 = 1; k0 < nn && !exact; k0 += UT) {
          real tvs[UT], fvs[UT];
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            const real tv = *pT;
            tvs[u] = LOG ? xg_log_store(tv) : tv;
            fvs[u] = *pF;
            if (k0 + u + 1 < nn) { pT += sT; pF += sF; }  // (wave-uniform test; the last level is re-read by a short tail)
          }
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            if (k0 + u >= nn || exact) break;
            const real tv = tvs[u];
            const double xk1 = (double)tv, fk1 = (double)fvs[u];
            if (tv != tv || xk1 < xk) { exact = true; break; }
            if (__ballot(lev < tv) != 0) {  // some lane emits in this interval
              const double slope = (fk1 - fk) / (xk1 - xk);
              const bool flat = (fk == fk1);
              while (lev < tv) {
                const double xv = (double)lev;
                double res = slope * (xv - xk) + fk;
                if (__ballot(res != res) != 0) {  // numpy's NaN fall-backs (interp_pair): a wave-uniform branch, rarely taken
                  if (res != res) {
                    res = slope * (xv - xk1) + fk1;
                    if (res != res && flat) res = fk;
                  }
                }
                if (xv == xk) res = fk;
                emit((real)res);
              }
            }
            xk = xk1; fk = fk1;
            while (fl < mm && __ballot(i <= fl) == 0) {  // rows every streaming lane has emitted leave as complete stores
              if (!((direct >> fl) & 1)) stg<real, true>(pfl, tile[(fl & (TWIN - 1)) * WAVE + lane]);
              ++fl;
              pfl += inner;
            }
          }
        }
        while (i < mm && !exact) emit((mask_edges && lev > tmax) ? (real)NAN : (real)fk);  // at or right of the column's last point
        flushed = fl;
      } else {
      int64_t i = 0;
      // the cursor's current target level, loaded once per target and validated on load
      real lev = LEV(0);
      if (lev != lev) exact = true;
      auto emit_and_advance = [&](double res) {
        real r = (real)res;
        if (mask_edges && (lev < tmin || lev > tmax)) r = (real)NAN;
        OUT(i, r);
        ++i;
        if (i < m) {
          const real nxt = LEV(i);
          if (nxt != nxt || nxt < lev) exact = true;
          lev = nxt;
        }
      };
      constexpr int UT = (STAGE & 1) ? 16 : 8;  // levels fetched ahead of use: the column loads do not wait on each other
                                          // (the LDS tile leaves 6 waves per CU: twice the loads per wave)
      for (int64_t k0 = 1; k0 < n && !exact; k0 += UT) {
        real tvs[UT], fvs[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          const int64_t k = (k0 + u < n) ? k0 + u : n - 1;
          const int64_t kk = flip ? n - 1 - k : k;
          tvs[u] = TH(kk);
          fvs[u] = pphi[kk * inner];
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          if (k0 + u >= n || exact) break;
          const real tv = tvs[u];
          const double xk1 = (double)tv, fk1 = (double)fvs[u];
          if (tv != tv || xk1 < xk) { exact = true; break; }
          // numpy forms the slope per TARGET, but it depends on the interval only: computed here once per level
          // with every lane active, the IEEE division leaves the divergent emission loop below (the lanes of a
          // wave emit at different levels, so that loop's body runs for the union of the lanes: ~2.5 x per level
          // on random columns) -- same operands, same rounding, same bits
          const bool per_target = (fast_path & 2) != 0;  // experiment (tunable dbg & 4): the division inside the loop
          const double slope = per_target ? 0.0 : (fk1 - fk) / (xk1 - xk);
          while (i < m && !exact) {
            const double xv = (double)lev;
            if (!(xv < xk1)) break;          // belongs to a later interval (or to the right edge)
            double res;
            if (xv < xk) res = lval;         // only possible in the first interval: left of the column
            else if (xv == xk) res = fk;
            else if (per_target) res = interp_pair(xv, xk, xk1, fk, fk1);
            else {
              res = slope * (xv - xk) + fk;
              if (res != res) {              // numpy's NaN fall-backs (interp_pair)
                res = slope * (xv - xk1) + fk1;
                if (res != res && fk == fk1) res = fk;
              }
            }
            emit_and_advance(res);
          }
          xk = xk1; fk = fk1;
          if (STAGE & 4) {
            // every lane still on the streaming path is here (lanes sent to the exact path rewrite their column
            // themselves): rows that all of them have emitted leave as complete stores
            while (flushed < m && __ballot(i <= flushed) == 0) {
              if (!((direct >> flushed) & 1)) stg<real, true>(pout + flushed * inner, tile[(flushed & (TWIN - 1)) * WAVE + lane]);
              ++flushed;
            }
          }
        }
      }
      // remaining targets are >= xp[n-1]: the last point itself (fp[n-1]) or right of it (rval == fp[n-1])
      while (i < m && !exact) emit_and_advance(fk);
      }
    }
  }

  // ---- exact path: numpy's search, probe for probe
  if (exact) {
  dir_all = true;
  bool flip = false;
  real tmin = real(0), tmax = real(0);
  bool have = false;
  if (!bypass_checks || mask_edges) {
    real first = real(0), last = real(0);
    for (int64_t k = 0; k < n; ++k) {
      const real v = TH(k);
      if (v != v) continue;
      if (!have) { first = v; tmin = v; tmax = v; have = true; }
      last = v;
      tmin = (v < tmin) ? v : tmin;
      tmax = (v > tmax) ? v : tmax;
    }
    if (!bypass_checks && have) flip = last < first;
  }
  auto XP = [&](int64_t k) -> double { return (double)TH(flip ? n - 1 - k : k); };
  auto FP = [&](int64_t k) -> double { return (double)pphi[(flip ? n - 1 - k : k) * inner]; };

  const double lval = FP(0), rval = FP(n - 1);
  int64_t j = 0;
  for (int64_t i = 0; i < m; ++i) {
    const real lev = LEV(i);
    const double xv = (double)lev;
    double res;
    if (xv != xv) {
      res = xv;
    } else if (n == 1) {
      const double x0 = XP(0);
      res = (xv < x0) ? lval : ((xv > x0) ? rval : FP(0));
    } else {
      j = interp_search(xv, XP, n, j);
      if (j == -1) res = lval;
      else if (j == n) res = rval;
      else if (j == n - 1) res = FP(j);
      else {
        const double xj = XP(j);
        if (xj == xv) res = FP(j);
        else res = interp_pair(xv, xj, XP(j + 1), FP(j), FP(j + 1));
      }
    }
    real r = (real)res;
    if (mask_edges && have && (lev < tmin || lev > tmax)) r = (real)NAN;
    OUT(i, r);
  }
  }
  if (STAGE & 4) {  // what is still in the ring (a lane that took the exact path has nothing there)
    for (int64_t r = 0; r < m; ++r)
      if (!dir_all && r >= flushed && !((direct >> r) & 1)) stg<real, true>(pout + r * inner, tile[(r & (TWIN - 1)) * WAVE + lane]);
  } else if (STAGE & 1) {  // the tile leaves as complete rows (LDS accesses of one wave are ordered: no barrier)
    for (int64_t i = 0; i < m; ++i) stg<real, true>(pout + i * inner, tile[i * WAVE + lane]);
  }
}

template <int JT>
__global__ __launch_bounds__(BLOCK) void k_transform_conservative(
    const real* __restrict__ phi, const real* __restrict__ theta, const real* __restrict__ bins,
    real* __restrict__ out, Geo g, MIdx mt) {
  const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (c >= g.outer * g.inner) return;
  const int64_t inner = g.inner, n = g.n_in, m = g.n_out;
  const int64_t o = c / inner, x = c - o * inner;
  const real* pphi = phi + (o * n) * inner + x;
  const real* pth = theta + outer_off(g, mt, o) + inner_off(g, mt, x);
  real* pout = out + (o * m) * inner + x;
  for (int64_t j0 = 0; j0 < m; j0 += JT) {
    real acc[JT], b1[JT], b2[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) {
      const int64_t j = (j0 + t < m) ? j0 + t : m - 1;
      acc[t] = (real)NAN;
      b1[t] = bins[j];
      b2[t] = bins[j + 1];
    }
    real t1 = pth[0];
    constexpr int UT = 4;  // cells fetched ahead of use (the loads of a pass do not depend on each other)
    for (int64_t i0 = 0; i0 < n; i0 += UT) {
      real tts[UT], pps[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int64_t i = (i0 + u < n) ? i0 + u : n - 1;
        tts[u] = pth[(i + 1) * mt.axis];
        pps[u] = pphi[i * inner];
      }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        if (i0 + u >= n) break;
        const real t2 = tts[u], p = pps[u];
        const real a1 = t1;
        t1 = t2;
        const bool n1 = a1 != a1, n2 = t2 != t2;
        if (n1 && n2) continue;
        real lo_, hi_;
        if (n1) { lo_ = hi_ = t2; }
        else if (n2) { lo_ = hi_ = a1; }
        else if (a1 < t2) { lo_ = a1; hi_ = t2; }
        else { lo_ = t2; hi_ = a1; }
        if (p != p) continue;
        if (b1[0] > hi_ || b2[JT - 1] < lo_) continue;  // bins increase: the cell misses this whole tile
#pragma unroll
        for (int t = 0; t < JT; ++t) {
          if (b1[t] > hi_ || b2[t] < lo_) continue;
          real add;
          if (hi_ == lo_) {
            add = p;
          } else {
            const real hmin = (b1[t] > lo_) ? b1[t] : lo_;  // python max(theta_min, theta_hat_1)
            const real hmax = (b2[t] < hi_) ? b2[t] : hi_;  // python min(theta_max, theta_hat_2)
            const real alpha = (hmax - hmin) / (hi_ - lo_);
            add = alpha * p;
          }
          acc[t] = (acc[t] != acc[t]) ? add : acc[t] + add;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < JT; ++t)
      if (j0 + t < m) pout[(j0 + t) * inner] = acc[t];
  }
}

// K9c conservative, accumulators in LDS: cell-major order like the reference's loops.  Each lane
// owns m accumulator slots ([bin][lane] layout) and a cursor into the sorted bin edges, so a cell
// only visits the bins it overlaps (1-3 for stratified columns) instead of testing all m; per bin
// the additions still arrive in cell order => same bits as K9b / the reference.  Column data is
// streamed once (2n + 1 loads per column, coalesced).
constexpr int CTB = 128;
#ifndef XG_CONS_UT
#define XG_CONS_UT 16
#endif
extern __shared__ __align__(16) unsigned char xg_dyn_lds[];

__global__ __launch_bounds__(CTB) void k_transform_conservative_lds(
    const real* __restrict__ phi, const real* __restrict__ theta, const real* __restrict__ bins,
    real* __restrict__ out, Geo g, MIdx mt) {
  const int64_t inner = g.inner, n = g.n_in, m = g.n_out;
  real* sb = reinterpret_cast<real*>(xg_dyn_lds);          // m + 1 edges, padded to an even count
  real* acc = sb + ((m + 2) & ~(int64_t)1) + threadIdx.x;  // slot of bin j: acc[j * CTB]
  for (int64_t j = threadIdx.x; j <= m; j += CTB) sb[j] = bins[j];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * CTB + threadIdx.x;
  if (c >= g.outer * g.inner) return;
  const int64_t o = c / inner, x = c - o * inner;
  const real* pphi = phi + (o * n) * inner + x;
  const real* pth = theta + outer_off(g, mt, o) + inner_off(g, mt, x);
  real* pout = out + (o * m) * inner + x;
  for (int64_t j = 0; j < m; ++j) acc[j * CTB] = (real)NAN;
  // cursor bin jlo with its two edges and its accumulator held in registers: a stratified column
  // stays in the same bin for several cells, which then cost no LDS round trip at all
  int jlo = 0;
  const int mm = (int)m;
  real e_lo = sb[0], e_hi = sb[1], a_cur = (real)NAN;
  auto move_to = [&](int j) {
    acc[jlo * CTB] = a_cur;
    jlo = j;
    e_lo = sb[j]; e_hi = sb[j + 1];
    a_cur = acc[j * CTB];
  };
  real t1 = pth[0];
  constexpr int UT = XG_CONS_UT;
  for (int64_t i0 = 0; i0 < n; i0 += UT) {
    real tts[UT], pps[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int64_t i = (i0 + u < n) ? i0 + u : n - 1;
      tts[u] = pth[(i + 1) * mt.axis];
      pps[u] = pphi[i * inner];
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (i0 + u >= n) break;
      const real t2 = tts[u], p = pps[u];
      const real a1 = t1;
      t1 = t2;
      const bool n1 = a1 != a1, n2 = t2 != t2;
      if (n1 && n2) continue;
      real lo_, hi_;
      if (n1) { lo_ = hi_ = t2; }
      else if (n2) { lo_ = hi_ = a1; }
      else if (a1 < t2) { lo_ = a1; hi_ = t2; }
      else { lo_ = t2; hi_ = a1; }
      if (p != p) continue;
      // first bin whose upper edge reaches the cell: min{j: edge[j+1] >= lo}
      if ((jlo > 0 && e_lo >= lo_) || (e_hi < lo_ && jlo < mm - 1)) {
        int j = jlo;
        while (j > 0 && sb[j] >= lo_) --j;
        while (j < mm - 1 && sb[j + 1] < lo_) ++j;
        move_to(j);
      }
      if (e_hi < lo_ || e_lo > hi_) continue;  // the cell lies above the last bin / below this one: no overlap at all
      auto share = [&](real e1, real e2) -> real {
        if (hi_ == lo_) return p;
        const real hmin = (e1 > lo_) ? e1 : lo_;
        const real hmax = (e2 < hi_) ? e2 : hi_;
        const real alpha = (hmax - hmin) / (hi_ - lo_);
        return alpha * p;
      };
      {
        const real add = share(e_lo, e_hi);
        a_cur = (a_cur != a_cur) ? add : a_cur + add;
      }
      real e1 = e_hi;
      for (int j = jlo + 1; j < mm; ++j) {   // further bins the cell reaches into (through LDS)
        if (e1 > hi_) break;
        const real e2 = sb[j + 1];
        const real add = share(e1, e2);
        const real old = acc[j * CTB];
        acc[j * CTB] = (old != old) ? add : old + add;
        e1 = e2;
      }
    }
  }
  acc[jlo * CTB] = a_cur;
  for (int64_t j = 0; j < m; ++j) pout[j * inner] = acc[j * CTB];
}


// K9d conservative, a SLIDING WINDOW of accumulators in LDS.  K9c is bound by latency, not by bandwidth: m
// accumulators per lane (50 x 8 B x 128 lanes = 51 KB per workgroup) leave 6 waves on a CU, far too few to
// hide the dependent chain of a cell (edge look-ups, an IEEE division per overlapped bin, the LDS
// read-modify-write).  A column sweeps its bins in order -- exactly for monotonic theta, nearly for real
// stratification -- so each lane keeps only CWIN consecutive bins [wb, wb + CWIN) in LDS (slot = bin mod
// CWIN, 32 KB per 256 lanes => 16+ waves per CU); a bin the window leaves is written to `out`, a bin it
// (re-)enters is read back from `out` if it ever received a contribution (one bit per bin, m <= 64) and is
// NaN otherwise, a bin beyond the window (a cell thicker than CWIN bins) is read-modify-written in `out`
// directly.  Wherever an accumulator lives, its additions arrive in cell order: same bits as K9b / K9c /
// the reference.  Bins that never received anything get their NaN in a final pass.
constexpr int CWB = 256;

template <bool UNI, int CWIN>
__global__ __launch_bounds__(CWB) void k_transform_conservative_win(
    const real* __restrict__ phi, const real* __restrict__ theta, const real* __restrict__ bins,
    real* __restrict__ out, Geo g, MIdx mt) {
  const int64_t inner = g.inner, n = g.n_in;
  const int m = (int)g.n_out;  // <= 64 (host)
  real* sb = reinterpret_cast<real*>(xg_dyn_lds);       // m + 1 edges, padded to an even count
  real* ring = sb + ((m + 2) & ~1) + threadIdx.x;       // slot s of this lane: ring[s * CWB]
  for (int j = threadIdx.x; j <= m; j += CWB) sb[j] = bins[j];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * CWB + threadIdx.x;
  if (c >= g.outer * g.inner) return;
  const int64_t o = c / inner, x = c - o * inner;
  const real* pphi = phi + (o * n) * inner + x;
  const real* pth = theta + outer_off(g, mt, o) + inner_off(g, mt, x);
  real* pout = out + (o * (int64_t)m) * inner + x;
#pragma unroll
  for (int s_ = 0; s_ < CWIN; ++s_) ring[s_ * CWB] = (real)NAN;
  u64 touched = 0;  // bins that hold a value, in the window or already in `out`
  int wb = 0;       // the window covers bins [wb, wb + CWIN); UNI: the same for every lane of the wave (see below)
  auto was_touched = [&](int j) -> bool { return (touched >> j) & 1ull; };
  // a value this lane stored earlier: read past the L1 (the store went through to L2)
  auto reload = [&](int j) -> real { return __hip_atomic_load(pout + (int64_t)j * inner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto slide_to = [&](int j0) {
    while (wb < j0) {  // forward: bin wb leaves, bin wb + CWIN enters the slot it frees
      if (was_touched(wb)) pout[(int64_t)wb * inner] = ring[(wb & (CWIN - 1)) * CWB];
      const int in = wb + CWIN;
      if (in < m) ring[(in & (CWIN - 1)) * CWB] = was_touched(in) ? reload(in) : (real)NAN;
      ++wb;
    }
    while (wb > j0) {  // backward (non-monotonic columns): bin wb - 1 enters, bin wb - 1 + CWIN leaves
      --wb;
      const int outb = wb + CWIN;
      if (outb < m && was_touched(outb)) pout[(int64_t)outb * inner] = ring[(outb & (CWIN - 1)) * CWB];
      ring[(wb & (CWIN - 1)) * CWB] = was_touched(wb) ? reload(wb) : (real)NAN;
    }
  };
  auto accumulate = [&](int j, real add) {
    if (j < wb + CWIN && (!UNI || j >= wb)) {
      real* slot = ring + (j & (CWIN - 1)) * CWB;
      const real old = *slot;
      *slot = (old != old) ? add : old + add;
    } else {  // the cell reaches beyond the window
      const real old = was_touched(j) ? reload(j) : (real)NAN;
      pout[(int64_t)j * inner] = (old != old) ? add : old + add;
    }
    touched |= 1ull << j;
  };
  int jlo = 0;  // cursor: first bin whose upper edge reaches the current cell
  // UNI: ONE window per wave.  A lane sliding its own window writes the bin it leaves on its own, a few lanes at a
  // time (partial lines, as in the linear transform: profiles/history/r02g_transform_linear_bounds.txt).  Here bin wb
  // leaves when the cursor of EVERY lane has passed it, as one complete row (NaN where a column never touched it);
  // a lane whose cursor falls behind the window (non-monotonic column) or whose cell reaches beyond it accumulates
  // in `out` directly, as before -- wherever an accumulator lives its additions arrive in cell order.
  // A lane whose current cell is void (NaN bounds or NaN value: land, missing data) does not hold the window back; if
  // it comes back to life behind the window it accumulates in `out` directly.
  bool idle = false;
  auto advance_window = [&]() {  // called where all lanes of the wave are present
    while (wb < m && __ballot(!idle && jlo <= wb) == 0) {
      real* slot = ring + (wb & (CWIN - 1)) * CWB;
      stg<real, true>(pout + (int64_t)wb * inner, *slot);
      const int in = wb + CWIN;
      if (in < m) *slot = was_touched(in) ? reload(in) : (real)NAN;
      ++wb;
    }
  };
  real e_lo = sb[0], e_hi = sb[1];
  real t1 = pth[0];
  constexpr int UT = 8;
  for (int64_t i0 = 0; i0 < n; i0 += UT) {
    real tts[UT], pps[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      const int64_t i = (i0 + u < n) ? i0 + u : n - 1;
      tts[u] = pth[(i + 1) * mt.axis];
      pps[u] = pphi[i * inner];
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (i0 + u >= n) break;
      if (UNI) advance_window();
      const real t2 = tts[u], p = pps[u];
      const real a1 = t1;
      t1 = t2;
      const bool n1 = a1 != a1, n2 = t2 != t2;
      idle = (n1 && n2) || (p != p);
      if (n1 && n2) continue;
      real lo_, hi_;
      if (n1) { lo_ = hi_ = t2; }
      else if (n2) { lo_ = hi_ = a1; }
      else if (a1 < t2) { lo_ = a1; hi_ = t2; }
      else { lo_ = t2; hi_ = a1; }
      if (p != p) continue;
      // first bin whose upper edge reaches the cell: min{j: edge[j+1] >= lo}
      if ((jlo > 0 && e_lo >= lo_) || (e_hi < lo_ && jlo < m - 1)) {
        int j = jlo;
        while (j > 0 && sb[j] >= lo_) --j;
        while (j < m - 1 && sb[j + 1] < lo_) ++j;
        jlo = j;
        e_lo = sb[j];
        e_hi = sb[j + 1];
      }
      if (e_hi < lo_ || e_lo > hi_) continue;  // the cell lies above the last bin / below this one: no overlap at all
      if (!UNI && jlo != wb) slide_to(jlo);
      real e1 = e_lo, e2 = e_hi;
      for (int j = jlo;;) {
        real add;
        if (hi_ == lo_) add = p;
        else {
          const real hmin = (e1 > lo_) ? e1 : lo_;  // python max(theta_min, theta_hat_1)
          const real hmax = (e2 < hi_) ? e2 : hi_;  // python min(theta_max, theta_hat_2)
          const real alpha = (hmax - hmin) / (hi_ - lo_);
          add = alpha * p;
        }
        accumulate(j, add);
        if (++j >= m) break;
        e1 = e2;
        if (e1 > hi_) break;  // bin j starts above the cell
        e2 = sb[j + 1];
      }
    }
  }
  if (UNI) {  // what is left of the window leaves as complete rows too; only bins beyond it need their NaN
    for (int s_ = 0; s_ < CWIN; ++s_) {
      const int b = wb + s_;
      if (b < m) stg<real, true>(pout + (int64_t)b * inner, ring[(b & (CWIN - 1)) * CWB]);
    }
    for (int j = wb + CWIN; j < m; ++j)
      if (!was_touched(j)) pout[(int64_t)j * inner] = (real)NAN;
    return;
  }
  for (int s_ = 0; s_ < CWIN; ++s_) {
    const int b = wb + s_;
    if (b < m && was_touched(b)) pout[(int64_t)b * inner] = ring[(b & (CWIN - 1)) * CWB];
  }
  for (int j = 0; j < m; ++j)
    if (!was_touched(j)) pout[(int64_t)j * inner] = (real)NAN;
}

// K9e: K9d's one-window-per-wave form (UNI) written for instruction count -- the kernel issues VALU instructions 66 % of
// its time (profiles/history/r03z_valu_issue_share.txt), so what it executes per cell is what it costs:
//   * the cursor walks forward with its two edges in registers (one LDS read per step); the backward search is a rare
//     separate branch (non-monotonic columns);
//   * a bin that has left the window was stored as a complete row (NaN where a column never touched it), so an accumulation
//     behind the window always reads `out` back: the per-bin "touched" bits are kept for accumulations AHEAD of the window
//     only (cells thicker than the window), not updated in the common path;
//   * the share of a bin is one expression (a cell of zero thickness selects its value after the division instead of
//     branching around it); 32-bit bin arithmetic throughout.
// Per bin the additions still arrive in cell order with the same operands: the same bits as K9b / K9c / K9d / the reference.
template <int CWIN>
__global__ __launch_bounds__(CWB) void k_transform_conservative_uni(
    const real* __restrict__ phi, const real* __restrict__ theta, const real* __restrict__ bins,
    real* __restrict__ out, Geo g, MIdx mt) {
  const int64_t inner = g.inner;
  const int n = (int)g.n_in, m = (int)g.n_out;  // m <= 64 (host)
  real* sb = reinterpret_cast<real*>(xg_dyn_lds);       // m + 1 edges, padded to an even count
  real* ring = sb + ((m + 2) & ~1) + threadIdx.x;       // slot s of this lane: ring[s * CWB]
  for (int j = threadIdx.x; j <= m; j += CWB) sb[j] = bins[j];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * CWB + threadIdx.x;
  if (c >= g.outer * g.inner) return;
  const int64_t o = c / inner, x = c - o * inner;
  const real* pphi = phi + (o * n) * inner + x;
  const real* pth = theta + outer_off(g, mt, o) + inner_off(g, mt, x);
  real* pout = out + (o * (int64_t)m) * inner + x;
#pragma unroll
  for (int s_ = 0; s_ < CWIN; ++s_) ring[s_ * CWB] = (real)NAN;
  u64 ahead = 0;  // bins beyond the window that received a contribution in `out` directly
  int wb = 0;     // the wave's window covers bins [wb, wb + CWIN)
  // addresses: the window's first row as a running pointer; a row outside the window through a 32-bit element offset
  // (host: m * inner < 2^32) formed where it is needed -- no 64-bit multiply in the per-cell path
  real* pwb = pout;
  const u32 inner32 = (u32)inner;
  auto reload_at = [&](const real* q) -> real { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto direct = [&](int j, real add) {  // outside the window: read-modify-write in `out`
    real* q = pout + (u64)((u32)j * inner32);
    real old;
    if (j < wb) old = reload_at(q);  // the row left the window: it holds a value or NaN
    else {
      old = ((ahead >> j) & 1ull) ? reload_at(q) : (real)NAN;
      ahead |= 1ull << j;
    }
    *q = (old != old) ? add : old + add;
  };
  int jlo = 0;  // cursor: first bin whose upper edge reaches the current cell
  bool idle = false;
  real e_lo = sb[0], e_hi = sb[1];
  real t1 = pth[0];
  const real* pT = pth + mt.axis;
  const real* pF = pphi;
  constexpr int UT = 8;
  for (int i0 = 0; i0 < n; i0 += UT) {
    real tts[UT], pps[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      tts[u] = *pT;
      pps[u] = *pF;
      if (i0 + u + 1 < n) { pT += mt.axis; pF += inner; }
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      if (i0 + u >= n) break;
      while (wb < m && __ballot(!idle && jlo <= wb) == 0) {  // bin wb: every live lane's cursor has passed it
        real* slot = ring + (wb & (CWIN - 1)) * CWB;
        stg<real, true>(pwb, *slot);
        const int in = wb + CWIN;
        if (in < m) *slot = ((ahead >> in) & 1ull) ? reload_at(pwb + (int64_t)CWIN * inner) : (real)NAN;
        ++wb;
        pwb += inner;
      }
      const real t2 = tts[u], p = pps[u];
      const real a1 = t1;
      t1 = t2;
      const bool n1 = a1 != a1, n2 = t2 != t2;
      idle = (n1 && n2) || (p != p);
      if (idle) continue;
      real lo_ = (a1 < t2) ? a1 : t2, hi_ = (a1 < t2) ? t2 : a1;  // (a NaN bound: both are the other one)
      if (n1) { lo_ = t2; hi_ = t2; }
      if (n2) { lo_ = a1; hi_ = a1; }
      // first bin whose upper edge reaches the cell: min{j: edge[j+1] >= lo}
      if (jlo > 0 && e_lo >= lo_) {  // the column turned back
        int j = jlo;
        while (j > 0 && sb[j] >= lo_) --j;
        jlo = j;
        e_lo = sb[j];
        e_hi = sb[j + 1];
      }
      while (e_hi < lo_ && jlo < m - 1) {
        ++jlo;
        e_lo = e_hi;
        e_hi = sb[jlo + 1];
      }
      if (e_hi < lo_ || e_lo > hi_) continue;  // the cell lies above the last bin / below this one: no overlap at all
      const real width = hi_ - lo_;
      const bool thin = (hi_ == lo_);
      real e1 = e_lo, e2 = e_hi;
      for (int j = jlo;;) {
        const real hmin = (e1 > lo_) ? e1 : lo_;  // python max(theta_min, theta_hat_1)
        const real hmax = (e2 < hi_) ? e2 : hi_;  // python min(theta_max, theta_hat_2)
        real add = ((hmax - hmin) / width) * p;
        if (thin) add = p;
        if (j >= wb && j < wb + CWIN) {
          real* slot = ring + (j & (CWIN - 1)) * CWB;
          const real old = *slot;
          *slot = (old != old) ? add : old + add;
        } else {
          direct(j, add);
        }
        if (++j >= m) break;
        e1 = e2;
        if (e1 > hi_) break;  // bin j starts above the cell
        e2 = sb[j + 1];
      }
    }
  }
  // what is left of the window leaves as complete rows too; only bins beyond it need their NaN
  for (int s_ = 0; s_ < CWIN; ++s_) {
    const int b = wb + s_;
    if (b < m) stg<real, true>(pwb + (int64_t)s_ * inner, ring[(b & (CWIN - 1)) * CWB]);
  }
  for (int j = wb + CWIN; j < m; ++j)
    if (!((ahead >> j) & 1ull)) pout[(int64_t)j * inner] = (real)NAN;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int XG_FN(xg_transform_linear)(const real* phi, const real* theta, const int64_t* theta_strides, const real* target,
                            const int64_t* target_strides, int64_t m, real* out, const int64_t* shape, int ndim,
                            int axis, int mask_edges, int bypass_checks, int logarithmic, void* stream) {
  if (!phi || !theta || !target || !out || !shape || !theta_strides || !target_strides)
    return fail(XG_ERR_INVALID, "NULL argument");
  if (m < 0) return fail(XG_ERR_INVALID, "negative number of target levels");
  Geo g; MIdx mt, mg;
  int rc = build_geo(shape, ndim, axis, m, theta_strides, target_strides, &g, &mt, &mg);
  if (rc) return rc;
  if (g.n_in < 1) return fail(XG_ERR_INVALID, "empty transform axis");
  const int64_t cols = g.outer * g.inner;
  if (cols == 0 || m == 0) return XG_OK;
  hipStream_t st = (hipStream_t)stream;
  const int fast = (tune().transform_fast ? 1 : 0) | ((tune().dbg & 4) ? 2 : 0);
  // outputs staged in LDS (one wave per workgroup) while the tile fits 48 KB: up to 96 (f64) / 192 (f32) target levels
  bool shared_levels = true;
  for (int d = 0; d < MAXD; ++d)
    if ((d < g.n_outer && mg.outer[d] != 0) || (d < g.n_inner && mg.inner[d] != 0)) shared_levels = false;
  const size_t lds = ((size_t)m * WAVE + (shared_levels ? (size_t)m : 0)) * sizeof(real);
  if (tune().transform_stage == 2 && shared_levels && m <= 4096) {  // target levels in LDS only
    const u64 nblocks = ((u64)cols + BLOCK - 1) / BLOCK;
    if ((rc = check_grid(nblocks))) return rc;
    if (logarithmic) hipLaunchKernelGGL((k_transform_linear<true, 2>), dim3((u32)nblocks), dim3(BLOCK), (size_t)m * sizeof(real), st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast);
    else hipLaunchKernelGGL((k_transform_linear<false, 2>), dim3((u32)nblocks), dim3(BLOCK), (size_t)m * sizeof(real), st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast);
    XG_LAUNCH_CHECK();
    return XG_OK;
  }
  if (tune().transform_stage >= 3 && m <= 64) {  // ring of 16 output rows per wave (+ the level table)
    const u64 nblocks = ((u64)cols + WAVE - 1) / WAVE;
    if ((rc = check_grid(nblocks))) return rc;
    const int tw = tune().transform_ring <= 4 ? 4 : tune().transform_ring <= 8 ? 8 : tune().transform_ring >= 32 ? 32 : 16;
    const size_t rl = ((size_t)tw * WAVE + (shared_levels ? (size_t)m + 1 : 0)) * sizeof(real);  // (+1: the lean loop's sentinel)
    const bool lean = shared_levels && (tune().transform_lean & 1) && (u64)m * (u64)g.inner < (1ull << 32);
#define XG_T(L, S, TW) hipLaunchKernelGGL((k_transform_linear<L, S, TW>), dim3((u32)nblocks), dim3(WAVE), rl, st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast)
#define XG_TS(L, TW) do { if (lean) XG_T(L, 14, TW); else if (shared_levels) XG_T(L, 6, TW); else XG_T(L, 4, TW); } while (0)
#define XG_TW(L) do { if (tw == 4) XG_TS(L, 4); else if (tw == 8) XG_TS(L, 8); else if (tw == 32) XG_TS(L, 32); else XG_TS(L, 16); } while (0)
    if (logarithmic) XG_TW(true); else XG_TW(false);
#undef XG_TW
#undef XG_TS
#undef XG_T
    XG_LAUNCH_CHECK();
    return XG_OK;
  }
  if (tune().transform_stage == 1 && lds <= 48 * 1024) {
    const u64 nblocks = ((u64)cols + WAVE - 1) / WAVE;
    if ((rc = check_grid(nblocks))) return rc;
#define XG_T(L, S) hipLaunchKernelGGL((k_transform_linear<L, S>), dim3((u32)nblocks), dim3(WAVE), lds, st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast)
    if (logarithmic) { if (shared_levels) XG_T(true, 3); else XG_T(true, 1); }
    else { if (shared_levels) XG_T(false, 3); else XG_T(false, 1); }
#undef XG_T
    XG_LAUNCH_CHECK();
    return XG_OK;
  }
  const u64 nblocks = ((u64)cols + BLOCK - 1) / BLOCK;
  if ((rc = check_grid(nblocks))) return rc;
  if (logarithmic) hipLaunchKernelGGL((k_transform_linear<true, 0>), dim3((u32)nblocks), dim3(BLOCK), 0, st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast);
  else hipLaunchKernelGGL((k_transform_linear<false, 0>), dim3((u32)nblocks), dim3(BLOCK), 0, st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast);
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_transform_conservative)(const real* phi, const real* theta, const int64_t* theta_strides, const real* bins,
                                  int64_t n_edges, real* out, const int64_t* shape, int ndim, int axis,
                                  void* stream) {
  if (!phi || !theta || !bins || !out || !shape || !theta_strides) return fail(XG_ERR_INVALID, "NULL argument");
  if (n_edges < 2) return fail(XG_ERR_INVALID, "need at least two bin edges");
  Geo g; MIdx mt;
  int rc = build_geo(shape, ndim, axis, n_edges - 1, theta_strides, nullptr, &g, &mt, nullptr);
  if (rc) return rc;
  const int64_t cols = g.outer * g.inner;
  if (cols == 0) return XG_OK;
  const int64_t m = n_edges - 1;
  if (tune().transform_win && m <= 64) {  // sliding accumulator window (K9d)
    const int cw = (tune().transform_win == 2 || tune().transform_cwin > 8) ? 16 : tune().transform_cwin <= 4 ? 4 : 8;
    const size_t wlds = ((size_t)((m + 2) & ~(int64_t)1) + (size_t)cw * CWB) * sizeof(real);
    const u64 nblocks = ((u64)cols + CWB - 1) / CWB;
    if ((rc = check_grid(nblocks))) return rc;
    const bool lean = (tune().transform_lean & 2) != 0 && tune().transform_win != 2 && (u64)m * (u64)g.inner < (1ull << 32);
    if (lean && cw == 16) hipLaunchKernelGGL((k_transform_conservative_uni<16>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    else if (lean && cw == 4) hipLaunchKernelGGL((k_transform_conservative_uni<4>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    else if (lean) hipLaunchKernelGGL((k_transform_conservative_uni<8>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    else if (tune().transform_win == 2) hipLaunchKernelGGL((k_transform_conservative_win<false, 16>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    else if (cw == 16) hipLaunchKernelGGL((k_transform_conservative_win<true, 16>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    else if (cw == 4) hipLaunchKernelGGL((k_transform_conservative_win<true, 4>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    else hipLaunchKernelGGL((k_transform_conservative_win<true, 8>), dim3((u32)nblocks), dim3(CWB), wlds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
    XG_LAUNCH_CHECK();
    return XG_OK;
  }
  const size_t lds = ((size_t)((m + 2) & ~(int64_t)1) + (size_t)m * CTB) * sizeof(real);
  if (tune().transform_lds_kb > 0 && lds <= (size_t)tune().transform_lds_kb * 1024u) {
    const u64 nblocks = ((u64)cols + CTB - 1) / CTB;
    if ((rc = check_grid(nblocks))) return rc;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)k_transform_conservative_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(XG_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_transform_conservative_lds, dim3((u32)nblocks), dim3(CTB), lds, (hipStream_t)stream, phi, theta, bins, out, g, mt);
  } else {
    const u64 nblocks = ((u64)cols + BLOCK - 1) / BLOCK;
    if ((rc = check_grid(nblocks))) return rc;
    hipLaunchKernelGGL((k_transform_conservative<8>), dim3((u32)nblocks), dim3(BLOCK), 0, (hipStream_t)stream, phi, theta, bins, out, g, mt);
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

}  // extern "C"
