// xg_vector.hip -- fused two-component operators (vorticity K7, divergence K7b, gradient / flux K7c) and the broadcasting binary op
// Part of libxgcm_hip.so; compiled twice (real = double / -DXG_F32), see xg_common.hpp.

#include "xg_common.hpp"

// rows of Y per wave-task of the fused two-component kernels (see STENCIL_SEG in xg_stencil.hip): A/B on one GPU
// (XG_HIP_LIB), 4320 x 4320 x 90: vorticity 70.0 / 71.7 / 71.3 % with 4 / 2 / 1 rows, gradient 73.9 / 75.7 / 75.6 %,
// flux 71.4 / 74.3 / 74.5 %; the two-axis kernel K8 78.3 / 78.6 / 75.1 %  =>  2
#ifndef XG_FUSED_SEG
#define XG_FUSED_SEG 2
#endif

namespace {

// ------------------------------------------------------------------------------------------
// broadcasting binary op (out C-contiguous, a/b addressed through strides; dims pre-coalesced)
// ------------------------------------------------------------------------------------------
struct BinGeo {
  int ndim;
  int64_t total;  // number of V-wide items
  int64_t shape[XG_MAX_NDIM];  // shape[ndim-1] counts V-wide items
  int64_t sa[XG_MAX_NDIM], sb[XG_MAX_NDIM];
  int idx32;                 // total < 2^32: the item index is peeled with multiply-shift divisions (FastDiv)
  FastDiv fs[XG_MAX_NDIM];   // divisors shape[d]
};

template <int BOP> __device__ __forceinline__ real bin2(real a, real b) {
  if (BOP == XG_BIN_MUL) return a * b;
#ifndef XG_INT
  if (BOP == XG_BIN_DIV) return a / b;
#endif
  if (BOP == XG_BIN_ADD) return a + b;
  return a - b;
}

// `nta` / `ntb`: the operand is streamed exactly once (no broadcast dim) => non-temporal loads; together with the
// XCD-banded block order that is worth 4 points on a three-stream kernel (tools/probes/streambench.hip SB_TRIAD: 75.4 %
// plain, 77.8 % with non-temporal loads, 79.1 % banded as well)
template <int BOP, int V, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_binary(const real* __restrict__ a, const real* __restrict__ b,
                                                  real* __restrict__ out, BinGeo g, ZBand zb, u32 nblk, int nta, int ntb) {
  typedef typename VecT<V>::type T;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  int64_t gid = (int64_t)lb * BLOCK + threadIdx.x;
  if (zb.on) {
    // (Z, P) items with one operand broadcast along Z (da / dx(Y,X)): band-major order keeps the
    // band of the small operand in L2 while all Z levels of the band stream by (as in K1 / K2S)
    u32 z, pin;
    if (gid >= (int64_t)zb.per_band.d * ((zb.Y + zb.B - 1) / zb.B)) return;
    if (!zband_map(zb, (u32)gid, z, pin)) return;
    gid = (int64_t)z * zb.Y + pin;
  }
  if (gid >= g.total) return;
  int64_t oa = 0, ob = 0;
  if (g.idx32) {  // (a 64-bit division per dim and thread cost this kernel 5 points: 0.74 -> 0.79 for da / dx(Y,X))
    u32 r = (u32)gid;
#pragma unroll
    for (int d = XG_MAX_NDIM - 1; d >= 0; --d) {
      if (d < g.ndim) {
        const u32 q = (d == 0) ? 0u : fdiv(r, g.fs[d]);  // (the slowest dim needs no division: r < shape[0] there)
        int64_t c = (int64_t)(r - q * g.fs[d].d);
        if (d == g.ndim - 1) c *= V;
        oa += c * g.sa[d];
        ob += c * g.sb[d];
        r = q;
      }
    }
  } else {
    int64_t r = gid;
#pragma unroll
    for (int d = XG_MAX_NDIM - 1; d >= 0; --d) {
      if (d < g.ndim) {
        int64_t s = g.shape[d];
        int64_t q = r / s;
        int64_t c = r - q * s;
        if (d == g.ndim - 1) c *= V;
        oa += c * g.sa[d];
        ob += c * g.sb[d];
        r = q;
      }
    }
  }
  const int64_t sa_in = g.sa[g.ndim - 1], sb_in = g.sb[g.ndim - 1];
  if (V > 1) {
    dv av, bv, o;
    if (sa_in == 1) av = nta ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(a + oa)) : *reinterpret_cast<const dv*>(a + oa);
    else {
#pragma unroll
      for (int k = 0; k < NV; ++k) av[k] = a[oa + k * sa_in];
    }
    if (sb_in == 1) bv = ntb ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(b + ob)) : *reinterpret_cast<const dv*>(b + ob);
    else {
#pragma unroll
      for (int k = 0; k < NV; ++k) bv[k] = b[ob + k * sb_in];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) o[k] = bin2<BOP>(av[k], bv[k]);
    stg_s<dv, NTS>(out + gid * NV, o);  // (`sc1 nt`, DESIGN rule 16: da / dx +2.8, a * b +2.2 points on one box, +-0 on another)
  } else {
    stg<real, NTS>(out + gid, bin2<BOP>(a[oa], b[ob]));
  }
}

#ifndef XG_INT  // the fused two-component operators divide by / multiply with float metrics: float builds only
// ------------------------------------------------------------------------------------------
// K7: fused relative vorticity ((v[j,i]-v[j,i-1]) - (u[j,i]-u[j-1,i])) / area, view (outer,Y,X).
// Same shape as K2S: lanes along X (V=2 when nx even), XCD-banded waves, each wave register-marches
// SEG rows of Y: SEG+1 rows of u (the j-1 halo row is an L2 hit), SEG rows of v plus the 8-byte
// left neighbour (same cache lines), SEG rows of area.  24 B/cell instead of 56 B unfused.
// ------------------------------------------------------------------------------------------
// offset of the (Y, X) plane of `area` that belongs to outer index o: the leading dims of the field
// that the area does not have broadcast (stride 0), the others advance it (e.g. a field (Z, face, j, i)
// with rAz(face, j, i)); dims are peeled innermost-first with multiply-shift division on the scalar unit
struct AreaIdx {
  int n;
  FastDiv fd[XG_MAX_NDIM];
  int64_t stride[XG_MAX_NDIM];
};
__device__ __forceinline__ int64_t area_outer_off(const AreaIdx& ai, int64_t o) {
  int64_t off = 0;
  u32 rem = (u32)o;
#pragma unroll
  for (int d = XG_MAX_NDIM - 1; d >= 0; --d) {
    if (d < ai.n) {
      const u32 q = fdiv(rem, ai.fd[d]);
      off += (int64_t)(rem - q * ai.fd[d].d) * ai.stride[d];
      rem = q;
    }
  }
  return off;
}

// ZK > 1 (z-banded launches only): the wave carries the same rows of ZK consecutive levels and loads the area
// rows once for all of them -- L2-resident metric rows still compete with the field loads for the CU's
// outstanding requests (measured on the 1-D kernels: derivative along X 74.8 -> 77.3 % with shared rows).
// SEG rows of a metric / area plane for a V-wide lane, row r at m[off + r * sy] (short tails repeat the last row), elements
// `sx` apart.  The FORM of the load -- one aligned vector, or element by element -- is the HOST's decision (`vec`: unit step,
// base, row step and every outer stride keep the vector's alignment -- `plane_vec_ok`), one scalar branch per plane: `ldm`'s
// own test is a lane-divergent branch per load, a row loop with such loads in it issues them one by one, each after the wait
// for the one before (the gradient: four round trips to the L2 in a row after the field had arrived), and a test made in the
// kernel joins its paths in front of the loads, where the wave then waits for everything in flight.
template <typename T, int SEG>
__device__ __forceinline__ void load_rows(T (&dst)[SEG], const real* __restrict__ m, int64_t off, int64_t sy, int64_t sx, int64_t nrow, bool vec) {
  if (sizeof(T) > sizeof(real) && vec) {
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) dst[s_] = *reinterpret_cast<const T*>(m + off + ((s_ < nrow) ? s_ : nrow - 1) * sy);
  } else {
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) dst[s_] = ldm<T>(m, off + ((s_ < nrow) ? s_ : nrow - 1) * sy, sx);
  }
}

__device__ __forceinline__ real vec_last(dv v) { return v[NV - 1]; }
__device__ __forceinline__ real vec_last(real v) { return v; }

template <int V, bool HAS_AREA, bool NTS, int SEG, int ZK = 1>
__global__ __launch_bounds__(BLOCK) void k_vorticity(
    const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ area,
    real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny, int64_t nx, FastDiv ntile,
    FastDiv nseg, ZBand zb, int bc_x, real fill_x, int bc_y, real fill_y, AreaIdx ai, int64_t a_sy,
    int64_t a_sx, const real* __restrict__ halo_x, const real* __restrict__ halo_y, int ntl) {
  typedef typename VecT<V>::type T;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  const u32 tile = w - r * ntile.d;
  u32 oo, sg;
  if (HAS_AREA && zb.on) {  // band-major: a (Y,X) area band stays in the XCD's L2 for all levels
    if (!zband_map(zb, r, oo, sg)) return;
    oo *= ZK;
  } else {
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    sg = r - oo * nseg.d;
  }
  const int nk = (ZK > 1 && (int64_t)nouter - (int64_t)oo < ZK) ? (int)(nouter - oo) : ZK;
  const int64_t o = o0 + oo;
  const int64_t a_base = HAS_AREA ? area_outer_off(ai, o) : 0;
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  const bool edge = (i0 == 0);
  const int64_t nidx = edge ? ((bc_x == XG_BC_PERIODIC) ? nx - 1 : 0) : i0 - 1;
  const bool fill_edge = edge && (bc_x == XG_BC_FILL);

  T uu[ZK][SEG + 1], vv[ZK][SEG], ar[SEG];
  real vl[ZK][SEG];
  bool f0 = false;
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    const int64_t ok = o + ((kz < nk) ? kz : nk - 1);  // a short last group repeats its last level (not stored)
    const real* pu = u + ok * ny * nx + i0;
    const real* pv = v + (ok * ny + j0) * nx;
    {
      int64_t q = j0 - 1;
      const real* src = pu + q * nx;
      if (q < 0) {
        f0 = (bc_y == XG_BC_FILL);
        src = pu + ((bc_y == XG_BC_PERIODIC) ? ny - 1 : 0) * nx;
        if (bc_y == XG_BC_HALO) src = halo_y + ok * nx + i0;  // pre-gathered row below the first one: (outer, 1, X)
      }
      uu[kz][0] = *reinterpret_cast<const T*>(src);  // the previous segment's last row: an L2 hit
    }
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) {
      const int64_t jr = (s_ < nrow) ? s_ : nrow - 1;  // clamp inside the array for short tails
      // rows nobody reads again stream past the caches (non-temporal); the segment's LAST u row is the next
      // segment's halo row and the v row carries the 8-byte neighbour loads, so those stay ordinary loads
      if ((ntl & 2) && s_ + 1 < SEG) uu[kz][s_ + 1] = __builtin_nontemporal_load(reinterpret_cast<const T*>(pu + (j0 + jr) * nx));
      else uu[kz][s_ + 1] = *reinterpret_cast<const T*>(pu + (j0 + jr) * nx);
      if (V > 1 && (ntl & 1)) {
        // the v row is read by this wave only: non-temporal, and the value left of a lane's vector comes from
        // the lane before it (lanes that left at the row's end are the highest ones); lane 0 loads its own.  The
        // shuffles come AFTER every load of the task has been issued (below): taken here, each one made the wave wait for
        // its row before the next row's loads went out -- four round trips to the memory one after the other
        vv[kz][s_] = __builtin_nontemporal_load(reinterpret_cast<const T*>(pv + jr * nx + i0));
        vl[kz][s_] = real(0);
        if ((threadIdx.x & 63) == 0 || edge)
          vl[kz][s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[ok * ny + j0 + jr] : pv[jr * nx + nidx];
      } else {
        vv[kz][s_] = *reinterpret_cast<const T*>(pv + jr * nx + i0);
        vl[kz][s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[ok * ny + j0 + jr]  // pre-gathered column left of the first: (outer, Y, 1)
                                                  : pv[jr * nx + nidx];
      }
    }
  }
  if (HAS_AREA) load_rows<T, SEG>(ar, area, a_base + j0 * a_sy + i0 * a_sx, a_sy, a_sx, nrow, (ntl & 4) != 0);
  if (V > 1 && (ntl & 1)) {
    const bool own = (threadIdx.x & 63) == 0 || edge;
#pragma unroll
    for (int kz = 0; kz < ZK; ++kz) {
#pragma unroll
      for (int s_ = 0; s_ < SEG; ++s_) {
        const real left = from_lane_below(vec_last(vv[kz][s_]));  // DPP wave_shr:1 (lane 0 reads 0 and is `own`)
        if (!own) vl[kz][s_] = left;
      }
    }
  }
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    if (kz >= nk) break;
    real* po = out + ((o + kz) * ny + j0) * nx + i0;
    const T u0 = f0 ? splat<T>(fill_y) : uu[kz][0];
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) {
      if (s_ < nrow) {
        const real left = fill_edge ? fill_x : vl[kz][s_];
        T z = dvdx_of(vv[kz][s_], left) - (uu[kz][s_ + 1] - (s_ == 0 ? u0 : uu[kz][s_]));
        if (HAS_AREA) z = z / ar[s_];
        stg_s<T, NTS>(po + s_ * nx, z);  // (`sc1 nt`, rule 16: vorticity +4.5, divergence +2.9 points; the two-output kernels keep `nt`)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K7b: fused horizontal divergence (delta_x u + delta_y v) / area of docs/ufunc_examples.md
// ("Divergence": u on (Y:center, X:left), v on (Y:left, X:center), both left -> center, i.e.
// padding_width (0,1) on both axes).  Mirror image of K7: SEG rows of u with their right
// neighbour, SEG+1 rows of v (the last one is the upper halo row of the segment).
// ------------------------------------------------------------------------------------------
template <int V, bool HAS_AREA, bool NTS, int SEG, int ZK = 1>
__global__ __launch_bounds__(BLOCK) void k_divergence(
    const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ area,
    real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny, int64_t nx, FastDiv ntile,
    FastDiv nseg, ZBand zb, int bc_x, real fill_x, int bc_y, real fill_y, AreaIdx ai, int64_t a_sy,
    int64_t a_sx, const real* __restrict__ halo_x, const real* __restrict__ halo_y, int ntl) {
  typedef typename VecT<V>::type T;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  const u32 tile = w - r * ntile.d;
  u32 oo, sg;
  if (HAS_AREA && zb.on) {
    if (!zband_map(zb, r, oo, sg)) return;
    oo *= ZK;
  } else {
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    sg = r - oo * nseg.d;
  }
  const int nk = (ZK > 1 && (int64_t)nouter - (int64_t)oo < ZK) ? (int)(nouter - oo) : ZK;
  const int64_t o = o0 + oo;
  const int64_t a_base = HAS_AREA ? area_outer_off(ai, o) : 0;
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  const bool edge = (i0 + V >= nx);
  const int64_t ridx = edge ? ((bc_x == XG_BC_PERIODIC) ? 0 : nx - 1) : i0 + V;
  const bool fill_edge = edge && (bc_x == XG_BC_FILL);

  T uu[ZK][SEG], vv[ZK][SEG + 1], ar[SEG];
  real ur[ZK][SEG];
  bool ftop = false;
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    const int64_t ok = o + ((kz < nk) ? kz : nk - 1);
    const real* pu = u + (ok * ny + j0) * nx;
    const real* pv = v + ok * ny * nx + i0;
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) {
      const int64_t jr = (s_ < nrow) ? s_ : nrow - 1;
      uu[kz][s_] = *reinterpret_cast<const T*>(pu + jr * nx + i0);
      ur[kz][s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[ok * ny + j0 + jr]  // pre-gathered column right of the last: (outer, Y, 1)
                                                : pu[jr * nx + ridx];
      vv[kz][s_] = *reinterpret_cast<const T*>(pv + (j0 + jr) * nx);
    }
    {
      int64_t q = j0 + nrow;  // the row above the segment's last row
      const real* src = pv + q * nx;
      if (q >= ny) {
        ftop = (bc_y == XG_BC_FILL);
        src = pv + ((bc_y == XG_BC_PERIODIC) ? 0 : ny - 1) * nx;
        if (bc_y == XG_BC_HALO) src = halo_y + ok * nx + i0;  // pre-gathered row above the last one: (outer, 1, X)
      }
      vv[kz][SEG] = *reinterpret_cast<const T*>(src);
    }
  }
  if (HAS_AREA) load_rows<T, SEG>(ar, area, a_base + j0 * a_sy + i0 * a_sx, a_sy, a_sx, nrow, (ntl & 4) != 0);
#pragma unroll
  for (int kz = 0; kz < ZK; ++kz) {
    if (kz >= nk) break;
    real* po = out + ((o + kz) * ny + j0) * nx + i0;
    const T top = ftop ? splat<T>(fill_y) : vv[kz][SEG];
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) {
      if (s_ < nrow) {
        const real right = fill_edge ? fill_x : ur[kz][s_];
        const T up = (s_ + 1 < nrow) ? vv[kz][s_ + 1] : top;
        T z = dudx_fwd(uu[kz][s_], right) + (up - vv[kz][s_]);
        if (HAS_AREA) z = z / ar[s_];
        stg_s<T, NTS>(po + s_ * nx, z);  // (`sc1 nt`, rule 16: vorticity +4.5, divergence +2.9 points; the two-output kernels keep `nt`)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// K7c: the two remaining fused grid ufuncs of docs/ufunc_examples.md, one field in, TWO fields out:
//   gradient: gx = (a - a[x-1]) / mx,  gy = (a - a[y-1]) / my      ("Gradient": center -> left on X, Y)
//   flux:     fx = u * interp(T, X),   fy = v * interp(T, Y)       ("Advection": center -> left on X, Y)
// Load pattern of K7/K8 (SEG+1 rows of the centre field + the 8-B left neighbour); the field is read
// once for both outputs: 24 B/cell instead of 32 (gradient), 40 instead of 80 (flux chain).
// ------------------------------------------------------------------------------------------
template <int V, int MODE, bool NTS, int SEG>   // MODE 0: gradient (optional metrics), 1: flux
__global__ __launch_bounds__(BLOCK) void k_pair2d(
    const real* __restrict__ a, const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ out_x,
    real* __restrict__ out_y, int64_t o0, u32 nouter, u32 nblk, int64_t ny, int64_t nx, FastDiv ntile, FastDiv nseg,
    int bc_x, real fill_x, int bc_y, real fill_y, const real* __restrict__ mx, AreaIdx aix, int64_t mx_sy,
    int64_t mx_sx, const real* __restrict__ my, AreaIdx aiy, int64_t my_sy, int64_t my_sx,
    const real* __restrict__ halo_x, const real* __restrict__ halo_y, ZBand zb, int ntl) {
  typedef typename VecT<V>::type T;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  const u32 tile = w - r * ntile.d;
  u32 oo, sg;
  if (zb.on) {  // band-major: the metric rows of a band of segments stay in the XCD's L2 for all outer indices (rule 4)
    if (!zband_map(zb, r, oo, sg)) return;
  } else {
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    sg = r - oo * nseg.d;
  }
  const int64_t o = o0 + oo;
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  const int64_t base = o * ny * nx;
  const real* pa = a + base + i0;
  const bool edge = (i0 == 0);
  const int64_t nidx = edge ? ((bc_x == XG_BC_PERIODIC) ? nx - 1 : 0) : i0 - 1;
  const bool fill_edge = edge && (bc_x == XG_BC_FILL);
  T aa[SEG + 1];
  real al[SEG];
  {
    int64_t q = j0 - 1;
    bool f = false;
    const real* src = pa + q * nx;
    if (q < 0) {
      f = (bc_y == XG_BC_FILL);
      src = pa + ((bc_y == XG_BC_PERIODIC) ? ny - 1 : 0) * nx;
      if (bc_y == XG_BC_HALO) src = halo_y + o * nx + i0;  // pre-gathered row below the first one: (outer, 1, X)
    }
    const T t = *reinterpret_cast<const T*>(src);
    aa[0] = f ? splat<T>(fill_y) : t;
  }
  // `ntl` bit 0 (vector lanes): rows the next segment does not need again are loaded non-temporally and the value left of a
  // lane's vector comes from the lane before it (DPP, after the loads) instead of an 8-byte load over the same cache lines --
  // K7's scheme.  Gradient without metrics 2.505 -> 2.401 ms (0.776 -> 0.810 of 8 TB/s), with one / two metric planes +1 %
  // (profiles/r06_kernels/r06be_ab_pair_nt.log).  What the metrics cost is their DIVISIONS, not their planes: none 0.805, one
  // 0.749, two 0.675 -- and the same plane passed twice 0.681, a second plane at shifted addresses 0.681 (r06bf_ab_grad_planes.log)
  const bool shl = V > 1 && (ntl & 1);
  const bool own = (threadIdx.x & 63) == 0 || edge;
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    const int64_t jr = j0 + ((s_ < nrow) ? s_ : nrow - 1);
    if (shl && s_ + 1 < SEG) aa[s_ + 1] = __builtin_nontemporal_load(reinterpret_cast<const T*>(pa + jr * nx));
    else aa[s_ + 1] = *reinterpret_cast<const T*>(pa + jr * nx);
    if (shl) {
      al[s_] = real(0);
      if (own) al[s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[o * ny + jr] : a[base + jr * nx + nidx];
    } else {
      al[s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[o * ny + jr]  // pre-gathered column left of the first: (outer, Y, 1)
                                            : a[base + jr * nx + nidx];
    }
  }
  if (shl) {
#pragma unroll
    for (int s_ = 0; s_ < SEG; ++s_) {
      const real left = from_lane_below(vec_last(aa[s_ + 1]));
      if (!own) al[s_] = left;
    }
  }
  const int64_t mxb = (MODE == 0 && mx) ? area_outer_off(aix, o) : 0;
  const int64_t myb = (MODE == 0 && my) ? area_outer_off(aiy, o) : 0;
  // (round 6: the metric rows / the u and v rows of all SEG rows loaded HERE, behind the field's loads and before anything
  // waits, in a host-decided vector form -- in the loop below each of them goes out only after the row before has been
  // stored, four round trips to the L2 one after the other -- made the gradient 9 % SLOWER (2.74 -> 3.00 ms, two process
  // pairs on one box) and left the flux where it was: what the row-by-row order has is stores leaving while later loads
  // arrive, the lesson of the scans' batch form.  profiles/r06_kernels/r06bb_ab_loads_first_vector_kernels.log)
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    if (s_ < nrow) {
      const int64_t j = j0 + s_;
      const real left = fill_edge ? fill_x : al[s_];
      T rx, ry;
      if (MODE == 0) {
        rx = dvdx_of(aa[s_ + 1], left);
        ry = aa[s_ + 1] - aa[s_];
        if (mx) rx = rx / ldm<T>(mx, mxb + j * mx_sy + i0 * mx_sx, mx_sx);
        if (my) ry = ry / ldm<T>(my, myb + j * my_sy + i0 * my_sx, my_sx);
      } else {
        // (u and v are read once; loading them non-temporally changed nothing: 4.234 / 4.262 ms, r06be_ab_pair_nt.log)
        const T uu = *reinterpret_cast<const T*>(u + base + j * nx + i0);
        const T vv = *reinterpret_cast<const T*>(v + base + j * nx + i0);
        rx = uu * interp_left_of(aa[s_ + 1], left);
        ry = vv * op2<XG_OP_INTERP>(aa[s_], aa[s_ + 1]);
      }
      // (rule 16: the flux runs at 0.77 with `sc1 nt` in every round, with `nt` at 0.77 or 0.70 from process to process; the
      // gradient -- one input stream, two metric planes to keep in the L2 -- loses 2 points with `sc1 nt` on either output or on
      // both (three rounds each, profiles/history/r03bl_ab_grad_drop.jsonl; r03ba_*; +2 once in r03be_*) and keeps `nt`)
      if (MODE == 1) {
        stg_s<T, NTS>(out_x + base + j * nx + i0, rx);
        stg_s<T, NTS>(out_y + base + j * nx + i0, ry);
      } else {
        stg<T, NTS>(out_x + base + j * nx + i0, rx);
        stg<T, NTS>(out_y + base + j * nx + i0, ry);
      }
    }
  }
}

#endif  // !XG_INT

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int XG_FN(xg_binary)(int op, const real* a, const int64_t* a_strides, const real* b, const int64_t* b_strides,
                  real* out, const int64_t* shape, int ndim, void* stream) {
  if (!a || !b || !out || (ndim > 0 && (!shape || !a_strides || !b_strides))) return fail(XG_ERR_INVALID, "NULL argument");
  if (op < XG_BIN_MUL || op > XG_BIN_SUB) return fail(XG_ERR_INVALID, "unknown binary op %d", op);
#ifdef XG_INT
  if (op == XG_BIN_DIV) return fail(XG_ERR_UNSUPPORTED, "true division leaves the integer domain: convert to float64 first");
#endif
  if (ndim < 0 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [0,%d]", ndim, XG_MAX_NDIM);
  BinGeo g;
  memset(&g, 0, sizeof(g));
  // drop size-1 dims, coalesce neighbours compatible for BOTH operands
  int n = 0;
  int64_t total = 1;
  for (int d = 0; d < ndim; ++d) {
    if (shape[d] < 0) return fail(XG_ERR_INVALID, "negative extent");
    total *= shape[d];
    if (shape[d] == 1) continue;
    if (n > 0 && g.sa[n - 1] == a_strides[d] * shape[d] && g.sb[n - 1] == b_strides[d] * shape[d]) {
      g.shape[n - 1] *= shape[d];
      g.sa[n - 1] = a_strides[d];
      g.sb[n - 1] = b_strides[d];
    } else {
      g.shape[n] = shape[d];
      g.sa[n] = a_strides[d];
      g.sb[n] = b_strides[d];
      ++n;
    }
  }
  if (total == 0) return XG_OK;
  if (n == 0) { g.shape[0] = 1; g.sa[0] = 0; g.sb[0] = 0; n = 1; }
  g.ndim = n;
  const int64_t last = g.shape[n - 1];
  const int64_t sa = g.sa[n - 1], sb = g.sb[n - 1];
  bool v2 = (last % NV == 0) && aligned16(out) && (sa == 0 || sa == 1) && (sb == 0 || sb == 1);
  if (v2 && sa == 1) {
    if (!aligned16(a)) v2 = false;
    for (int d = 0; d < n - 1; ++d) if (g.sa[d] % NV) v2 = false;
  }
  if (v2 && sb == 1) {
    if (!aligned16(b)) v2 = false;
    for (int d = 0; d < n - 1; ++d) if (g.sb[d] % NV) v2 = false;
  }
  const int V = v2 ? NV : 1;
  g.shape[n - 1] = last / V;
  g.total = total / V;
  g.idx32 = (tune().bin_idx32 && (u64)g.total < 0xffffffffull) ? 1 : 0;
  for (int d = 0; d < XG_MAX_NDIM; ++d) g.fs[d] = make_fastdiv(d < n ? (u64)g.shape[d] : 1);
  u64 nitems = (u64)g.total;
  ZBand zb = make_zband(false, 0, 0, 1);
  if (tune().zband && n == 2 && (g.sa[0] == 0) != (g.sb[0] == 0) && g.shape[0] >= 2) {
    // exactly one operand is broadcast along the slow dim and re-read once per level: band it
    const u64 Z = (u64)g.shape[0], P = (u64)g.shape[1];
    const u32 B = 16384;  // items per band: 256 KiB of the broadcast operand at 16 B per item
    const u64 padded = ((P + B - 1) / B) * B * Z;
    if (P > 2 * (u64)B && padded < 0x7fffffffull) {
      zb = make_zband(true, Z, P, B);
      if (zb.on) nitems = padded;
    }
  }
  const u64 nblocks = (nitems + BLOCK - 1) / BLOCK;
  int rc;
  if ((rc = check_grid(nblocks + 8))) return rc;
  const u32 grid = (u32)(((nblocks + 7) / 8) * 8);  // XCD-banded block order
  // an operand without a broadcast dim is read exactly once: stream it past the caches
  auto streamed_once = [&](const int64_t* st_) { for (int d = 0; d < n; ++d) if (st_[d] == 0 && g.shape[d] > 1) return 0; return 1; };
  const int nta = tune().nt_load && streamed_once(g.sa), ntb = tune().nt_load && streamed_once(g.sb);
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
#define XG_GO(O, V_, NTS) hipLaunchKernelGGL((k_binary<O, V_, NTS>), dim3(grid), dim3(BLOCK), 0, st, a, b, out, g, zb, (u32)nblocks, nta, ntb)
#define XG_O(O) do { if (V > 1) { if (nts) XG_GO(O, NV, true); else XG_GO(O, NV, false); } else { if (nts) XG_GO(O, 1, true); else XG_GO(O, 1, false); } } while (0)
  switch (op) { case XG_BIN_MUL: XG_O(XG_BIN_MUL); break; case XG_BIN_DIV: XG_O(XG_BIN_DIV); break; case XG_BIN_ADD: XG_O(XG_BIN_ADD); break; default: XG_O(XG_BIN_SUB); }
#undef XG_O
#undef XG_GO
  XG_LAUNCH_CHECK();
  return XG_OK;
}

#ifndef XG_INT
// may every lane of a V-wide kernel load its piece of a metric / area plane as ONE aligned vector in every row and at
// every outer index?  (lanes start at multiples of NV along X)
static bool plane_vec_ok(const real* m, const AreaIdx& ai, int64_t sy, int64_t sx) {
  if (!m || sx != 1 || sy % NV != 0 || !aligned16(m)) return false;
  for (int d = 0; d < ai.n; ++d)
    if (ai.stride[d] % NV != 0) return false;
  return true;
}

static int curl_div_impl(bool div, const real* u, const real* v, const real* area, const int64_t* area_strides,
                         real* out, const int64_t* shape, int ndim, int bc_x, real fill_x, int bc_y, real fill_y,
                         void* stream, const real* halo_x = nullptr, const real* halo_y = nullptr) {
  if (!u || !v || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if (ndim < 2 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [2,%d]", ndim, XG_MAX_NDIM);
  if (area && !area_strides) return fail(XG_ERR_INVALID, "area without strides");
  if (bc_x < XG_BC_PERIODIC || bc_x > XG_BC_HALO || bc_y < XG_BC_PERIODIC || bc_y > XG_BC_HALO)
    return fail(XG_ERR_INVALID, "vorticity / divergence need a boundary mode on both axes");
  if ((bc_x == XG_BC_HALO && !halo_x) || (bc_y == XG_BC_HALO && !halo_y))
    return fail(XG_ERR_INVALID, "XG_BC_HALO without the halo buffer of that axis");
  const int64_t ny = shape[ndim - 2], nx = shape[ndim - 1];
  int64_t outer = 1;
  for (int d = 0; d < ndim - 2; ++d) outer *= shape[d];
  if (outer == 0 || ny == 0 || nx == 0) return XG_OK;
  // area: (Y, X) strides + one stride per leading dim (0 = broadcast); adjacent leading dims are merged
  int64_t a_sy = 0, a_sx = 0;
  bool area_bcast_all = true;
  AreaIdx ai;
  memset(&ai, 0, sizeof(ai));
  for (int d = 0; d < XG_MAX_NDIM; ++d) ai.fd[d] = make_fastdiv(1);
  if (area) {
    a_sy = area_strides[ndim - 2];
    a_sx = area_strides[ndim - 1];
    for (int d = 0; d < ndim - 2; ++d) {
      if (shape[d] == 1) continue;
      const int64_t st = area_strides[d];
      if (st != 0) area_bcast_all = false;
      if (ai.n > 0 && ai.stride[ai.n - 1] == st * shape[d]) {  // merges with the previous (slower) dim
        ai.fd[ai.n - 1] = make_fastdiv((u64)ai.fd[ai.n - 1].d * (u64)shape[d]);
        ai.stride[ai.n - 1] = st;
        continue;
      }
      ai.fd[ai.n] = make_fastdiv((u64)shape[d]);
      ai.stride[ai.n] = st;
      ++ai.n;
    }
    if (outer > 0xffffffffll) return fail(XG_ERR_UNSUPPORTED, "more than 2^32 (Y,X) planes");
  }
  const int V = (aligned16(u) && aligned16(v) && aligned16(out) && nx % NV == 0 &&
                 (bc_y != XG_BC_HALO || aligned16(halo_y))) ? NV : 1;
  constexpr int SEG = XG_FUSED_SEG;
  const u64 ntile = (u64)((nx + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 nseg_rows = (u64)((ny + SEG - 1) / SEG);
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
  // bit 0: v rows non-temporal (neighbour by lane shuffle), bit 1: inner u rows, bit 2: the area rows are aligned vectors
  const int vnt = (tune().nt_load ? (tune().vec_nt & 3) : 0) | ((V > 1 && plane_vec_ok(area, ai, a_sy, a_sx)) ? 4 : 0);
  const u64 nseg = nseg_rows;
  const u64 per_outer = ntile * nseg;
  if (per_outer > MAX_ITEMS) return fail(XG_ERR_UNSUPPORTED, "extent too large for the vorticity kernel");
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  // band height: the area rows of a band must survive in the XCD's 4 MB L2 while TWO fields and the output of all its
  // levels stream by.  16 rows: PMC reads 1.06x the algorithmic bytes (the halo u row of every band and level is the
  // 6 %); 24 rows 1.17x, 32 rows 1.26x -- the area is then re-read from the fabric once per level group -- at the same
  // speed within 1 % on an otherwise idle device (profiles/history/r03g_*, r03h_*)
  const u32 zbr = (u32)(tune().vec_zb_rows > 1 ? tune().vec_zb_rows : 16);
  const u32 ZB_SEGS = (zbr + SEG - 1) / SEG;
  ZBand zb = make_zband(false, 0, 0, 1);
  u64 outer_step = outer_per;
  // levels per wave-task sharing the area rows: z-banded launches with the default vector lanes and stores only
  int zk = (V > 1 && nts) ? tune().vec_zk : 1;
  zk = zk >= 4 ? 4 : (zk >= 2 ? 2 : 1);
  u64 zgroups = (u64)outer;
  if (area && area_bcast_all && tune().zband && outer >= 2) {
    const u64 padded_segs = ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS;
    zgroups = ((u64)outer + zk - 1) / zk;
    if (padded_segs * zgroups * ntile <= MAX_ITEMS) {
      zb = make_zband(true, zgroups, nseg, ZB_SEGS);
      if (zb.on) outer_step = (u64)outer;  // one launch over all levels
    }
  }
  if (!zb.on) zk = 1;
  for (int64_t o0 = 0; o0 < outer; o0 += (int64_t)outer_step) {
    const u32 nouter = (u32)((outer - o0 < (int64_t)outer_step) ? outer - o0 : (int64_t)outer_step);
    const u64 units = zb.on ? ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS * zgroups * ntile : (u64)nouter * per_outer;
    const u32 nblk = (u32)((units + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
#define XG_GZ(V_, A_, NTS, ZK_) do { if (div) hipLaunchKernelGGL((k_divergence<V_, A_, NTS, SEG, ZK_>), dim3(grid), dim3(BLOCK), 0, st, u, v, area, out, o0, nouter, nblk, ny, nx, fnt, fns, zb, bc_x, fill_x, bc_y, fill_y, ai, a_sy, a_sx, halo_x, halo_y, vnt); \
                                else hipLaunchKernelGGL((k_vorticity<V_, A_, NTS, SEG, ZK_>), dim3(grid), dim3(BLOCK), 0, st, u, v, area, out, o0, nouter, nblk, ny, nx, fnt, fns, zb, bc_x, fill_x, bc_y, fill_y, ai, a_sy, a_sx, halo_x, halo_y, vnt); } while (0)
#define XG_GO(V_, A_, NTS) XG_GZ(V_, A_, NTS, 1)
#define XG_A(V_, A_) do { if (nts) XG_GO(V_, A_, true); else XG_GO(V_, A_, false); } while (0)
    if (zk == 4) XG_GZ(NV, true, true, 4);
    else if (zk == 2) XG_GZ(NV, true, true, 2);
    else if (V > 1) { if (area) XG_A(NV, true); else XG_A(NV, false); }
    else { if (area) XG_A(1, true); else XG_A(1, false); }
#undef XG_A
#undef XG_GO
#undef XG_GZ
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_vorticity)(const real* u, const real* v, const real* area, const int64_t* area_strides, real* out,
                     const int64_t* shape, int ndim, int bc_x, real fill_x, int bc_y, real fill_y, void* stream) {
  if (bc_x == XG_BC_HALO || bc_y == XG_BC_HALO) return fail(XG_ERR_INVALID, "XG_BC_HALO needs xg_vorticity_halo");
  return curl_div_impl(false, u, v, area, area_strides, out, shape, ndim, bc_x, fill_x, bc_y, fill_y, stream);
}

int XG_FN(xg_divergence)(const real* u, const real* v, const real* area, const int64_t* area_strides, real* out,
                      const int64_t* shape, int ndim, int bc_x, real fill_x, int bc_y, real fill_y, void* stream) {
  if (bc_x == XG_BC_HALO || bc_y == XG_BC_HALO) return fail(XG_ERR_INVALID, "XG_BC_HALO needs xg_divergence_halo");
  return curl_div_impl(true, u, v, area, area_strides, out, shape, ndim, bc_x, fill_x, bc_y, fill_y, stream);
}

int XG_FN(xg_vorticity_halo)(const real* u, const real* v, const real* halo_x, const real* halo_y, const real* area,
                          const int64_t* area_strides, real* out, const int64_t* shape, int ndim, int bc_x,
                          real fill_x, int bc_y, real fill_y, void* stream) {
  return curl_div_impl(false, u, v, area, area_strides, out, shape, ndim, bc_x, fill_x, bc_y, fill_y, stream, halo_x, halo_y);
}

int XG_FN(xg_divergence_halo)(const real* u, const real* v, const real* halo_x, const real* halo_y, const real* area,
                           const int64_t* area_strides, real* out, const int64_t* shape, int ndim, int bc_x,
                           real fill_x, int bc_y, real fill_y, void* stream) {
  return curl_div_impl(true, u, v, area, area_strides, out, shape, ndim, bc_x, fill_x, bc_y, fill_y, stream, halo_x, halo_y);
}

static int area_index(const real* m, const int64_t* strides, const int64_t* shape, int ndim, AreaIdx* ai, int64_t* sy,
                      int64_t* sx) {
  memset(ai, 0, sizeof(*ai));
  for (int d = 0; d < XG_MAX_NDIM; ++d) ai->fd[d] = make_fastdiv(1);
  *sy = *sx = 0;
  if (!m) return 0;
  if (!strides) return fail(XG_ERR_INVALID, "metric without strides");
  *sy = strides[ndim - 2];
  *sx = strides[ndim - 1];
  for (int d = 0; d < ndim - 2; ++d) {
    if (shape[d] == 1) continue;
    const int64_t st = strides[d];
    if (ai->n > 0 && ai->stride[ai->n - 1] == st * shape[d]) {
      ai->fd[ai->n - 1] = make_fastdiv((u64)ai->fd[ai->n - 1].d * (u64)shape[d]);
      ai->stride[ai->n - 1] = st;
      continue;
    }
    ai->fd[ai->n] = make_fastdiv((u64)shape[d]);
    ai->stride[ai->n] = st;
    ++ai->n;
  }
  return 0;
}

static int pair2d_impl(int mode, const real* a, const real* u, const real* v, real* out_x, real* out_y,
                       const int64_t* shape, int ndim, int bc_x, real fill_x, int bc_y, real fill_y, const real* mx,
                       const int64_t* mx_strides, const real* my, const int64_t* my_strides, void* stream,
                       const real* halo_x = nullptr, const real* halo_y = nullptr) {
  if (!a || !out_x || !out_y || !shape || (mode == 1 && (!u || !v))) return fail(XG_ERR_INVALID, "NULL array argument");
  if ((bc_x == XG_BC_HALO && !halo_x) || (bc_y == XG_BC_HALO && !halo_y)) return fail(XG_ERR_INVALID, "halo mode without a halo array");
  const int bc_max = (halo_x || halo_y) ? XG_BC_HALO : XG_BC_EXTEND;
  if (ndim < 2 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [2,%d]", ndim, XG_MAX_NDIM);
  if (bc_x < XG_BC_PERIODIC || bc_x > bc_max || bc_y < XG_BC_PERIODIC || bc_y > bc_max)
    return fail(XG_ERR_INVALID, "gradient / flux need a boundary mode on both axes");
  const int64_t ny = shape[ndim - 2], nx = shape[ndim - 1];
  int64_t outer = 1;
  for (int d = 0; d < ndim - 2; ++d) outer *= shape[d];
  if (outer == 0 || ny == 0 || nx == 0) return XG_OK;
  if (outer > 0xffffffffll) return fail(XG_ERR_UNSUPPORTED, "more than 2^32 (Y,X) planes");
  AreaIdx aix, aiy;
  int64_t mx_sy, mx_sx, my_sy, my_sx;
  int rc;
  if ((rc = area_index(mx, mx_strides, shape, ndim, &aix, &mx_sy, &mx_sx))) return rc;
  if ((rc = area_index(my, my_strides, shape, ndim, &aiy, &my_sy, &my_sx))) return rc;
  bool al = aligned16(a) && aligned16(out_x) && aligned16(out_y) && nx % NV == 0;
  if (mode == 1) al = al && aligned16(u) && aligned16(v);
  if (bc_y == XG_BC_HALO) al = al && aligned16(halo_y);
  const int V = al ? NV : 1;
  constexpr int SEG = XG_FUSED_SEG;
  const u64 ntile = (u64)((nx + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 nseg = (u64)((ny + SEG - 1) / SEG);
  const u64 per_outer = ntile * nseg;
  if (per_outer > MAX_ITEMS) return fail(XG_ERR_UNSUPPORTED, "extent too large for the fused two-output kernel");
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
  const int vnt = tune().nt_load ? (tune().vec_nt & 1) : 0;  // bit 0: field rows non-temporal + the left neighbour by DPP
  // gradient with metrics that every outer index shares (dxC(Y,X), dyC(Y,X) under a (Z,Y,X) field): band-major order, or
  // both planes come from the fabric again for every level (0.56 of 8 TB/s level-major).  Two metrics: 8-row bands (rule 13)
  ZBand zb = make_zband(false, 0, 0, 1);
  u64 outer_step = outer_per;
  // rows per band: twice `vec_zb_rows` (32) with one metric plane, `vec_zb_rows` (16) with two -- round 4, PMC per band
  // height (profiles/history/r04b_ab_bands_grad.log): two metrics 8 -> 16 rows reads 5.97 -> 5.65 GB (traffic 1.041 -> 1.021x), 32
  // rows 5.61 GB, all at the same speed; the two `nt`-stored outputs leave the L2 room the one-output kernels do not have
  // (their cliff sits between 8 and 16 rows for two metrics, r04b_ab_bands_met.log)
  const u32 gzb = (u32)(tune().vec_zb_rows > 1 ? tune().vec_zb_rows : 16);
  const u32 ZB_SEGS = (u32)((((mx && my) ? gzb : 2 * gzb) + SEG - 1) / SEG);
  auto shared = [](const real* m, const AreaIdx& ai) {  // absent, or broadcast along every leading dim
    for (int d = 0; m && d < ai.n; ++d)
      if (ai.stride[d] != 0) return false;
    return true;
  };
  if (mode == 0 && (mx || my) && shared(mx, aix) && shared(my, aiy) && tune().zband && outer >= 2) {
    const u64 padded = ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS * (u64)outer * ntile;
    if (padded <= MAX_ITEMS) {
      zb = make_zband(true, (u64)outer, nseg, ZB_SEGS);
      if (zb.on) outer_step = (u64)outer;
    }
  }
  for (int64_t o0 = 0; o0 < outer; o0 += (int64_t)outer_step) {
    const u32 nouter = (u32)((outer - o0 < (int64_t)outer_step) ? outer - o0 : (int64_t)outer_step);
    const u64 waves = zb.on ? ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS * (u64)outer * ntile : (u64)nouter * per_outer;
    const u32 nblk = (u32)((waves + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
#define XG_GO(V_, M_, NTS) hipLaunchKernelGGL((k_pair2d<V_, M_, NTS, SEG>), dim3(grid), dim3(BLOCK), 0, st, a, u, v, out_x, out_y, o0, nouter, nblk, ny, nx, fnt, fns, bc_x, fill_x, bc_y, fill_y, mx, aix, mx_sy, mx_sx, my, aiy, my_sy, my_sx, halo_x, halo_y, zb, vnt)
#define XG_M(V_, M_) do { if (nts) XG_GO(V_, M_, true); else XG_GO(V_, M_, false); } while (0)
    if (V > 1) { if (mode) XG_M(NV, 1); else XG_M(NV, 0); }
    else { if (mode) XG_M(1, 1); else XG_M(1, 0); }
#undef XG_M
#undef XG_GO
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_gradient)(const real* a, real* out_x, real* out_y, const int64_t* shape, int ndim, int bc_x, real fill_x,
                    int bc_y, real fill_y, const real* mx, const int64_t* mx_strides, const real* my,
                    const int64_t* my_strides, void* stream) {
  return pair2d_impl(0, a, nullptr, nullptr, out_x, out_y, shape, ndim, bc_x, fill_x, bc_y, fill_y, mx, mx_strides, my,
                     my_strides, stream);
}

int XG_FN(xg_flux)(const real* u, const real* v, const real* t, real* out_x, real* out_y, const int64_t* shape, int ndim,
                int bc_x, real fill_x, int bc_y, real fill_y, void* stream) {
  return pair2d_impl(1, t, u, v, out_x, out_y, shape, ndim, bc_x, fill_x, bc_y, fill_y, nullptr, nullptr, nullptr, nullptr,
                     stream);
}

int XG_FN(xg_gradient_halo)(const real* a, const real* halo_x, const real* halo_y, real* out_x, real* out_y,
                         const int64_t* shape, int ndim, int bc_x, real fill_x, int bc_y, real fill_y, const real* mx,
                         const int64_t* mx_strides, const real* my, const int64_t* my_strides, void* stream) {
  return pair2d_impl(0, a, nullptr, nullptr, out_x, out_y, shape, ndim, bc_x, fill_x, bc_y, fill_y, mx, mx_strides, my,
                     my_strides, stream, halo_x, halo_y);
}

int XG_FN(xg_flux_halo)(const real* u, const real* v, const real* t, const real* halo_x, const real* halo_y, real* out_x,
                     real* out_y, const int64_t* shape, int ndim, int bc_x, real fill_x, int bc_y, real fill_y,
                     void* stream) {
  return pair2d_impl(1, t, u, v, out_x, out_y, shape, ndim, bc_x, fill_x, bc_y, fill_y, nullptr, nullptr, nullptr, nullptr,
                     stream, halo_x, halo_y);
}

#endif  // !XG_INT

}  // extern "C"
