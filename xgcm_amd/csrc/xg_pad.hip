// xg_pad.hip -- generic N-D pad (user grid ufuncs) and the token-map gather of complex topologies
// Part of libxgcm_hip.so; compiled twice (real = double / -DXG_F32), see xg_common.hpp.

#include "xg_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// generic N-D pad (user grid ufuncs with arbitrary widths; padding.py:765-871).  Steps are
// stored in APPLICATION order; the kernel walks them backwards: a fill halo of a later-applied
// axis wins over anything an earlier axis would have produced (numpy.pad chain semantics).
// ------------------------------------------------------------------------------------------
struct PadGeo {
  int ndim;
  int64_t total;
  int64_t out_shape[XG_MAX_NDIM], out_stride[XG_MAX_NDIM];
  int64_t in_shape[XG_MAX_NDIM], in_stride[XG_MAX_NDIM];
  int64_t lo[XG_MAX_NDIM];
  int bc[XG_MAX_NDIM];
  real fill[XG_MAX_NDIM];
  // u32 path: the output index is peeled dim by dim in MEMORY order (innermost first) with
  // multiply-shift division; mem_step[k] = which application step owns memory dim k
  FastDiv mem_fd[XG_MAX_NDIM];
  int mem_step[XG_MAX_NDIM];
};

// One thread per output element.  Measured alternatives: 16-B output groups per thread (the K1g
// trick) double the index arithmetic per thread and LOSE (47 % -> 35 % of 8 TB/s): this kernel is
// bound by its per-element address computation, not by the 8-B accesses.
template <typename I>
__global__ __launch_bounds__(BLOCK) void k_pad(const real* __restrict__ in, real* __restrict__ out, PadGeo p) {
  const int64_t gid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (gid >= p.total) return;
  int64_t coord[XG_MAX_NDIM];  // per application step
  if (sizeof(I) == 4) {
    u32 rem = (u32)gid;
#pragma unroll
    for (int k = XG_MAX_NDIM - 1; k >= 0; --k) {
      if (k < p.ndim) {
        const u32 q = fdiv(rem, p.mem_fd[k]);
        coord[p.mem_step[k]] = (int64_t)(rem - q * p.mem_fd[k].d);
        rem = q;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < XG_MAX_NDIM; ++t)
      if (t < p.ndim) coord[t] = (int64_t)(((u64)gid / (u64)p.out_stride[t]) % (u64)p.out_shape[t]);
  }
  int64_t src = 0;
  bool filled = false;
  real fv = real(0);
#pragma unroll
  for (int t = XG_MAX_NDIM - 1; t >= 0; --t) {
    if (t < p.ndim && !filled) {
      int64_t q = coord[t] - p.lo[t];
      const int64_t n = p.in_shape[t];
      if (q < 0 || q >= n) {
        if (p.bc[t] == XG_BC_FILL) { filled = true; fv = p.fill[t]; }
        else if (p.bc[t] == XG_BC_PERIODIC) { q %= n; if (q < 0) q += n; }
        else { q = (q < 0) ? 0 : n - 1; }
      }
      src += q * p.in_stride[t];
    }
  }
  out[gid] = filled ? fv : in[src];
}

// K_pad rows: the same pad with the index work hoisted to the scalar unit.  A wave owns 64*V consecutive
// cells of ONE output row (row = every dim but the innermost); the row's coordinates, the fill /
// wrap / clamp decisions of the outer dims and the source row offset are wave-uniform, a lane only
// resolves the innermost coordinate.  The walk order of the reference's chain is kept by splitting
// the outer steps into those applied after the innermost dim (they win) and those applied before.
// V == NV when the innermost dim is not padded: rows are straight 16-B copies or fills.
template <int V, bool INNER, int TPW>
__global__ __launch_bounds__(BLOCK) void k_pad_rows(const real* __restrict__ in, real* __restrict__ out, PadGeo p,
                                                    u32 nrows, FastDiv ntile, int nt) {
  typedef typename VecT<V>::type T;
  // nt & 1: non-temporal vector stores, nt & 2: non-temporal loads of the straight rows (streamed once); nt & 4: the
  // workgroups of the launch cut into 8 contiguous bands, one per XCD
  u32 lb = blockIdx.x;
  if (nt & 4) {
    const u32 pb = (gridDim.x + 7) >> 3;
    lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  }
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  // ntile = wave-tasks per row; a wave-task = TPW consecutive 64-lane tiles of one row (unrolled: their loads overlap): the row
  // logic below is a few hundred scalar instructions, more than a 1-KB tile's share of the HBM time (DESIGN rule 15)
  const u32 r = fdiv(w, ntile);
  if (r >= nrows) return;
  constexpr u32 tpw = TPW;
  const u32 tile0 = (w - r * ntile.d) * tpw;
  const int nd = p.ndim;
  const int t_in = p.mem_step[nd - 1];  // application step of the innermost memory dim
  // peel the row index over the outer memory dims (innermost of them first)
  int64_t coord[XG_MAX_NDIM];
  u32 rem = r;
#pragma unroll
  for (int k = XG_MAX_NDIM - 2; k >= 0; --k) {
    if (k < nd - 1) {
      const u32 q = fdiv(rem, p.mem_fd[k]);
      coord[p.mem_step[k]] = (int64_t)(rem - q * p.mem_fd[k].d);
      rem = q;
    }
  }
  int64_t src = 0;
  bool fill_after = false, fill_before = false;
  real fv_after = real(0), fv_before = real(0);
#pragma unroll
  for (int t = XG_MAX_NDIM - 1; t >= 0; --t) {
    if (t < nd && t != t_in) {
      const bool later = t > t_in;
      if ((later && fill_after) || (!later && (fill_after || fill_before))) continue;
      int64_t q = coord[t] - p.lo[t];
      const int64_t n = p.in_shape[t];
      if (q < 0 || q >= n) {
        if (p.bc[t] == XG_BC_FILL) {
          if (later) { fill_after = true; fv_after = p.fill[t]; }
          else { fill_before = true; fv_before = p.fill[t]; }
        } else if (p.bc[t] == XG_BC_PERIODIC) { q %= n; if (q < 0) q += n; }
        else { q = (q < 0) ? 0 : n - 1; }
      }
      src += q * p.in_stride[t];
    }
  }
  const int64_t Lo = p.out_shape[t_in], Li = p.in_shape[t_in];
  auto elem = [&](int64_t xx) -> real {  // value of output cell xx of this row
    if (fill_after) return fv_after;
    int64_t q = xx - p.lo[t_in];
    if (q < 0 || q >= Li) {
      if (p.bc[t_in] == XG_BC_FILL) return p.fill[t_in];
      if (p.bc[t_in] == XG_BC_PERIODIC) { q %= Li; if (q < 0) q += Li; }
      else q = (q < 0) ? 0 : Li - 1;
    }
    return fill_before ? fv_before : in[src + q];
  };
  real* drow = out + (int64_t)r * Lo;
#pragma unroll
  for (u32 tile = tile0; tile < tile0 + tpw; ++tile) {
  if (V > 1 && !INNER) {  // aligned rows, innermost dim not padded: straight vector copies or fills
    const int64_t x = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
    if (x >= Lo) break;
    T val;
    if (fill_after) val = splat<T>(fv_after);
    else if (fill_before) val = splat<T>(fv_before);
    else val = (nt & 2) ? ldg<T, true>(in + src + x) : *reinterpret_cast<const T*>(in + src + x);
    if (nt & 1) stg<T, true>(drow + x, val);
    else *reinterpret_cast<T*>(drow + x) = val;
  } else if (V > 1) {
    // any row length: the row starts `lead` cells before a 16-B boundary of the output; those cells and
    // the cells after the last whole group go out as scalars (lane 0 of tile 0 / the lane that owns them),
    // everything in between as NV narrow gathers + one 16-B store
    const int64_t lead = (NV - (int64_t)(((int64_t)r * Lo) % NV)) % NV;
    const int lane = threadIdx.x & 63;
    if (tile == 0 && lane == 0)
      for (int64_t xx = 0; xx < lead && xx < Lo; ++xx) drow[xx] = elem(xx);
    const int64_t x = lead + ((int64_t)tile * WAVE + lane) * NV;
    if (x >= Lo) break;
    if (x + NV <= Lo) {
      dv val;
      const int64_t q0 = x - p.lo[t_in];
      if (!fill_after && !fill_before && q0 >= 0 && q0 + NV <= Li) {
        // the group's sources lie inside the row (all but the groups touching a halo): no per-element
        // wrap / clamp / fill logic, NV consecutive narrow loads served by L1
        // (one aligned 16-B load per lane + the next vector's cells from the neighbouring lane by DPP, rule 12, measured:
        // padX 0.644 -> 0.566; rows whose sources ARE aligned gain nothing either, 0.673 -> 0.677 -- the loads are not what
        // bounds this kernel; profiles/history/r03ak_ab_pad_dpp.jsonl)
        const real* s = in + src + q0;
#pragma unroll
        for (int k = 0; k < NV; ++k) val[k] = s[k];
      } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) val[k] = elem(x + k);
      }
      if (nt & 1) stg<dv, true>(drow + x, val);
      else *reinterpret_cast<dv*>(drow + x) = val;
    } else {
      for (int64_t xx = x; xx < Lo; ++xx) drow[xx] = elem(xx);
    }
  } else {
    const int64_t x = (int64_t)tile * WAVE + (threadIdx.x & 63);
    if (x >= Lo) break;
    drow[x] = elem(x);
  }
  }
}

// ------------------------------------------------------------------------------------------
// halo gather through a token map (complex topologies: north fold, face connections;
// padding.py:260-572,619-762).  The host turns the reference's padding procedure into ONE
// int64 token per cell of the padded "mapped" dims (face/Y/X ...), shared by all other dims:
//   |t| in [1, 2^62):  source element k = |t| - 1; k < P0 -> `in`, else `partner` (k - P0),
//                      k counted row-major over that source's mapped dims
//   |t| >= 2^62:       fill value number |t| - 2^62
//   t < 0:             negated (vector components across a fold / reversed connection)
// Interior cells never read the map (the padded interior IS the input), so the map costs
// traffic only on the halo frame.
// ------------------------------------------------------------------------------------------
#define XG_TOKEN_FILL_BASE (1ll << 62)
struct GatherSrc {
  int n_mapped;
  int64_t m_extent[XG_MAX_NDIM], m_stride[XG_MAX_NDIM];  // mapped dims in the source's own order
  int64_t u_stride[XG_MAX_NDIM];                         // per OUT dim; 0 for mapped dims
  int64_t mapped_size;                                   // prod(m_extent)
  int trailing;  // the mapped dims are the source's trailing dims: element k sits at offset k of its block
};
struct GatherGeo {
  int ndim;
  int64_t total;
  int64_t out_shape[XG_MAX_NDIM];
  int mapped[XG_MAX_NDIM];
  int64_t lo[XG_MAX_NDIM];        // interior offset per out dim (mapped dims)
  int64_t in_shape[XG_MAX_NDIM];  // per out dim
  int64_t in_stride[XG_MAX_NDIM];
  real fills[XG_MAX_NDIM];
  int n_fills;
  GatherSrc src[2];
  FastDiv out_fd[XG_MAX_NDIM];
};

// offset of element k (row-major over the source's mapped dims) inside the source array
__device__ __forceinline__ int64_t gather_mapped_off(const GatherSrc& S, int64_t k) {
  if (S.trailing) return k;
  int64_t o = 0;
#pragma unroll
  for (int m = XG_MAX_NDIM - 1; m >= 0; --m) {
    if (m < S.n_mapped) {
      const int64_t n = S.m_extent[m];
      const int64_t q = k / n;
      o += (k - q * n) * S.m_stride[m];
      k = q;
    }
  }
  return o;
}

template <typename I>
__global__ __launch_bounds__(BLOCK) void k_gather(const real* __restrict__ in, const real* __restrict__ partner,
                                                  real* __restrict__ out, const int64_t* __restrict__ tokens,
                                                  GatherGeo g) {
  const int64_t gid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (gid >= g.total) return;
  I rem = (I)gid;
  int64_t c[XG_MAX_NDIM];
#pragma unroll
  for (int d = XG_MAX_NDIM - 1; d >= 0; --d) {
    if (d < g.ndim) {
      const I n = (I)g.out_shape[d];
      const I q = (sizeof(I) == 4) ? (I)fdiv((u32)rem, g.out_fd[d]) : rem / n;
      c[d] = (int64_t)(rem - q * n);
      rem = q;
    }
  }
  bool interior = true;
  int64_t off = 0, p = 0;
#pragma unroll
  for (int d = 0; d < XG_MAX_NDIM; ++d) {
    if (d < g.ndim) {
      if (g.mapped[d]) {
        const int64_t ci = c[d] - g.lo[d];
        interior = interior && ci >= 0 && ci < g.in_shape[d];
        off += ci * g.in_stride[d];
        p = p * g.out_shape[d] + c[d];
      } else {
        off += c[d] * g.in_stride[d];
      }
    }
  }
  if (interior) { out[gid] = in[off]; return; }
  const int64_t t = tokens[p];
  const int64_t a = t < 0 ? -t : t;
  real v;
  if (a >= XG_TOKEN_FILL_BASE) {
    const int64_t f = a - XG_TOKEN_FILL_BASE;
    v = g.fills[f < g.n_fills ? f : 0];
  } else {
    int64_t k = a - 1;
    const int s = (k >= g.src[0].mapped_size) ? 1 : 0;
    k -= s ? g.src[0].mapped_size : 0;
    const GatherSrc& S = g.src[s];
    int64_t o = 0;
#pragma unroll
    for (int d = 0; d < XG_MAX_NDIM; ++d)
      if (d < g.ndim && !g.mapped[d]) o += c[d] * S.u_stride[d];
    o += gather_mapped_off(S, k);
    v = s ? partner[o] : in[o];
  }
  out[gid] = t < 0 ? -v : v;
}

// k_gather, row-wise: a wave owns 64 aligned 16-B groups of ONE output row; the row's coordinates, its
// interior test over the outer dims, the interior source offset and the token-row base are
// wave-uniform.  Interior cells are narrow loads of consecutive inputs; halo cells decode a token
// (without any division when the mapped dims are the source's trailing dims, the usual
// (time, depth, face, j, i) layouts).
template <int TPW>
__global__ __launch_bounds__(BLOCK) void k_gather_rows(const real* __restrict__ in, const real* __restrict__ partner,
                                                       real* __restrict__ out, const int64_t* __restrict__ tokens,
                                                       GatherGeo g, u32 nrows, FastDiv ntile, int band) {
  u32 lb = blockIdx.x;
  if (band) {  // the workgroups of the launch cut into 8 contiguous bands, one per XCD
    const u32 pb = (gridDim.x + 7) >> 3;
    lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  }
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  if (r >= nrows) return;
  const u32 tile0 = (w - r * ntile.d) * TPW;  // ntile = wave-tasks per row, TPW consecutive tiles each (see k_pad_rows)
  const int nd = g.ndim;
  u32 rem = r;
  bool interior = true;   // over the outer dims
  int64_t off = 0;        // interior source offset of the row
  int64_t uoff[2] = {0, 0};  // unmapped-dim offset of the row in `in` / `partner`
  int64_t prow = 0, pmul = 1;  // token index of the row start (outer mapped dims, row-major)
#pragma unroll
  for (int d = XG_MAX_NDIM - 2; d >= 0; --d) {
    if (d < nd - 1) {
      const u32 q = fdiv(rem, g.out_fd[d]);
      const int64_t cd = (int64_t)(rem - q * g.out_fd[d].d);
      rem = q;
      if (g.mapped[d]) {
        const int64_t ci = cd - g.lo[d];
        interior = interior && ci >= 0 && ci < g.in_shape[d];
        off += ci * g.in_stride[d];
        prow += cd * pmul;
        pmul *= g.out_shape[d];
      } else {
        off += cd * g.in_stride[d];
        uoff[0] += cd * g.src[0].u_stride[d];
        uoff[1] += cd * g.src[1].u_stride[d];
      }
    }
  }
  const int di = nd - 1;
  const int64_t Lo = g.out_shape[di], Li = g.in_shape[di];
  const bool in_mapped = g.mapped[di] != 0;
  const int64_t lo_in = in_mapped ? g.lo[di] : 0;
  auto elem = [&](int64_t x) -> real {  // value of output cell x of this row
    const int64_t ci = x - lo_in;
    if (interior && ci >= 0 && ci < Li) return in[off + ci];
    const int64_t t = tokens[in_mapped ? prow * Lo + x : prow];
    const int64_t a = t < 0 ? -t : t;
    real v;
    if (a >= XG_TOKEN_FILL_BASE) {
      const int64_t f = a - XG_TOKEN_FILL_BASE;
      v = g.fills[f < g.n_fills ? f : 0];
    } else {
      int64_t kk = a - 1;
      const int s = (kk >= g.src[0].mapped_size) ? 1 : 0;
      kk -= s ? g.src[0].mapped_size : 0;
      int64_t o = uoff[s] + gather_mapped_off(g.src[s], kk);
      if (!in_mapped) o += x * g.src[s].u_stride[di];
      v = s ? partner[o] : in[o];
    }
    return t < 0 ? -v : v;
  };
  // the row starts `lead` cells before a 16-B boundary of the output: those and the cells after the
  // last whole group leave as scalars, the groups in between as one 16-B store each
  real* drow = out + (int64_t)r * Lo;
  const int64_t lead = (NV - (int64_t)(((int64_t)r * Lo) % NV)) % NV;
  const int lane = threadIdx.x & 63;
  if (tile0 == 0 && lane == 0)
    for (int64_t x = 0; x < lead && x < Lo; ++x) drow[x] = elem(x);
#pragma unroll
  for (u32 tile = tile0; tile < tile0 + TPW; ++tile) {
  const int64_t x0 = lead + ((int64_t)tile * WAVE + lane) * NV;
  if (x0 >= Lo) break;
  if (x0 + NV <= Lo) {
    dv val;
    const int64_t c0 = x0 - lo_in;
    if (interior && c0 >= 0 && c0 + NV <= Li) {
      // the whole group is interior (all but the groups touching the halo frame): NV consecutive narrow loads, no
      // per-element interior test, no token logic
      const real* sp = in + off + c0;
#pragma unroll
      for (int k = 0; k < NV; ++k) val[k] = sp[k];
    } else {
#pragma unroll
      for (int k = 0; k < NV; ++k) val[k] = elem(x0 + k);
    }
    stg<dv, true>(drow + x0, val);
  } else {
    for (int64_t x = x0; x < Lo; ++x) drow[x] = elem(x);
  }
  }
}

// ------------------------------------------------------------------------------------------
// halo slab -> the halo cells of an array that already holds its interior (Grid.cumsum on a connected axis: the scan
// writes the padded layout in one pass, the halo cells of the cumulative field -- the neighbouring faces' edge values --
// are gathered from that very buffer and put in place here; no padded copy).  One thread per halo cell.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_halo_put(const real* __restrict__ halo, real* __restrict__ out, u64 total, u64 inner,
                                                   u64 n_out, u32 lo, u32 nh) {
  const u64 gid = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (gid >= total) return;
  const u64 x = gid % inner, r = gid / inner;
  const u64 h = r % nh, o = r / nh;
  const u64 j = (h < lo) ? h : n_out - nh + h;  // low halo cells first, then the high ones
  out[(o * n_out + j) * inner + x] = halo[gid];
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int XG_FN(xg_pad)(const real* in, real* out, const int64_t* shape, int ndim, const int64_t* lo, const int64_t* hi,
               const int* bc, const real* fill, const int* order, void* stream) {
  if (!in || !out || !shape || !lo || !hi || !bc) return fail(XG_ERR_INVALID, "NULL argument");
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  PadGeo p;
  memset(&p, 0, sizeof(p));
  p.ndim = ndim;
  int64_t ostride[XG_MAX_NDIM], istride[XG_MAX_NDIM], oshape[XG_MAX_NDIM];
  bool seen[XG_MAX_NDIM] = {false};
  int64_t total = 1, itotal = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    if (lo[d] < 0 || hi[d] < 0) return fail(XG_ERR_INVALID, "negative pad width");
    oshape[d] = shape[d] + lo[d] + hi[d];
    ostride[d] = total;
    istride[d] = itotal;
    total *= oshape[d];
    itotal *= shape[d];
    if ((lo[d] || hi[d]) && (bc[d] < XG_BC_PERIODIC || bc[d] > XG_BC_EXTEND))
      return fail(XG_ERR_INVALID, "axis %d is padded but has no boundary mode", d);
    if ((lo[d] || hi[d]) && shape[d] == 0) return fail(XG_ERR_INVALID, "cannot pad an empty axis");
  }
  for (int t = 0; t < ndim; ++t) {
    int d = order ? order[t] : t;
    if (d < 0 || d >= ndim || seen[d]) return fail(XG_ERR_INVALID, "order is not a permutation");
    seen[d] = true;
    p.out_shape[t] = oshape[d];
    p.out_stride[t] = ostride[d];
    p.in_shape[t] = shape[d];
    p.in_stride[t] = istride[d];
    p.lo[t] = lo[d];
    p.bc[t] = bc[d];
    p.fill[t] = fill ? fill[d] : real(0);
  }
  for (int t = 0; t < ndim; ++t) {
    const int d = order ? order[t] : t;  // memory dim of application step t
    p.mem_step[d] = t;
    p.mem_fd[d] = make_fastdiv((u64)(oshape[d] > 0 ? oshape[d] : 1));
  }
  p.total = total;
  if (total == 0) return XG_OK;
  int rc;
  hipStream_t st = (hipStream_t)stream;
  // long rows: one wave per (row, 64-cell tile) with the row logic on the scalar unit
  const int64_t Lrow = oshape[ndim - 1];
  const int64_t nrows64 = Lrow > 0 ? total / Lrow : 0;
  const bool inner_padded = lo[ndim - 1] != 0 || hi[ndim - 1] != 0;
  if (tune().pad_rows && Lrow >= 64 && nrows64 < 0x7fffffffll && in_stride_inner_is_one(istride, ndim) && aligned16(out)) {
    // aligned rows with an untouched innermost dim: vector copies; everything else: per-element gathers,
    // 16-B stores between the row's first and last 16-B boundary
    const bool straight = !inner_padded && Lrow % NV == 0 && aligned16(in);
    const u64 ntiles = (u64)((Lrow + (NV - 1) + (int64_t)WAVE * NV - 1) / ((int64_t)WAVE * NV));
    const u64 tpw = (u64)(tune().pad_tpw >= 4 ? 4 : tune().pad_tpw >= 2 ? 2 : 1);  // tiles per wave-task
    const u64 nt = (ntiles + tpw - 1) / tpw;
    const u64 waves = (u64)nrows64 * nt;
    if (waves < 0x7fffffffull) {
      const u64 nb = (waves + WPB - 1) / WPB;
      if ((rc = check_grid(nb))) return rc;
      const FastDiv fnt = make_fastdiv(nt);
      const int band = (tune().pad_nt & 4) ? 1 : 0;
      const u32 grid = band ? (u32)(((nb + 7) / 8) * 8) : (u32)nb;
      const int ntf = tune().pad_nt;
#define XG_PR(INNER_, T_) hipLaunchKernelGGL((k_pad_rows<NV, INNER_, T_>), dim3(grid), dim3(BLOCK), 0, st, in, out, p, (u32)nrows64, fnt, ntf)
      if (straight) { if (tpw == 4) XG_PR(false, 4); else if (tpw == 2) XG_PR(false, 2); else XG_PR(false, 1); }
      else { if (tpw == 4) XG_PR(true, 4); else if (tpw == 2) XG_PR(true, 2); else XG_PR(true, 1); }
#undef XG_PR
      XG_LAUNCH_CHECK();
      return XG_OK;
    }
  }
  const u64 nblocks = ((u64)total + BLOCK - 1) / BLOCK;
  if ((rc = check_grid(nblocks))) return rc;
  if (total < 0x7fffffffll) hipLaunchKernelGGL((k_pad<u32>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, p);
  else hipLaunchKernelGGL((k_pad<u64>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, p);
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_gather)(const real* in, const real* partner, real* out, const int64_t* in_shape,
                  const int64_t* partner_shape, const int64_t* out_shape, int ndim, const int* mapped,
                  const int* partner_perm, const int64_t* lo, const int64_t* tokens, int64_t n_tokens,
                  const real* fills, int n_fills, void* stream) {
  if (!in || !out || !in_shape || !out_shape || !mapped || !lo || !tokens) return fail(XG_ERR_INVALID, "NULL argument");
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  if (n_fills < 0 || n_fills > XG_MAX_NDIM || (n_fills > 0 && !fills)) return fail(XG_ERR_INVALID, "bad fill table");
  if (partner && (!partner_shape || !partner_perm)) return fail(XG_ERR_INVALID, "partner without shape/permutation");
  GatherGeo g;
  memset(&g, 0, sizeof(g));
  g.ndim = ndim;
  g.n_fills = n_fills;
  for (int f = 0; f < n_fills; ++f) g.fills[f] = fills[f];
  int64_t total = 1, pmap = 1, istr = 1;
  for (int d = ndim - 1; d >= 0; --d) {
    if (in_shape[d] < 0 || out_shape[d] < 0) return fail(XG_ERR_INVALID, "negative extent");
    g.out_shape[d] = out_shape[d];
    g.out_fd[d] = make_fastdiv((u64)(out_shape[d] > 0 ? out_shape[d] : 1));
    g.in_shape[d] = in_shape[d];
    g.in_stride[d] = istr;
    istr *= in_shape[d];
    g.mapped[d] = mapped[d] ? 1 : 0;
    g.lo[d] = mapped[d] ? lo[d] : 0;
    total *= out_shape[d];
    if (mapped[d]) pmap *= out_shape[d];
    else if (in_shape[d] != out_shape[d]) return fail(XG_ERR_INVALID, "unmapped dim %d changes length", d);
  }
  if (pmap != n_tokens) return fail(XG_ERR_INVALID, "token plane has %lld cells, padded mapped dims have %lld", (long long)n_tokens, (long long)pmap);
  // source 0 = `in` (dims in out order); source 1 = `partner` (its dim k is out dim partner_perm[k])
  for (int s = 0; s < 2; ++s) {
    GatherSrc& S = g.src[s];
    S.mapped_size = 1;
    if (s == 1 && !partner) { S.mapped_size = 0; continue; }
    const int64_t* shp = s ? partner_shape : in_shape;
    int64_t str = 1;
    int64_t strides[XG_MAX_NDIM];
    for (int k = ndim - 1; k >= 0; --k) { strides[k] = str; str *= shp[k]; }
    bool seen[XG_MAX_NDIM] = {false};
    for (int k = 0; k < ndim; ++k) {
      const int d = s ? partner_perm[k] : k;
      if (d < 0 || d >= ndim || seen[d]) return fail(XG_ERR_INVALID, "partner_perm is not a permutation");
      seen[d] = true;
      if (mapped[d]) {
        S.m_extent[S.n_mapped] = shp[k];
        S.m_stride[S.n_mapped] = strides[k];
        S.mapped_size *= shp[k];
        ++S.n_mapped;
      } else {
        if (shp[k] != out_shape[d]) return fail(XG_ERR_INVALID, "source %d: unmapped dim %d has another length", s, d);
        S.u_stride[d] = strides[k];
      }
    }
    // mapped dims trailing in THIS source's own dim order: no unmapped dim after the first mapped one
    S.trailing = 1;
    bool seen_mapped = false;
    for (int k = 0; k < ndim; ++k) {
      const int d = s ? partner_perm[k] : k;
      if (mapped[d]) seen_mapped = true;
      else if (seen_mapped && shp[k] != 1) S.trailing = 0;
    }
  }
  g.total = total;
  if (total == 0) return XG_OK;
  int rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t Lrow = out_shape[ndim - 1];
  const int64_t nrows64 = Lrow > 0 ? total / Lrow : 0;
  if (tune().pad_rows && Lrow >= 64 && aligned16(out) && nrows64 < 0x7fffffffll) {
    const u64 ntiles = (u64)((Lrow + (NV - 1) + (int64_t)WAVE * NV - 1) / ((int64_t)WAVE * NV));
    // tiles per wave-task (see xg_pad): 1 -> 2 tiles 0.641 -> 0.683 on a periodic (Y, X) frame; 4 tiles unrolled spill (22 ms
    // for 1.9): not instantiated (profiles/history/r03at_ab_gather_tpw.jsonl)
    const u64 tpw = (u64)(tune().pad_tpw >= 2 ? 2 : 1);
    const u64 nt = (ntiles + tpw - 1) / tpw;
    const u64 waves = (u64)nrows64 * nt;
    if (waves < 0x7fffffffull) {
      const u64 nb = (waves + WPB - 1) / WPB;
      if ((rc = check_grid(nb))) return rc;
      const int band = (tune().pad_nt & 4) ? 1 : 0;
      const u32 ggrid = band ? (u32)(((nb + 7) / 8) * 8) : (u32)nb;
      if (tpw == 2) hipLaunchKernelGGL(k_gather_rows<2>, dim3(ggrid), dim3(BLOCK), 0, st, in, partner, out, tokens, g, (u32)nrows64, make_fastdiv(nt), band);
      else hipLaunchKernelGGL(k_gather_rows<1>, dim3(ggrid), dim3(BLOCK), 0, st, in, partner, out, tokens, g, (u32)nrows64, make_fastdiv(nt), band);
      XG_LAUNCH_CHECK();
      return XG_OK;
    }
  }
  const u64 nblocks = ((u64)total + BLOCK - 1) / BLOCK;
  if ((rc = check_grid(nblocks))) return rc;
  if (total < 0x7fffffffll) hipLaunchKernelGGL((k_gather<u32>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, partner, out, tokens, g);
  else hipLaunchKernelGGL((k_gather<u64>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, partner, out, tokens, g);
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_halo_put)(const real* halo, real* out, const int64_t* shape, int ndim, int axis, int pad_lo, int pad_hi,
                       void* stream) {
  if (!shape) return fail(XG_ERR_INVALID, "NULL shape");
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  if (axis < 0 || axis >= ndim) return fail(XG_ERR_INVALID, "axis %d out of range for ndim %d", axis, ndim);
  if (pad_lo < 0 || pad_hi < 0) return fail(XG_ERR_INVALID, "negative halo width");
  const int64_t nh = (int64_t)pad_lo + pad_hi;
  if (shape[axis] < nh) return fail(XG_ERR_INVALID, "axis of %lld cells cannot hold %lld halo cells", (long long)shape[axis], (long long)nh);
  int64_t outer = 1, inner = 1;
  for (int d = 0; d < axis; ++d) outer *= shape[d];
  for (int d = axis + 1; d < ndim; ++d) inner *= shape[d];
  const u64 total = (u64)outer * (u64)nh * (u64)inner;
  if (total == 0) return XG_OK;
  if (!halo || !out) return fail(XG_ERR_INVALID, "NULL array argument");
  const u64 nblocks = (total + BLOCK - 1) / BLOCK;
  int rc;
  if ((rc = check_grid(nblocks))) return rc;
  hipLaunchKernelGGL(k_halo_put, dim3((u32)nblocks), dim3(BLOCK), 0, (hipStream_t)stream, halo, out, total, (u64)inner, (u64)shape[axis],
                     (u32)pad_lo, (u32)nh);
  XG_LAUNCH_CHECK();
  return XG_OK;
}

}  // extern "C"
