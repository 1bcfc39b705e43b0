// xg_copy.hip -- strided N-d copies: the data movement AROUND the operators (transposes, flips, broadcasts, slices)
// Part of libxgcm_hip.so; compiled ONCE (type-independent; the float64 pass of the build).
//
// The reference moves data with xarray / numpy views that numpy materialises on demand: `DataArray.transpose` before and
// after a grid ufunc (xgcm/grid_ufunc.py:56-103,885-904), `[..., ::-1]` around the vertical transform of a decreasing
// coordinate (xgcm/transform.py:180-192), `xr.concat` of halo pieces.  Here those are ONE entry point over two stride
// vectors: dst[i0, i1, ...] = src[i0, i1, ...] with element strides per dim on both sides (source strides may be negative
// -- a flip -- or zero -- a broadcast).  Three kernels, chosen on the host after dims of extent 1 are dropped and
// neighbours that are contiguous on BOTH sides are merged:
//   rows       source and destination run along the same dim with unit stride: wave-task = 64 x 16-B groups of one row
//              (rows whose starts are not 16-B aligned on either side: one element per lane; a reversed source row reads
//              descending addresses, still one cache line after the other);
//   transpose  the source's unit-stride dim differs from the destination's: square tiles of 256-B rows through LDS (32 x 32
//              for 8-byte elements, 64 x 64 below; padded rows: no bank conflicts); other dims are batch;
//   gather     anything else (no unit stride on the source side): one element per lane, coalesced on the destination.

#include "xg_common.hpp"

#ifdef XG_PRIMARY

namespace {

constexpr int CMAX = 6;  // dims after merging (callers pass up to XG_MAX_NDIM = 8; more than 6 unmergeable dims: refused)

struct CopyGeo {
  int nd;                 // dims, slowest first; the LAST one is the destination's unit-stride dim
  int64_t shape[CMAX];
  int64_t ss[CMAX];       // source strides (elements)
  int64_t ds[CMAX];       // destination strides (elements)
};

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_copy_rows(const T* __restrict__ src, T* __restrict__ dst, CopyGeo g, u32 ntile,
                                                     u64 ntask, int vec, int nt) {
  // wave-task = (row, tile of 64 lane groups); row = flat index over dims 0 .. nd-2 (wave-uniform: the decode stays on
  // the scalar unit)
  constexpr int V = 16 / (int)sizeof(T);
  const u64 task = banded_wave_id();
  if (task >= ntask) return;
  const u32 lane = threadIdx.x & 63;
  u64 row = task / ntile;
  const u32 tile = (u32)(task - row * ntile);
  int64_t so = 0, dof = 0;
#pragma unroll
  for (int d = CMAX - 2; d >= 0; --d) {
    if (d < g.nd - 1) {
      const u64 q = row / (u64)g.shape[d];
      const int64_t i = (int64_t)(row - q * (u64)g.shape[d]);
      so += i * g.ss[d];
      dof += i * g.ds[d];
      row = q;
    }
  }
  const int64_t n = g.shape[g.nd - 1];
  const int64_t sl = g.ss[g.nd - 1];  // +1, -1 or 0
  if (vec) {  // unit source stride (2: reversed), every row start 16-B aligned on both sides, n a multiple of V
    typedef T tv __attribute__((ext_vector_type(V)));
    const int64_t x = ((int64_t)tile * 64 + lane) * V;
    if (x >= n) return;
    // a reversed row: the group's V elements sit at so - x - (V - 1) .. so - x, loaded as one vector and turned around
    const T* p = vec == 2 ? src + so - x - (V - 1) : src + so + x;
    tv v = nt ? __builtin_nontemporal_load(reinterpret_cast<const tv*>(p)) : *reinterpret_cast<const tv*>(p);
    if (vec == 2) {
      tv w;
#pragma unroll
      for (int k = 0; k < V; ++k) w[k] = v[V - 1 - k];
      v = w;
    }
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<tv*>(dst + dof + x));
    else *reinterpret_cast<tv*>(dst + dof + x) = v;
  } else {
#pragma unroll
    for (int k = 0; k < V; ++k) {  // V passes of 64 consecutive elements: every pass is one coalesced access per side
      const int64_t x = ((int64_t)tile * V + k) * 64 + lane;
      if (x < n) dst[dof + x] = src[so + x * sl];
    }
  }
}

template <typename T, int TS>
__global__ void __launch_bounds__(BLOCK) k_copy_transpose(const T* __restrict__ src, T* __restrict__ dst, CopyGeo g, int t,
                                                          u32 tiles_t, u32 tiles_l, u64 nblk) {
  // block = one TS x TS tile over (dim t: the source's unit-stride dim, last dim: the destination's); the other dims are
  // batch.  TS * sizeof(T) = 256 B: a tile row is two full cache lines on both sides (32 for 8-byte elements, 64 below)
  __shared__ T tile[TS][TS + 1];
  constexpr int R = BLOCK / TS;  // tile rows per pass
  const u32 pb = (gridDim.x + 7) >> 3;
  const u64 b = (u64)((blockIdx.x & 7) * pb + (blockIdx.x >> 3));  // XCD-banded block order
  if (b >= nblk) return;
  u64 r = b;
  const u32 tl = (u32)(r % tiles_l); r /= tiles_l;
  const u32 tt = (u32)(r % tiles_t); r /= tiles_t;
  int64_t so = 0, dof = 0;
#pragma unroll
  for (int d = CMAX - 2; d >= 0; --d) {
    if (d < g.nd - 1 && d != t) {
      const u64 q = r / (u64)g.shape[d];
      const int64_t i = (int64_t)(r - q * (u64)g.shape[d]);
      so += i * g.ss[d];
      dof += i * g.ds[d];
      r = q;
    }
  }
  const int L = g.nd - 1;
  const int64_t nt_ = g.shape[t], nl = g.shape[L];
  const u32 tx = threadIdx.x % TS, ty = threadIdx.x / TS;
  // read: lanes run along dim t (source stride +-1), tile rows along the last dim
#pragma unroll
  for (int k = 0; k < TS / R; ++k) {
    const int64_t it = (int64_t)tt * TS + tx, il = (int64_t)tl * TS + ty + k * R;
    if (it < nt_ && il < nl) tile[ty + k * R][tx] = src[so + it * g.ss[t] + il * g.ss[L]];
  }
  __syncthreads();
  // write: lanes run along the last dim (destination stride 1), tile rows along dim t
#pragma unroll
  for (int k = 0; k < TS / R; ++k) {
    const int64_t il = (int64_t)tl * TS + tx, it = (int64_t)tt * TS + ty + k * R;
    if (it < nt_ && il < nl) dst[dof + it * g.ds[t] + il] = tile[tx][ty + k * R];
  }
}

template <typename T>
__global__ void __launch_bounds__(BLOCK) k_copy_gather(const T* __restrict__ src, T* __restrict__ dst, CopyGeo g, u64 total) {
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  u64 r = i;
  int64_t so = 0, dof = 0;
#pragma unroll
  for (int d = CMAX - 1; d >= 0; --d) {
    if (d < g.nd) {
      const u64 q = r / (u64)g.shape[d];
      const int64_t k = (int64_t)(r - q * (u64)g.shape[d]);
      so += k * g.ss[d];
      dof += k * g.ds[d];
      r = q;
    }
  }
  dst[dof] = src[so];
}

// the same with 16-byte destination groups: the destination's last dim has unit stride and a multiple of V elements, every
// row starts 16-B aligned -- V strided loads, one vector store (4-byte lanes moved 256 B per wave and store: 0.38 of 8 TB/s)
template <typename T>
__global__ void __launch_bounds__(BLOCK) k_copy_gather_vec(const T* __restrict__ src, T* __restrict__ dst, CopyGeo g, u64 groups) {
  constexpr int V = 16 / (int)sizeof(T);
  typedef T tv __attribute__((ext_vector_type(V)));
  const u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= groups) return;
  const int L = g.nd - 1;
  const u64 per_row = (u64)g.shape[L] / V;
  u64 r = i / per_row;
  const int64_t x0 = (int64_t)(i - r * per_row) * V;
  int64_t so = x0 * g.ss[L], dof = x0;
#pragma unroll
  for (int d = CMAX - 2; d >= 0; --d) {
    if (d < L) {
      const u64 q = r / (u64)g.shape[d];
      const int64_t k = (int64_t)(r - q * (u64)g.shape[d]);
      so += k * g.ss[d];
      dof += k * g.ds[d];
      r = q;
    }
  }
  tv v;
#pragma unroll
  for (int k = 0; k < V; ++k) v[k] = src[so + k * g.ss[L]];
  *reinterpret_cast<tv*>(dst + dof) = v;
}

template <typename T>
int copy_launch(const void* src_, void* dst_, const CopyGeo& g, hipStream_t st) {
  const T* src = static_cast<const T*>(src_);
  T* dst = static_cast<T*>(dst_);
  constexpr int V = 16 / (int)sizeof(T);
  const int L = g.nd - 1;
  u64 total = 1;
  for (int d = 0; d < g.nd; ++d) total *= (u64)g.shape[d];
  const bool dst_unit = g.ds[L] == 1;
  const int64_t sl = g.ss[L];
  if (dst_unit && (sl == 1 || sl == -1 || sl == 0)) {
    u64 rows = total / (u64)g.shape[L];
    // (a reversed row is read from its far end: the vector that ends at element 0 must be 16-B aligned)
    const T* first_vec = sl == 1 ? src : src - (V - 1);
    bool vec = (sl == 1 || sl == -1) && sizeof(T) >= 4 && g.shape[L] % V == 0 && (reinterpret_cast<uintptr_t>(first_vec) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int d = 0; d < L && vec; ++d)
      if (g.shape[d] > 1 && (g.ss[d] % V || g.ds[d] % V)) vec = false;
    const u32 ntile = (u32)((g.shape[L] + 64 * V - 1) / (64 * V));
    const u64 ntask = rows * ntile;
    const u64 nblocks = (((ntask + WPB - 1) / WPB + 7) / 8) * 8;
    int rc = check_grid(nblocks);
    if (rc) return rc;
    hipLaunchKernelGGL((k_copy_rows<T>), dim3((u32)nblocks), dim3(BLOCK), 0, st, src, dst, g, ntile, ntask, vec ? (sl == 1 ? 1 : 2) : 0, tune().nt_store ? 1 : 0);
    XG_LAUNCH_CHECK();
    return XG_OK;
  }
  if (dst_unit) {
    int t = -1;
    for (int d = 0; d < L; ++d)
      if ((g.ss[d] == 1 || g.ss[d] == -1) && g.shape[d] >= 8) t = d;
    if (t >= 0) {
      // tile rows of 256 B (64 elements of 4 bytes) measured +18 % over 128 B on full tiles, but a short extent fills wide
      // tiles badly (75 levels: 59 % of two 64-wide tiles, 78 % of three 32-wide ones): pick by the filled fraction
      auto filled = [&](int ts) {
        const double a = (double)g.shape[t] / (double)(((g.shape[t] + ts - 1) / ts) * ts);
        const double b = (double)g.shape[L] / (double)(((g.shape[L] + ts - 1) / ts) * ts);
        return a * b;
      };
      // (8-byte elements keep 32 x 32: with 64 x 64 tiles -- 512-B rows, 33 KB of LDS per workgroup -- (Z,Y,X) -> (Z,X,Y) fell from
      // 0.67 to 0.52 of 8 TB/s, two process pairs on one box, profiles/r06_kernels/r06bj_ab_transpose_tile.log)
      const bool wide = sizeof(T) < 8 && filled(64) * 1.15 >= filled(32);
      const int ts = wide ? 64 : 32;
      const u32 tiles_t = (u32)((g.shape[t] + ts - 1) / ts), tiles_l = (u32)((g.shape[L] + ts - 1) / ts);
      u64 batch = 1;
      for (int d = 0; d < L; ++d)
        if (d != t) batch *= (u64)g.shape[d];
      const u64 nblk = batch * tiles_t * tiles_l;
      const u64 grid = ((nblk + 7) / 8) * 8;
      int rc = check_grid(grid);
      if (rc) return rc;
      if constexpr (sizeof(T) < 8) {
        if (wide) {
          hipLaunchKernelGGL((k_copy_transpose<T, 64>), dim3((u32)grid), dim3(BLOCK), 0, st, src, dst, g, t, tiles_t, tiles_l, nblk);
          XG_LAUNCH_CHECK();
          return XG_OK;
        }
      }
      hipLaunchKernelGGL((k_copy_transpose<T, 32>), dim3((u32)grid), dim3(BLOCK), 0, st, src, dst, g, t, tiles_t, tiles_l, nblk);
      XG_LAUNCH_CHECK();
      return XG_OK;
    }
  }
  bool gvec = dst_unit && sizeof(T) < 8 && g.shape[L] % V == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  for (int d = 0; d < L && gvec; ++d)
    if (g.shape[d] > 1 && g.ds[d] % V) gvec = false;
  if (gvec) {
    const u64 groups = total / V, nb = (groups + BLOCK - 1) / BLOCK;
    int rc = check_grid(nb);
    if (rc) return rc;
    hipLaunchKernelGGL((k_copy_gather_vec<T>), dim3((u32)nb), dim3(BLOCK), 0, st, src, dst, g, groups);
    XG_LAUNCH_CHECK();
    return XG_OK;
  }
  const u64 nblocks = (total + BLOCK - 1) / BLOCK;
  int rc = check_grid(nblocks);
  if (rc) return rc;
  hipLaunchKernelGGL((k_copy_gather<T>), dim3((u32)nblocks), dim3(BLOCK), 0, st, src, dst, g, total);
  XG_LAUNCH_CHECK();
  return XG_OK;
}

}  // namespace

extern "C" {

int xg_copy_nd(const void* src, const int64_t* src_strides, void* dst, const int64_t* dst_strides, const int64_t* shape,
               int ndim, int elem_bytes, void* stream) {
  if (!shape || !src_strides || !dst_strides) return fail(XG_ERR_INVALID, "NULL shape / stride argument");
  if (ndim < 0 || ndim > XG_MAX_NDIM) return fail(XG_ERR_INVALID, "ndim %d not in [0,%d]", ndim, XG_MAX_NDIM);
  if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(XG_ERR_INVALID, "element size %d not 1, 2, 4 or 8", elem_bytes);
  // drop extent-1 dims, refuse negative extents, leave on empty arrays
  int64_t sh[XG_MAX_NDIM], ss[XG_MAX_NDIM], ds[XG_MAX_NDIM];
  int nd = 0;
  for (int d = 0; d < ndim; ++d) {
    if (shape[d] < 0) return fail(XG_ERR_INVALID, "negative extent");
    if (shape[d] == 0) return XG_OK;
    if (shape[d] >= 0x7fffffffll) return fail(XG_ERR_UNSUPPORTED, "extent of 2^31 or more");
    if (shape[d] == 1) continue;  // (whatever stride a dim of one element carries)
    if (dst_strides[d] < 0) return fail(XG_ERR_INVALID, "destination strides must be positive");
    if (dst_strides[d] == 0) return fail(XG_ERR_INVALID, "destination stride 0 on a dim of extent %lld (cells written more than once)", (long long)shape[d]);
    sh[nd] = shape[d]; ss[nd] = src_strides[d]; ds[nd] = dst_strides[d];
    ++nd;
  }
  if (!src || !dst) return fail(XG_ERR_INVALID, "NULL array argument");
  if (nd == 0) { sh[0] = 1; ss[0] = 1; ds[0] = 1; nd = 1; }
  // order the dims by DESTINATION stride, largest first (the destination's unit-stride dim last): any permutation of a
  // contiguous destination is then written in memory order
  for (int i = 1; i < nd; ++i)
    for (int j = i; j > 0 && ds[j - 1] < ds[j]; --j) {
      int64_t t;
      t = sh[j]; sh[j] = sh[j - 1]; sh[j - 1] = t;
      t = ss[j]; ss[j] = ss[j - 1]; ss[j - 1] = t;
      t = ds[j]; ds[j] = ds[j - 1]; ds[j - 1] = t;
    }
  // merge neighbours that are contiguous on both sides: (d, d+1) with stride[d] == stride[d+1] * shape[d+1]
  int m = 0;
  for (int d = 1; d < nd; ++d) {
    if (ss[m] == ss[d] * sh[d] && ds[m] == ds[d] * sh[d] && sh[m] * sh[d] < 0x7fffffffll) {
      sh[m] *= sh[d]; ss[m] = ss[d]; ds[m] = ds[d];
    } else {
      ++m;
      sh[m] = sh[d]; ss[m] = ss[d]; ds[m] = ds[d];
    }
  }
  nd = m + 1;
  if (nd > CMAX) return fail(XG_ERR_UNSUPPORTED, "%d dims remain after merging (at most %d)", nd, CMAX);
  CopyGeo g;
  g.nd = nd;
  for (int d = 0; d < CMAX; ++d) {
    g.shape[d] = d < nd ? sh[d] : 1;
    g.ss[d] = d < nd ? ss[d] : 0;
    g.ds[d] = d < nd ? ds[d] : 0;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (elem_bytes) {
    case 8: return copy_launch<uint64_t>(src, dst, g, st);
    case 4: return copy_launch<uint32_t>(src, dst, g, st);
    case 2: return copy_launch<uint16_t>(src, dst, g, st);
    default: return copy_launch<uint8_t>(src, dst, g, st);
  }
}

}  // extern "C"

#endif  // XG_PRIMARY
