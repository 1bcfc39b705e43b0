// xgcm_hip.hip -- hand-written CDNA4 (gfx950) kernels + C ABI for the xgcm staggered-grid hot path.
//
// Design (see DESIGN.md section 3 for the measurements behind each rule): every op is HBM-bound
// (<= 3 flop per 16 B), so
//  (1) every cell is read once and written once: the boundary halo (periodic / fill / extend) is
//      index arithmetic inside the kernel, never a padded copy; metric multiply / divide ride along;
//  (2) lanes run along the contiguous (last) dimension with 16-byte accesses whatever the op axis
//      is, ONE vector per lane, thread id == linear memory order (a copy written this way streams
//      at 80 % of the 8 TB/s spec; 2-8 tiles per thread or grid-stride loops lose 10-35 %);
//  (3) the set of rows in flight stays compact: along a strided axis a wave register-marches only
//      4 rows (whole columns only when a row is a whole plane, i.e. the Z axis);
//  (4) workgroup b runs on XCD b % 8, each XCD has its own L2: the linear work sequence is cut into
//      8 contiguous bands, one per XCD, so halo-row re-reads and broadcast metrics hit that XCD's
//      L2 ("banding", "z-banding") -- speed only, never correctness;
//  (5) per-item index math is 32-bit with multiply-shift division (FastDiv) and wave-uniform parts
//      on the scalar unit; launches are split on the host so item counts stay below 2^31.
// No MFMA, no LDS tiling of the field (nothing is reused), LDS only for cross-wave scan carries.
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off (bitwise parity with numpy forbids
// FMA contraction of a*m - b*m and reciprocal-based division).

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/xgcm_hip.h"

// The file is compiled twice into the same shared library: once with real = double (exports *_f64
// plus the type-independent helpers) and once with -DXG_F32 (real = float, exports *_f32 only).
#ifdef XG_F32
typedef float real;
#define XG_FN(name) name##_f32
#else
typedef double real;
#define XG_FN(name) name##_f64
#define XG_PRIMARY 1
#endif

// thread-local error text shared by both translation units (hidden: not part of the ABI)
#define XG_ERRBUF_LEN 512
extern "C" __attribute__((visibility("hidden"))) char* xg_internal_errbuf(void);
#ifdef XG_PRIMARY
static thread_local char g_errbuf[XG_ERRBUF_LEN] = {0};
extern "C" __attribute__((visibility("hidden"))) char* xg_internal_errbuf(void) { return g_errbuf; }
#endif

namespace {

constexpr int NV = 16 / (int)sizeof(real);             // elements of a 16-byte lane vector: 2 (f64) / 4 (f32)
typedef real dv __attribute__((ext_vector_type(NV)));  // THE lane vector: every fast path moves 16 B per lane
typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int MAXD = 4;    // coalesced dims on either side of the op axis
constexpr int WAVE = 64;
constexpr int BLOCK = 256; // 4 waves; each wave owns one wave-task
constexpr int WPB = BLOCK / WAVE;

// ------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(xg_internal_errbuf(), XG_ERRBUF_LEN, fmt, ap);
  va_end(ap);
  return code;
}

#define XG_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(XG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                   \
  } while (0)

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// tunables (read once; override with env vars for on-device tuning sessions)
struct Tune {
  int seg;       // rows marched per wave-task along a strided stencil axis
  int nt_store;  // non-temporal stores (+2-4 %; non-temporal LOADS measured -1 % and are not used)
  int seg_max_tiles;  // rows of at most this many 64-lane tiles use the banded short-segment kernel
  int scan_narrow_below; // marching scans with fewer wave-tasks than this use one element per lane
  int pad_rows;          // row-wise generic pad (wave-uniform row logic); 0: one thread per cell
  int transform_lds_kb;  // LDS budget of the cell-major conservative kernel (0: always the register-tile kernel)
  int transform_fast;  // streaming path of the linear transform for well-formed columns (0: always the exact search)
  int zchunk;         // x-tiles per column chunk when the short-segment kernel serves whole-plane rows (0: march)
  int zband;          // band-major row order when all metrics are broadcast along the slowest dim
  int scan_vec;       // aligned-output-group scan for cumsum along the contiguous axis
  int contig_gen;     // pair-wise general path for odd / length-changing rows on the contiguous axis
  int deep_waves;     // marching scans/reductions with fewer wave-tasks than this keep 16 loads in flight
  int march_lds_kb;   // optional dynamic LDS request for the column-marching kernels, only to cap residency
                      // (experiment: +8 % on a bare march in tools/streambench.hip, but -20 % on the real
                      // kernels whose index/metric math then has too few waves to hide behind) => default 0
  Tune() {
    march_lds_kb = env_int("XG_MARCH_LDS_KB", 0);
    contig_gen = env_int("XG_CONTIG_GEN", 1);
    scan_vec = env_int("XG_SCAN_VEC", 1);
    deep_waves = env_int("XG_DEEP_WAVES", 8192);  // neutral on its own, pays together with scan_narrow_below
    zband = env_int("XG_ZBAND", 1);
    zchunk = env_int("XG_ZCHUNK", 256);
    transform_fast = env_int("XG_TRANSFORM_FAST", 1);
    pad_rows = env_int("XG_PAD_ROWS", 1);
    scan_narrow_below = env_int("XG_SCAN_NARROW_BELOW", 8192);
    transform_lds_kb = env_int("XG_TRANSFORM_LDS_KB", 64);
    seg_max_tiles = env_int("XG_SEG_MAX_TILES", 2048);
    seg = env_int("XG_SEG", 1 << 30);  // long march: whole column by default
    nt_store = env_int("XG_NT_STORE", 1);
  }
};
const Tune& tune() {
  static Tune t;
  return t;
}

// ------------------------------------------------------------------------------------------
// geometry: a C-contiguous N-D array seen as (outer..., n, inner...) around the op axis,
// with adjacent dims coalesced whenever every metric's strides allow it.
// ------------------------------------------------------------------------------------------
// Exact u32 division by a launch-time constant without a divide instruction (Granlund-Montgomery
// round-up method): q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(n, m).  On wave-uniform
// operands the compiler keeps all of it on the scalar unit (s_mul_hi_u32).
struct FastDiv {
  u32 d, m, s1, s2;
};
inline FastDiv make_fastdiv(u64 d64) {
  FastDiv f = {1u, 1u, 0u, 0u};
  if (d64 < 1) d64 = 1;
  if (d64 > 0xffffffffull) d64 = 0xffffffffull;  // callers check idx32 before relying on it
  const u32 d = (u32)d64;
  u32 l = 0;
  while ((1ull << l) < (u64)d) ++l;
  f.d = d;
  f.m = (u32)((((1ull << 32) * ((1ull << l) - (u64)d)) / (u64)d) + 1ull);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 1 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ u32 fdiv(u32 n, const FastDiv& f) {
  const u32 t = __umulhi(n, f.m);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// "z-banding": when every metric of a launch is broadcast along the slowest outer dim (a 2-D
// dx(Y,X) weighting a (Z,Y,X) field), rows are visited band by band -- all Z levels of a band of
// B rows before the next band -- so the band's metric values are fetched once and then served by
// the XCD's L2 for the other Z-1 levels instead of being re-read from the Infinity Cache per level.
struct ZBand {
  u32 on, Z, B, Y;       // Y = rows (or segments) per level, B = rows (segments) per band
  FastDiv per_band, fB;  // divisors Z*B and B
};
inline ZBand make_zband(bool on, u64 Z, u64 Y, u32 B) {
  ZBand z;
  memset(&z, 0, sizeof(z));
  z.per_band = make_fastdiv(1);
  z.fB = make_fastdiv(1);
  if (!on || Z < 2 || Z * (u64)B > 0x7fffffffull) return z;
  z.on = 1; z.Z = (u32)Z; z.B = B; z.Y = (u32)Y;
  z.per_band = make_fastdiv(Z * B);
  z.fB = make_fastdiv(B);
  return z;
}
// work index r (band-major) -> (z, y); false if the band's tail row does not exist
__device__ __forceinline__ bool zband_map(const ZBand& zb, u32 r, u32& z, u32& y) {
  const u32 b = fdiv(r, zb.per_band);
  const u32 rem = r - b * zb.per_band.d;
  z = fdiv(rem, zb.fB);
  y = b * zb.B + (rem - z * zb.B);
  return y < zb.Y;
}

// Column chunking for the short-segment kernel when a "row" of the strided axis is a whole plane
// (Z of a (Z,Y,X) field): the x-tiles of a row are cut into chunks of `ch` tiles and the waves
// are ordered (outer, chunk, segment, tile in chunk), so the halo row a segment re-reads was
// loaded `ch` waves earlier by the same XCD (L2 hit) instead of a whole plane earlier (HBM).
struct Chunk {
  u32 on, nchunk;
  FastDiv ch, per_group, fnchunk;  // divisors: tiles per chunk, nseg * ch, chunks per row
};
inline Chunk make_chunk(u64 ntile, u64 nseg, u32 ch) {
  Chunk c;
  memset(&c, 0, sizeof(c));
  c.ch = c.per_group = c.fnchunk = make_fastdiv(1);
  if (ch == 0 || ntile <= ch) return c;
  c.on = 1;
  c.nchunk = (u32)((ntile + ch - 1) / ch);
  c.ch = make_fastdiv(ch);
  c.per_group = make_fastdiv(nseg * ch);
  c.fnchunk = make_fastdiv(c.nchunk);
  return c;
}

struct MIdx {  // element strides of one metric in the coalesced coordinate system
  int64_t outer[MAXD];
  int64_t axis;
  int64_t inner[MAXD];
};

struct Geo {
  int n_outer, n_inner;
  int64_t outer_shape[MAXD];
  int64_t inner_shape[MAXD];
  int64_t outer;  // prod(outer_shape)
  int64_t inner;  // prod(inner_shape)
  int64_t n_in, n_out;
  int idx32;      // outer, inner, n_in, n_out all < 2^32: u32 index math + FastDiv allowed
  FastDiv outer_fd[MAXD], inner_fd[MAXD];
};

// Build Geo (+ up to two MIdx) from the public (shape, ndim, axis, strides) description.
// strides arrays may be NULL (metric absent).  Size-1 dims are dropped, mergeable neighbours
// merged.  Returns 0 or an error.
int build_geo(const int64_t* shape, int ndim, int axis, int64_t n_out, const int64_t* s1,
              const int64_t* s2, Geo* g, MIdx* m1, MIdx* m2) {
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  if (axis < 0 || axis >= ndim) return fail(XG_ERR_INVALID, "axis %d out of range for ndim %d", axis, ndim);
  for (int d = 0; d < ndim; ++d)
    if (shape[d] < 0) return fail(XG_ERR_INVALID, "negative extent");
  memset(g, 0, sizeof(*g));
  if (m1) memset(m1, 0, sizeof(*m1));
  if (m2) memset(m2, 0, sizeof(*m2));
  g->n_in = shape[axis];
  g->n_out = n_out;
  g->outer = 1;
  g->inner = 1;
  if (s1 && m1) m1->axis = s1[axis];
  if (s2 && m2) m2->axis = s2[axis];

  auto group = [&](int lo, int hi, int64_t* gshape, int64_t* st1, int64_t* st2, int* count,
                   int64_t* prod) -> int {
    int n = 0;
    for (int d = lo; d < hi; ++d) {
      if (shape[d] == 1) continue;
      int64_t a = s1 ? s1[d] : 0, b = s2 ? s2[d] : 0;
      if (n > 0) {
        // previous (slower) dim merges with this one iff stride_prev == stride_this * extent_this
        bool ok = (st1[n - 1] == a * shape[d]) && (st2[n - 1] == b * shape[d]);
        if (ok) {
          gshape[n - 1] *= shape[d];
          st1[n - 1] = a;
          st2[n - 1] = b;
          continue;
        }
      }
      if (n == MAXD) return fail(XG_ERR_UNSUPPORTED, "more than %d non-coalescable dims on one side of the axis", MAXD);
      gshape[n] = shape[d];
      st1[n] = a;
      st2[n] = b;
      ++n;
    }
    *count = n;
    *prod = 1;
    for (int i = 0; i < n; ++i) *prod *= gshape[i];
    for (int d = lo; d < hi; ++d)
      if (shape[d] == 0) *prod = 0;
    return 0;
  };
  int64_t o1[MAXD] = {0}, o2[MAXD] = {0}, i1[MAXD] = {0}, i2[MAXD] = {0};
  int rc = group(0, axis, g->outer_shape, o1, o2, &g->n_outer, &g->outer);
  if (rc) return rc;
  rc = group(axis + 1, ndim, g->inner_shape, i1, i2, &g->n_inner, &g->inner);
  if (rc) return rc;
  for (int i = 0; i < MAXD; ++i) {
    if (m1) { m1->outer[i] = o1[i]; m1->inner[i] = i1[i]; }
    if (m2) { m2->outer[i] = o2[i]; m2->inner[i] = i2[i]; }
    g->outer_fd[i] = make_fastdiv(i < g->n_outer ? (u64)g->outer_shape[i] : 1);
    g->inner_fd[i] = make_fastdiv(i < g->n_inner ? (u64)g->inner_shape[i] : 1);
  }
  const int64_t lim = 0xffffffffll;
  g->idx32 = (g->outer <= lim && g->inner <= lim && g->n_in <= lim && g->n_out <= lim) ? 1 : 0;
  return 0;
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
template <int V> struct VecT;
template <> struct VecT<1> { typedef real type; };
template <> struct VecT<NV> { typedef dv type; };

template <typename T, bool NT>
__device__ __forceinline__ T ldg(const real* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
  return *reinterpret_cast<const T*>(p);
}
template <typename T, bool NT>
__device__ __forceinline__ void stg(real* p, T v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
  else *reinterpret_cast<T*>(p) = v;
}

// x-difference of a V-wide lane given the value just left of it
__device__ __forceinline__ dv dvdx_of(dv vc, real vl) {
  dv o;
  o[0] = vc[0] - vl;
#pragma unroll
  for (int k = 1; k < NV; ++k) o[k] = vc[k] - vc[k - 1];
  return o;
}
__device__ __forceinline__ real dvdx_of(real vc, real vl) { return vc - vl; }
// two-point interpolation of a lane vector towards its LEFT neighbour `tl`: (t[k-1] + t[k]) / 2
__device__ __forceinline__ dv interp_left_of(dv tc, real tl) {
  dv o;
  o[0] = (tl + tc[0]) * real(0.5);
#pragma unroll
  for (int k = 1; k < NV; ++k) o[k] = (tc[k - 1] + tc[k]) * real(0.5);
  return o;
}
__device__ __forceinline__ real interp_left_of(real tc, real tl) { return (tl + tc) * real(0.5); }
// forward difference of a lane vector whose RIGHT neighbour is `ur`: (u[k+1] - u[k])
__device__ __forceinline__ dv dudx_fwd(dv uc, real ur) {
  dv o;
#pragma unroll
  for (int k = 0; k < NV - 1; ++k) o[k] = uc[k + 1] - uc[k];
  o[NV - 1] = ur - uc[NV - 1];
  return o;
}
__device__ __forceinline__ real dudx_fwd(real uc, real ur) { return ur - uc; }

// two-point bodies; l = a[..., i], r = a[..., i+1] of the padded array (gridops.py:23-24,76-77,123-175)
template <int OP>
__device__ __forceinline__ real op2(real l, real r) {
  if (OP == XG_OP_DIFF) return r - l;
  if (OP == XG_OP_INTERP) return (l + r) * real(0.5);  // == (l + r) / 2.0 bit for bit
  if (OP == XG_OP_MIN) return (l < r || l != l) ? l : r;  // NaN-propagating like np.min
  return (l > r || l != l) ? l : r;
}
template <int OP> __device__ __forceinline__ dv op2(dv l, dv r) {
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = op2<OP>(l[k], r[k]);
  return o;
}

__device__ __forceinline__ real splat1(real f, real*) { return f; }
__device__ __forceinline__ dv splat1(real f, dv*) {
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = f;
  return o;
}
template <typename T> __device__ __forceinline__ T splat(real f) { return splat1(f, (T*)nullptr); }

// offset of flat outer index `o` in a metric (unrolled so Geo/MIdx stay in SGPRs)
__device__ __forceinline__ int64_t outer_off(const Geo& g, const MIdx& m, int64_t o) {
  int64_t off = 0;
#pragma unroll
  for (int d = MAXD - 1; d >= 0; --d) {
    if (d < g.n_outer) {
      int64_t s = g.outer_shape[d];
      int64_t q = o / s;
      off += (o - q * s) * m.outer[d];
      o = q;
    }
  }
  return off;
}
__device__ __forceinline__ int64_t inner_off(const Geo& g, const MIdx& m, int64_t x) {
  int64_t off = 0;
#pragma unroll
  for (int d = MAXD - 1; d >= 0; --d) {
    if (d < g.n_inner) {
      int64_t s = g.inner_shape[d];
      int64_t q = x / s;
      off += (x - q * s) * m.inner[d];
      x = q;
    }
  }
  return off;
}

// u32 variants (valid when g.idx32): no divide instructions
__device__ __forceinline__ int64_t outer_off32(const Geo& g, const MIdx& m, u32 o) {
  int64_t off = 0;
#pragma unroll
  for (int d = MAXD - 1; d >= 0; --d) {
    if (d < g.n_outer) {
      const u32 q = fdiv(o, g.outer_fd[d]);
      off += (int64_t)(o - q * g.outer_fd[d].d) * m.outer[d];
      o = q;
    }
  }
  return off;
}
__device__ __forceinline__ int64_t inner_off32(const Geo& g, const MIdx& m, u32 x) {
  int64_t off = 0;
#pragma unroll
  for (int d = MAXD - 1; d >= 0; --d) {
    if (d < g.n_inner) {
      const u32 q = fdiv(x, g.inner_fd[d]);
      off += (int64_t)(x - q * g.inner_fd[d].d) * m.inner[d];
      x = q;
    }
  }
  return off;
}
// metric offset + lane step along the coalesced inner dims for a V-wide lane starting at inner index x
// (valid when g.idx32; the single-inner-dim case -- metric varies only along X -- is just a multiply)
__device__ __forceinline__ void inner_off_step32(const Geo& g, const MIdx& m, u32 x, bool pair, int64_t& off, int64_t& step) {
  if (g.n_inner == 1) {
    off = (int64_t)x * m.inner[0];
    step = m.inner[0];
    return;
  }
  off = inner_off32(g, m, x);
  step = pair ? inner_off32(g, m, x + 1) - off : 0;
}
__device__ __forceinline__ int64_t outer_offx(const Geo& g, const MIdx& m, int64_t o) {
  return g.idx32 ? outer_off32(g, m, (u32)o) : outer_off(g, m, o);
}
__device__ __forceinline__ int64_t inner_offx(const Geo& g, const MIdx& m, int64_t x) {
  return g.idx32 ? inner_off32(g, m, (u32)x) : inner_off(g, m, x);
}

// metric value(s) for a V-wide lane at metric offset `off` (second element `step` further on)
template <typename T> __device__ __forceinline__ T ldm(const real* m, int64_t off, int64_t step);
template <> __device__ __forceinline__ real ldm<real>(const real* m, int64_t off, int64_t) { return m[off]; }
template <> __device__ __forceinline__ dv ldm<dv>(const real* m, int64_t off, int64_t step) {
  // metric contiguous along the lanes and 16-B aligned here: one dwordx4 load instead of NV narrow ones
  if (step == 1 && (((reinterpret_cast<uintptr_t>(m) / sizeof(real)) + (uintptr_t)off) & (NV - 1)) == 0)
    return *reinterpret_cast<const dv*>(m + off);
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = m[off + k * step];
  return o;
}

__device__ __forceinline__ u64 wave_id() {
  // uniform per wave; readfirstlane keeps the task decomposition on the scalar unit
  u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return (u64)blockIdx.x * WPB + w;
}

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
// Linear-order stencil kernels.  Measured on MI355X (profiles/r01_streambench_*.txt): a kernel
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
    MIdx mi, const real* __restrict__ m_out, MIdx mo) {
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
    dv a = *reinterpret_cast<const dv*>(prow + i0);
    real n = prow[nidx];
    if (HAS_MI) {
      a = a * ldm<dv>(m_in, mib + (int64_t)i0 * mi.axis, mi.axis);
      n = n * m_in[mib + (int64_t)nidx * mi.axis];
    }
    if (edge && bc == XG_BC_FILL) n = fill;
    if (edge && bc == XG_BC_HALO) n = halo[(row0 + r) * (int64_t)(Lo - Li + 1)];  // one halo cell per row here
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
    stg<dv, NTS>(orow + i0, res);
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
    stg<dv, NTS>(po, res);
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k)
      if (e0 + k < nelem) po[k] = res[k];
  }
}

// ------------------------------------------------------------------------------------------
// K2S: stencil along a STRIDED axis, the small-row case (few x-tiles per row, e.g. Y of a
// (Z,Y,X) field).  Measured on MI355X with random data (profiles/r01_streambench_d_*.txt):
//   * the set of rows in flight must stay compact: each wave register-marches only SEG (= 4)
//     rows -- SEG+1 independent 16-B loads, then SEG stores -- instead of a long segment;
//   * the halo row a segment re-reads must come from the SAME XCD's L2: workgroup b runs on XCD
//     b % 8 (observed dispatch rule, used for speed only), so the linear wave sequence is cut
//     into 8 contiguous bands, one per XCD ("banding").  Each XCD then streams one compact
//     address range and its re-reads never cross the fabric.  6.3 TB/s vs 5.1 TB/s without.
// One wave = one x-tile of one SEG-row segment; the (outer, segment, tile) split and all row
// bases are wave-uniform (scalar unit, FastDiv).
// ------------------------------------------------------------------------------------------
template <int OP, int V, int MET, bool NTS, int SEG>
__global__ __launch_bounds__(BLOCK) void k_stencil_strided_seg(
    const real* __restrict__ in, real* __restrict__ out, Geo g, int64_t o0, u32 nouter, u32 nblk,
    FastDiv ntile, FastDiv nseg, ZBand zb, Chunk ck, int pad_lo, int bc, real fill,
    const real* __restrict__ halo, const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo) {
  typedef typename VecT<V>::type T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  // banding: XCD (b % 8) owns logical blocks [xcd * pb, (xcd + 1) * pb)
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  u32 oo, sg, tile;
  if (ck.on) {  // (outer, chunk, segment, tile in chunk)
    const u32 cg = fdiv(w, ck.per_group);
    const u32 rem = w - cg * ck.per_group.d;
    sg = fdiv(rem, ck.ch);
    oo = fdiv(cg, ck.fnchunk);
    tile = (cg - oo * ck.fnchunk.d) * ck.ch.d + (rem - sg * ck.ch.d);
    if (oo >= nouter || tile >= ntile.d) return;
  } else {
    const u32 r = fdiv(w, ntile);
    tile = w - r * ntile.d;
    if (MET != 0 && zb.on) {  // band-major order over (segment band, outer, segment)
      if (!zband_map(zb, r, oo, sg)) return;
    } else {
      oo = fdiv(r, nseg);
      if (oo >= nouter) return;
      sg = r - oo * nseg.d;
    }
  }
  const int64_t o = o0 + oo;
  const int64_t inner = g.inner;
  const int64_t x = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (x >= inner) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (g.n_out - j0 < SEG) ? g.n_out - j0 : SEG;  // rows this segment really has
  const real* pin = in + (o * g.n_in) * inner + x;
  real* pout = out + (o * g.n_out + j0) * inner + x;

  int64_t mib = 0, mob = 0, mis = 0, mos = 0;  // (host guarantees g.idx32 when metrics are present)
  if (HAS_MI) {
    inner_off_step32(g, mi, (u32)x, V > 1, mib, mis);
    mib += outer_off32(g, mi, (u32)o);
  }
  if (HAS_MO) {
    inner_off_step32(g, mo, (u32)x, V > 1, mob, mos);
    mob += outer_off32(g, mo, (u32)o) + j0 * mo.axis;
  }

  // padded index k = j0 + u  ->  input row q (wave-uniform), fill flag
  T v[SEG + 1];
#pragma unroll
  for (int u = 0; u <= SEG; ++u) {
    int64_t k = j0 + ((u <= nrow) ? u : nrow);  // clamp inside the padded range for short tails
    int64_t q = k - pad_lo;
    bool f = false;
    const real* src = pin;
    if (q < 0 || q >= g.n_in) {
      f = (bc == XG_BC_FILL);
      if (bc == XG_BC_HALO) {  // pre-gathered halo rows, layout (outer, pad_lo + pad_hi, inner)
        src = halo + (o * (g.n_out - g.n_in + 1)) * inner + x;
        q = (q < 0) ? 0 : pad_lo;
      } else {
        q = (q < 0) ? ((bc == XG_BC_PERIODIC) ? g.n_in - 1 : 0) : ((bc == XG_BC_PERIODIC) ? 0 : g.n_in - 1);
      }
    }
    T t = *reinterpret_cast<const T*>(src + q * inner);
    if (HAS_MI) t = t * ldm<T>(m_in, mib + q * mi.axis, mis);
    v[u] = f ? splat<T>(fill) : t;
  }
#pragma unroll
  for (int u = 0; u < SEG; ++u) {
    if (u < nrow) {
      T res = op2<OP>(v[u], v[u + 1]);
      if (HAS_MO) res = res / ldm<T>(m_out, mob + u * mo.axis, mos);
      stg<T, NTS>(pout + u * inner, res);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K5: cumsum along a STRIDED axis (one lane = one column pair, sequential => bit-exact with
// numpy.cumsum / nancumsum), trim/pad table folded into the output index, halo cells written
// from registers at the end of the march.
// ------------------------------------------------------------------------------------------
struct ScanArgs {
  int reverse, skipna, trim_lo, trim_hi, pad_lo, pad_hi, bc;
  real fill;
};

__device__ __forceinline__ real nan0(real v) { return (v != v) ? real(0) : v; }
__device__ __forceinline__ dv nan0(dv v) {
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = nan0(v[k]);
  return o;
}

template <int V, int MET, bool NTL, bool NTS, int U>
__global__ __launch_bounds__(BLOCK) void k_cumsum_strided(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo) {
  typedef typename VecT<V>::type T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  // U independent loads in flight per lane (the scan chain only consumes them)

  const u64 w = wave_id();
  const u32 tile = (u32)(w % ntile);
  const int64_t o = (int64_t)(w / ntile);
  if (o >= g.outer) return;
  const int lane = threadIdx.x & 63;
  const int64_t x = ((int64_t)tile * WAVE + lane) * V;
  if (x >= g.inner) return;
  const int64_t inner = g.inner, n = g.n_in;
  const real* pin = in + (o * n) * inner + x;
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
  auto put = [&](int64_t j, T v) {  // j = output index along the axis
    if (HAS_MO) v = v / ldm<T>(m_out, mo_base + j * mo.axis, mo_step);
    stg<T, NTS>(pout + j * inner, v);
  };

  const int64_t first_kept = a.trim_lo, last_kept = n - 1 - a.trim_hi;  // index space
  const int64_t shift = a.pad_lo - a.trim_lo;
  T acc = splat<T>(real(0)), c_first = splat<T>(real(0)), c_last = splat<T>(real(0));
  bool started = false;
  auto step = [&](int64_t idx, T v) {
    if (HAS_MI) v = v * ldm<T>(m_in, mi_base + idx * mi.axis, mi_step);
    if (a.skipna) v = nan0(v);
    acc = started ? acc + v : v;
    started = true;
    if (idx == first_kept) c_first = acc;
    if (idx == last_kept) c_last = acc;
    if (idx >= first_kept && idx <= last_kept) put(idx + shift, acc);
  };
  int64_t t = 0;
  for (; t + U <= n; t += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t idx = a.reverse ? n - 1 - (t + u) : t + u;
      v[u] = ldg<T, NTL>(pin + idx * inner);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(a.reverse ? n - 1 - (t + u) : t + u, v[u]);
  }
  for (; t < n; ++t) {
    int64_t idx = a.reverse ? n - 1 - t : t;
    step(idx, ldg<T, NTL>(pin + idx * inner));
  }
  // halo cells of the padded cumulative result (xgcm/grid.py:1385-1391; numpy.pad semantics)
  if (a.pad_lo) {
    T h = (a.bc == XG_BC_FILL) ? splat<T>(a.fill) : (a.bc == XG_BC_PERIODIC ? c_last : c_first);
    put(0, h);
  }
  if (a.pad_hi) {
    T h = (a.bc == XG_BC_FILL) ? splat<T>(a.fill) : (a.bc == XG_BC_PERIODIC ? c_first : c_last);
    put(g.n_out - 1, h);
  }
}

// ------------------------------------------------------------------------------------------
// K6: cumsum along the CONTIGUOUS axis: one workgroup per row, chunks of 256 elements in scan
// order, wave-level Hillis-Steele scan with cross-lane shuffles, 4 wave totals through LDS,
// running carry in a register.  Re-associated sum => tolerance parity (not bit-exact).
// Tried and rejected (measured, 3600-long rows): a 1024-thread workgroup per row with 16-B loads
// (2 barrier-separated passes leave too little in flight: 3.2 TB/s) and a barrier-free wave per
// row owning element PAIRS (its two 8-B stores per lane interleave -> half-filled write
// sectors: 2.8 TB/s).  This one-element-per-lane form keeps every load/store instruction a
// contiguous 512 B and runs at 4.5-4.7 TB/s.
// ------------------------------------------------------------------------------------------
template <int MET>
__global__ __launch_bounds__(BLOCK) void k_cumsum_contig(
    const real* __restrict__ in, real* __restrict__ out, Geo g, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo) {
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  __shared__ real wtot[2][WPB];
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t n = g.n_in;
  const real* prow = in + row * n;
  real* orow = out + row * g.n_out;
  int64_t mi_base = 0, mo_base = 0;
  if (HAS_MI) mi_base = outer_off(g, mi, row);
  if (HAS_MO) mo_base = outer_off(g, mo, row);
  const int64_t first_kept = a.trim_lo, last_kept = n - 1 - a.trim_hi;
  const int64_t shift = a.pad_lo - a.trim_lo;
  auto put = [&](int64_t j, real v) {
    if (HAS_MO) v = v / m_out[mo_base + j * mo.axis];
    orow[j] = v;
  };
  auto fetch = [&](int64_t k) -> real {
    real v = real(0);
    if (k < n) {
      const int64_t idx = a.reverse ? n - 1 - k : k;
      v = prow[idx];
      if (HAS_MI) v = v * m_in[mi_base + idx * mi.axis];
      if (a.skipna) v = nan0(v);
    }
    return v;
  };
  real carry = real(0);
  int buf = 0;
  real cur = fetch(tid);
  for (int64_t base = 0; base < n; base += BLOCK, buf ^= 1) {
    const int64_t k = base + tid;
    const int64_t idx = a.reverse ? n - 1 - k : k;
    const real v = cur;
    cur = fetch(k + BLOCK);  // next chunk's load is in flight across this chunk's scan + barrier
    real s = v;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      real t = __shfl_up(s, d, WAVE);
      if (lane >= d) s += t;
    }
    if (lane == WAVE - 1) wtot[buf][wv] = s;
    __syncthreads();
    real woff = real(0), tot = real(0);
#pragma unroll
    for (int i = 0; i < WPB; ++i) {
      real t = wtot[buf][i];
      if (i < wv) woff += t;
      tot += t;
    }
    const real c = carry + (woff + s);
    carry += tot;
    if (k < n) {
      if (idx >= first_kept && idx <= last_kept) put(idx + shift, c);
      if (idx == first_kept) {
        if (a.pad_lo && a.bc == XG_BC_EXTEND) put(0, c);
        if (a.pad_hi && a.bc == XG_BC_PERIODIC) put(g.n_out - 1, c);
      }
      if (idx == last_kept) {
        if (a.pad_lo && a.bc == XG_BC_PERIODIC) put(0, c);
        if (a.pad_hi && a.bc == XG_BC_EXTEND) put(g.n_out - 1, c);
      }
    }
  }
  if (tid == 0 && a.bc == XG_BC_FILL) {
    if (a.pad_lo) put(0, a.fill);
    if (a.pad_hi) put(g.n_out - 1, a.fill);
  }
}

// ------------------------------------------------------------------------------------------
// K6v: cumsum along the CONTIGUOUS axis when output rows are 16-B aligned (n_out % NV == 0, no
// periodic halo).  Threads own aligned groups of NV consecutive OUTPUTS (one 16-B store each) and
// fetch the NV inputs behind them with narrow consecutive loads (input = output index - shift, so
// it may be misaligned: served by L1, the trick that made K1g fast); a thread scans its group,
// waves scan the group totals with shuffles, wave totals go through LDS, the running carry stays
// in a register.  Inputs that map outside the output range (at most one, when the trim is on the
// side the scan starts from) seed the carry.  Re-associated sum => 1e-12 parity like K6.
// ------------------------------------------------------------------------------------------
template <int MET, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_cumsum_contig_vec(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 nrows, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo) {
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  __shared__ real wtot[2][WPB];
  const u32 pb = (nrows + 7) >> 3;
  const u32 row = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);  // XCD banding over rows
  if (row >= nrows) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t n = g.n_in, no = g.n_out;
  const real* prow = in + (int64_t)row * n;
  real* orow = out + (int64_t)row * no;
  int64_t mi_base = 0, mo_base = 0;
  if (HAS_MI) mi_base = outer_off(g, mi, row);
  if (HAS_MO) mo_base = outer_off(g, mo, row);
  const int64_t shift = a.pad_lo - a.trim_lo;
  auto fetch = [&](int64_t idx) -> real {  // weighted, NaN-cleaned input or 0 outside the row
    if (idx < 0 || idx >= n) return real(0);
    real v = prow[idx];
    if (HAS_MI) v = v * m_in[mi_base + idx * mi.axis];
    if (a.skipna) v = nan0(v);
    return v;
  };
  // the one input (if any) that precedes everything in scan order but maps outside [0, no)
  real carry = real(0);
  if (!a.reverse && shift < 0) carry = fetch(0);
  if (a.reverse && (n - 1 + shift) >= no) carry = fetch(n - 1);
  const int64_t groups = no / NV;
  int buf = 0;
  auto group_lo = [&](int64_t t) -> int64_t { return a.reverse ? no - NV * (t + 1) : NV * t; };
  real xn[NV];  // the next pass's inputs are loaded before this pass's scan and barrier
#pragma unroll
  for (int k = 0; k < NV; ++k) xn[k] = (tid < groups) ? fetch(group_lo(tid) + k - shift) : real(0);
  for (int64_t base = 0; base < groups; base += BLOCK, buf ^= 1) {
    const int64_t t = base + tid;
    const bool act = t < groups;
    const int64_t jlo = group_lo(t);
    real x[NV], l[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) x[k] = xn[k];
    {
      const int64_t tn = t + BLOCK;
      const int64_t jn = group_lo(tn);
#pragma unroll
      for (int k = 0; k < NV; ++k) xn[k] = (tn < groups) ? fetch(jn + k - shift) : real(0);
    }
    if (!a.reverse) {
      l[0] = x[0];
#pragma unroll
      for (int k = 1; k < NV; ++k) l[k] = l[k - 1] + x[k];
    } else {
      l[NV - 1] = x[NV - 1];
#pragma unroll
      for (int k = NV - 2; k >= 0; --k) l[k] = l[k + 1] + x[k];
    }
    const real mine = a.reverse ? l[0] : l[NV - 1];
    real s = mine;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      real u = __shfl_up(s, d, WAVE);
      if (lane >= d) s += u;
    }
    real excl = __shfl_up(s, 1, WAVE);
    if (lane == 0) excl = real(0);
    if (lane == WAVE - 1) wtot[buf][wv] = s;
    __syncthreads();
    real woff = real(0), tot = real(0);
#pragma unroll
    for (int i = 0; i < WPB; ++i) {
      real u = wtot[buf][i];
      if (i < wv) woff += u;
      tot += u;
    }
    const real before = carry + (woff + excl);
    carry += tot;
    if (act) {
      dv res;
#pragma unroll
      for (int k = 0; k < NV; ++k) res[k] = before + l[k];
      // halo cells (fill / extend only here): j = 0 and j = no - 1 sit next to a kept cell of the same group
      if (a.pad_lo && jlo == 0) res[0] = (a.bc == XG_BC_FILL) ? (real)a.fill : res[1];
      if (a.pad_hi && jlo + NV == no) res[NV - 1] = (a.bc == XG_BC_FILL) ? (real)a.fill : res[NV - 2];
      if (HAS_MO) res = res / ldm<dv>(m_out, mo_base + jlo * mo.axis, mo.axis);
      stg<dv, NTS>(orow + jlo, res);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K4: weighted sum along a STRIDED axis: one lane per output column pair, sequential in k
// (bit-exact with numpy's reduction over a non-last axis).
// ------------------------------------------------------------------------------------------
template <int V, bool HAS_W, bool NTL, int U>
__global__ __launch_bounds__(BLOCK) void k_reduce_strided(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, int skipna,
    const real* __restrict__ wgt, MIdx mw) {
  typedef typename VecT<V>::type T;
  // U independent loads in flight per lane
  const u64 w = wave_id();
  const u32 tile = (u32)(w % ntile);
  const int64_t o = (int64_t)(w / ntile);
  if (o >= g.outer) return;
  const int lane = threadIdx.x & 63;
  const int64_t x = ((int64_t)tile * WAVE + lane) * V;
  if (x >= g.inner) return;
  const int64_t inner = g.inner, n = g.n_in;
  const real* pin = in + (o * n) * inner + x;
  int64_t mb = 0, ms = 0;
  if (HAS_W) {
    mb = outer_off(g, mw, o) + inner_off(g, mw, x);
    ms = (V > 1) ? inner_off(g, mw, x + 1) - inner_off(g, mw, x) : 0;
  }
  T acc = splat<T>(real(0));
  bool started = false;
  auto step = [&](int64_t k, T v) {
    if (HAS_W) v = v * ldm<T>(wgt, mb + k * mw.axis, ms);
    if (skipna) v = nan0(v);
    acc = started ? acc + v : v;
    started = true;
  };
  int64_t k = 0;
  for (; k + U <= n; k += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ldg<T, NTL>(pin + (k + u) * inner);
#pragma unroll
    for (int u = 0; u < U; ++u) step(k + u, v[u]);
  }
  for (; k < n; ++k) step(k, ldg<T, NTL>(pin + k * inner));
  *reinterpret_cast<T*>(out + o * inner + x) = acc;
}

// K4b: weighted sum along the CONTIGUOUS axis: one wave per row, lane-strided partial sums then
// a shuffle tree (tolerance parity; numpy itself is pairwise here).
template <bool HAS_W, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_reduce_contig(const real* __restrict__ in,
                                                         real* __restrict__ out, Geo g, int skipna,
                                                         const real* __restrict__ wgt, MIdx mw) {
  const u64 row = wave_id();
  if ((int64_t)row >= g.outer) return;
  const int lane = threadIdx.x & 63;
  const int64_t n = g.n_in;
  const real* prow = in + row * n;
  int64_t mb = 0;
  if (HAS_W) mb = outer_off(g, mw, row);
  real acc = real(0);
  int64_t k0 = 0;
  if (VEC) {  // rows 16-B aligned (host): 16-B loads, NV partial sums per lane, two loads in flight
    dv a = splat<dv>(real(0));
    const int64_t nvec = n / NV;
    for (int64_t t = lane; t < nvec; t += WAVE) {
      dv v = *reinterpret_cast<const dv*>(prow + t * NV);
      if (HAS_W) v = v * ldm<dv>(wgt, mb + t * NV * mw.axis, mw.axis);
      if (skipna) v = nan0(v);
      a = a + v;
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) acc += a[c];
    k0 = nvec * NV;
  }
  for (int64_t k = k0 + lane; k < n; k += WAVE) {
    real v = prow[k];
    if (HAS_W) v = v * wgt[mb + k * mw.axis];
    if (skipna) v = nan0(v);
    acc += v;
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) acc += __shfl_down(acc, d, WAVE);
  if (lane == 0) out[row] = acc;
}

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
template <int V, bool INNER>
__global__ __launch_bounds__(BLOCK) void k_pad_rows(const real* __restrict__ in, real* __restrict__ out, PadGeo p,
                                                    u32 nrows, FastDiv ntile) {
  typedef typename VecT<V>::type T;
  const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  if (r >= nrows) return;
  const u32 tile = w - r * ntile.d;
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
  if (V > 1 && !INNER) {  // aligned rows, innermost dim not padded: straight vector copies or fills
    const int64_t x = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
    if (x >= Lo) return;
    T val;
    if (fill_after) val = splat<T>(fv_after);
    else if (fill_before) val = splat<T>(fv_before);
    else val = *reinterpret_cast<const T*>(in + src + x);
    *reinterpret_cast<T*>(drow + x) = val;
  } else if (V > 1) {
    // any row length: the row starts `lead` cells before a 16-B boundary of the output; those cells and
    // the cells after the last whole group go out as scalars (lane 0 of tile 0 / the lane that owns them),
    // everything in between as NV narrow gathers + one 16-B store
    const int64_t lead = (NV - (int64_t)(((int64_t)r * Lo) % NV)) % NV;
    const int lane = threadIdx.x & 63;
    if (tile == 0 && lane == 0)
      for (int64_t xx = 0; xx < lead && xx < Lo; ++xx) drow[xx] = elem(xx);
    const int64_t x = lead + ((int64_t)tile * WAVE + lane) * NV;
    if (x >= Lo) return;
    if (x + NV <= Lo) {
      dv val;
#pragma unroll
      for (int k = 0; k < NV; ++k) val[k] = elem(x + k);
      *reinterpret_cast<dv*>(drow + x) = val;
    } else {
      for (int64_t xx = x; xx < Lo; ++xx) drow[xx] = elem(xx);
    }
  } else {
    const int64_t x = (int64_t)tile * WAVE + (threadIdx.x & 63);
    if (x >= Lo) return;
    drow[x] = elem(x);
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
__global__ __launch_bounds__(BLOCK) void k_gather_rows(const real* __restrict__ in, const real* __restrict__ partner,
                                                       real* __restrict__ out, const int64_t* __restrict__ tokens,
                                                       GatherGeo g, u32 nrows, FastDiv ntile) {
  const u32 w = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  if (r >= nrows) return;
  const u32 tile = w - r * ntile.d;
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
  if (tile == 0 && lane == 0)
    for (int64_t x = 0; x < lead && x < Lo; ++x) drow[x] = elem(x);
  const int64_t x0 = lead + ((int64_t)tile * WAVE + lane) * NV;
  if (x0 >= Lo) return;
  if (x0 + NV <= Lo) {
    dv val;
#pragma unroll
    for (int k = 0; k < NV; ++k) val[k] = elem(x0 + k);
    *reinterpret_cast<dv*>(drow + x0) = val;
  } else {
    for (int64_t x = x0; x < Lo; ++x) drow[x] = elem(x);
  }
}

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

template <bool LOG>
__global__ __launch_bounds__(BLOCK) void k_transform_linear(
    const real* __restrict__ phi, const real* __restrict__ theta, const real* __restrict__ target,
    real* __restrict__ out, Geo g, MIdx mt, MIdx mg, int mask_edges, int bypass_checks, int fast_path) {
  const int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (c >= g.outer * g.inner) return;
  const int64_t inner = g.inner, n = g.n_in, m = g.n_out;
  const int64_t o = c / inner, x = c - o * inner;
  const real* pphi = phi + (o * n) * inner + x;
  const real* pth = theta + outer_off(g, mt, o) + inner_off(g, mt, x);
  const real* ptg = target + outer_off(g, mg, o) + inner_off(g, mg, x);
  real* pout = out + (o * m) * inner + x;
  // theta as the reference sees it: the storage type's log first (np.log in interp_1d_linear)
  auto TH = [&](int64_t k) -> real { real v = pth[k * mt.axis]; return LOG ? xg_log_store(v) : v; };
  auto LEV = [&](int64_t i) -> real { real v = ptg[i * mg.axis]; return LOG ? xg_log_store(v) : v; };

  // ---- fast path: a well-formed column (no NaN, monotonic theta) and non-decreasing, NaN-free
  // targets.  numpy's search then returns max{j: xp[j] <= key} whatever its probing path, so the
  // column is streamed ONCE level by level (coalesced across lanes) while a per-lane cursor walks
  // the targets; any violation met on the way sends the lane to the exact path below, which
  // rewrites every output of the column.
  bool exact = !fast_path || n < 2;
  if (!exact) {
    const real a0 = TH(0), a1 = TH(n - 1);
    if (a0 != a0 || a1 != a1) exact = true;
    else {
      const bool flip = !bypass_checks && (a1 < a0);
      const real tmin = flip ? a1 : a0, tmax = flip ? a0 : a1;  // == nanmin / nanmax once monotonic
      if (bypass_checks && a1 < a0) exact = true;               // decreasing but not flipped: numpy's path decides
      int64_t i = 0;
      double xk = (double)(flip ? a1 : a0);
      double fk = (double)pphi[(flip ? n - 1 : 0) * inner];
      const double lval = fk;
      // the cursor's current target level, loaded once per target and validated on load
      real lev = LEV(0);
      if (lev != lev) exact = true;
      auto emit_and_advance = [&](double res) {
        real r = (real)res;
        if (mask_edges && (lev < tmin || lev > tmax)) r = (real)NAN;
        pout[i * inner] = r;
        ++i;
        if (i < m) {
          const real nxt = LEV(i);
          if (nxt != nxt || nxt < lev) exact = true;
          lev = nxt;
        }
      };
      constexpr int UT = 8;  // levels fetched ahead of use: the column loads do not wait on each other
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
          while (i < m && !exact) {
            const double xv = (double)lev;
            if (!(xv < xk1)) break;          // belongs to a later interval (or to the right edge)
            double res;
            if (xv < xk) res = lval;         // only possible in the first interval: left of the column
            else if (xv == xk) res = fk;
            else res = interp_pair(xv, xk, xk1, fk, fk1);
            emit_and_advance(res);
          }
          xk = xk1; fk = fk1;
        }
      }
      // remaining targets are >= xp[n-1]: the last point itself (fp[n-1]) or right of it (rval == fp[n-1])
      while (i < m && !exact) emit_and_advance(fk);
    }
    if (!exact) return;
  }

  // ---- exact path: numpy's search, probe for probe
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
    pout[i * inner] = r;
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

// ------------------------------------------------------------------------------------------
// broadcasting binary op (out C-contiguous, a/b addressed through strides; dims pre-coalesced)
// ------------------------------------------------------------------------------------------
struct BinGeo {
  int ndim;
  int64_t total;  // number of V-wide items
  int64_t shape[XG_MAX_NDIM];  // shape[ndim-1] counts V-wide items
  int64_t sa[XG_MAX_NDIM], sb[XG_MAX_NDIM];
};

template <int BOP> __device__ __forceinline__ real bin2(real a, real b) {
  if (BOP == XG_BIN_MUL) return a * b;
  if (BOP == XG_BIN_DIV) return a / b;
  if (BOP == XG_BIN_ADD) return a + b;
  return a - b;
}

template <int BOP, int V, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_binary(const real* __restrict__ a, const real* __restrict__ b,
                                                  real* __restrict__ out, BinGeo g, ZBand zb) {
  typedef typename VecT<V>::type T;
  int64_t gid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  if (zb.on) {
    // (Z, P) items with one operand broadcast along Z (da / dx(Y,X)): band-major order keeps the
    // band of the small operand in L2 while all Z levels of the band stream by (as in K1 / K2S)
    u32 z, pin;
    if (gid >= (int64_t)zb.per_band.d * ((zb.Y + zb.B - 1) / zb.B)) return;
    if (!zband_map(zb, (u32)gid, z, pin)) return;
    gid = (int64_t)z * zb.Y + pin;
  }
  if (gid >= g.total) return;
  int64_t r = gid, oa = 0, ob = 0;
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
  const int64_t sa_in = g.sa[g.ndim - 1], sb_in = g.sb[g.ndim - 1];
  if (V > 1) {
    dv av, bv, o;
    if (sa_in == 1) av = *reinterpret_cast<const dv*>(a + oa);
    else {
#pragma unroll
      for (int k = 0; k < NV; ++k) av[k] = a[oa + k * sa_in];
    }
    if (sb_in == 1) bv = *reinterpret_cast<const dv*>(b + ob);
    else {
#pragma unroll
      for (int k = 0; k < NV; ++k) bv[k] = b[ob + k * sb_in];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) o[k] = bin2<BOP>(av[k], bv[k]);
    stg<dv, NTS>(out + gid * NV, o);
  } else {
    stg<real, NTS>(out + gid, bin2<BOP>(a[oa], b[ob]));
  }
}

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

template <int V, bool HAS_AREA, bool NTS, int SEG>
__global__ __launch_bounds__(BLOCK) void k_vorticity(
    const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ area,
    real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny, int64_t nx, FastDiv ntile,
    FastDiv nseg, ZBand zb, int bc_x, real fill_x, int bc_y, real fill_y, AreaIdx ai, int64_t a_sy,
    int64_t a_sx, const real* __restrict__ halo_x, const real* __restrict__ halo_y) {
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
  } else {
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    sg = r - oo * nseg.d;
  }
  const int64_t o = o0 + oo;
  const int64_t a_base = HAS_AREA ? area_outer_off(ai, o) : 0;
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  const real* pu = u + o * ny * nx + i0;
  const real* pv = v + (o * ny + j0) * nx;
  real* po = out + (o * ny + j0) * nx + i0;
  const bool edge = (i0 == 0);
  const int64_t nidx = edge ? ((bc_x == XG_BC_PERIODIC) ? nx - 1 : 0) : i0 - 1;
  const bool fill_edge = edge && (bc_x == XG_BC_FILL);

  T uu[SEG + 1], vv[SEG];
  real vl[SEG];
  {
    int64_t q = j0 - 1;
    bool f = false;
    const real* src = pu + q * nx;
    if (q < 0) {
      f = (bc_y == XG_BC_FILL);
      src = pu + ((bc_y == XG_BC_PERIODIC) ? ny - 1 : 0) * nx;
      if (bc_y == XG_BC_HALO) src = halo_y + o * nx + i0;  // pre-gathered row below the first one: (outer, 1, X)
    }
    T t = *reinterpret_cast<const T*>(src);
    uu[0] = f ? splat<T>(fill_y) : t;
  }
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    const int64_t jr = (s_ < nrow) ? s_ : nrow - 1;  // clamp inside the array for short tails
    uu[s_ + 1] = *reinterpret_cast<const T*>(pu + (j0 + jr) * nx);
    vv[s_] = *reinterpret_cast<const T*>(pv + jr * nx + i0);
    vl[s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[o * ny + j0 + jr]  // pre-gathered column left of the first: (outer, Y, 1)
                                          : pv[jr * nx + nidx];
  }
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    if (s_ < nrow) {
      const real left = fill_edge ? fill_x : vl[s_];
      T z = dvdx_of(vv[s_], left) - (uu[s_ + 1] - uu[s_]);
      if (HAS_AREA) z = z / ldm<T>(area, a_base + (j0 + s_) * a_sy + i0 * a_sx, a_sx);
      stg<T, NTS>(po + s_ * nx, z);
    }
  }
}

// ------------------------------------------------------------------------------------------
// K7b: fused horizontal divergence (delta_x u + delta_y v) / area of docs/ufunc_examples.md
// ("Divergence": u on (Y:center, X:left), v on (Y:left, X:center), both left -> center, i.e.
// padding_width (0,1) on both axes).  Mirror image of K7: SEG rows of u with their right
// neighbour, SEG+1 rows of v (the last one is the upper halo row of the segment).
// ------------------------------------------------------------------------------------------
template <int V, bool HAS_AREA, bool NTS, int SEG>
__global__ __launch_bounds__(BLOCK) void k_divergence(
    const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ area,
    real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny, int64_t nx, FastDiv ntile,
    FastDiv nseg, ZBand zb, int bc_x, real fill_x, int bc_y, real fill_y, AreaIdx ai, int64_t a_sy,
    int64_t a_sx, const real* __restrict__ halo_x, const real* __restrict__ halo_y) {
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
  } else {
    oo = fdiv(r, nseg);
    if (oo >= nouter) return;
    sg = r - oo * nseg.d;
  }
  const int64_t o = o0 + oo;
  const int64_t a_base = HAS_AREA ? area_outer_off(ai, o) : 0;
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * V;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  const real* pu = u + (o * ny + j0) * nx;
  const real* pv = v + o * ny * nx + i0;
  real* po = out + (o * ny + j0) * nx + i0;
  const bool edge = (i0 + V >= nx);
  const int64_t ridx = edge ? ((bc_x == XG_BC_PERIODIC) ? 0 : nx - 1) : i0 + V;
  const bool fill_edge = edge && (bc_x == XG_BC_FILL);

  T uu[SEG], vv[SEG + 1];
  real ur[SEG];
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    const int64_t jr = (s_ < nrow) ? s_ : nrow - 1;
    uu[s_] = *reinterpret_cast<const T*>(pu + jr * nx + i0);
    ur[s_] = (edge && bc_x == XG_BC_HALO) ? halo_x[o * ny + j0 + jr]  // pre-gathered column right of the last: (outer, Y, 1)
                                          : pu[jr * nx + ridx];
    vv[s_] = *reinterpret_cast<const T*>(pv + (j0 + jr) * nx);
  }
  {
    int64_t q = j0 + nrow;  // the row above the segment's last row
    bool f = false;
    const real* src = pv + q * nx;
    if (q >= ny) {
      f = (bc_y == XG_BC_FILL);
      src = pv + ((bc_y == XG_BC_PERIODIC) ? 0 : ny - 1) * nx;
      if (bc_y == XG_BC_HALO) src = halo_y + o * nx + i0;  // pre-gathered row above the last one: (outer, 1, X)
    }
    T t = *reinterpret_cast<const T*>(src);
    vv[SEG] = f ? splat<T>(fill_y) : t;
  }
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    if (s_ < nrow) {
      const real right = fill_edge ? fill_x : ur[s_];
      const T up = (s_ + 1 < nrow) ? vv[s_ + 1] : vv[SEG];
      T z = dudx_fwd(uu[s_], right) + (up - vv[s_]);
      if (HAS_AREA) z = z / ldm<T>(area, a_base + (j0 + s_) * a_sy + i0 * a_sx, a_sx);
      stg<T, NTS>(po + s_ * nx, z);
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
    int64_t mx_sx, const real* __restrict__ my, AreaIdx aiy, int64_t my_sy, int64_t my_sx) {
  typedef typename VecT<V>::type T;
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  const u32 tile = w - r * ntile.d;
  const u32 oo = fdiv(r, nseg);
  if (oo >= nouter) return;
  const u32 sg = r - oo * nseg.d;
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
    if (q < 0) { f = (bc_y == XG_BC_FILL); q = (bc_y == XG_BC_PERIODIC) ? ny - 1 : 0; }
    const T t = *reinterpret_cast<const T*>(pa + q * nx);
    aa[0] = f ? splat<T>(fill_y) : t;
  }
#pragma unroll
  for (int s_ = 0; s_ < SEG; ++s_) {
    const int64_t jr = j0 + ((s_ < nrow) ? s_ : nrow - 1);
    aa[s_ + 1] = *reinterpret_cast<const T*>(pa + jr * nx);
    al[s_] = a[base + jr * nx + nidx];
  }
  const int64_t mxb = (MODE == 0 && mx) ? area_outer_off(aix, o) : 0;
  const int64_t myb = (MODE == 0 && my) ? area_outer_off(aiy, o) : 0;
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
        const T uu = *reinterpret_cast<const T*>(u + base + j * nx + i0);
        const T vv = *reinterpret_cast<const T*>(v + base + j * nx + i0);
        rx = uu * interp_left_of(aa[s_ + 1], left);
        ry = vv * op2<XG_OP_INTERP>(aa[s_], aa[s_ + 1]);
      }
      stg<T, NTS>(out_x + base + j * nx + i0, rx);
      stg<T, NTS>(out_y + base + j * nx + i0, ry);
    }
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
template <int OP, bool NTS, int SEG>
__global__ __launch_bounds__(BLOCK) void k_stencil2d(
    const real* __restrict__ in, real* __restrict__ out, int64_t o0, u32 nouter, u32 nblk, int64_t ny,
    int64_t nx, FastDiv ntile, FastDiv nseg, int order, int plx, int bcx, real fillx, int ply, int bcy,
    real filly) {
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 w = __builtin_amdgcn_readfirstlane(lb * WPB + (threadIdx.x >> 6));
  const u32 r = fdiv(w, ntile);
  const u32 tile = w - r * ntile.d;
  const u32 oo = fdiv(r, nseg);
  if (oo >= nouter) return;
  const u32 sg = r - oo * nseg.d;
  const int64_t o = o0 + oo;
  const int64_t i0 = ((int64_t)tile * WAVE + (threadIdx.x & 63)) * NV;
  if (i0 >= nx) return;
  const int64_t j0 = (int64_t)sg * SEG;
  const int64_t nrow = (ny - j0 < SEG) ? ny - j0 : SEG;
  const real* pin = in + o * ny * nx;
  real* po = out + (o * ny + j0) * nx + i0;

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

  dv pr[SEG + 1];
  real nb[SEG + 1];
  bool rowfill[SEG + 1];
#pragma unroll
  for (int u = 0; u <= SEG; ++u) {
    int64_t k = j0 + ((u <= nrow) ? u : nrow);
    int64_t q = k - ply;
    bool f = false;
    if (q < 0) { f = (bcy == XG_BC_FILL); q = (bcy == XG_BC_PERIODIC) ? ny - 1 : 0; }
    else if (q >= ny) { f = (bcy == XG_BC_FILL); q = (bcy == XG_BC_PERIODIC) ? 0 : ny - 1; }
    rowfill[u] = f;
    pr[u] = *reinterpret_cast<const dv*>(pin + q * nx + i0);
    nb[u] = pin[q * nx + nidx];
  }
  if (order == 0) {  // X first, then Y on the intermediate
    dv tx[SEG + 1];
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      const dv t = opx(pr[u], fill_edge ? fillx : nb[u]);
      tx[u] = rowfill[u] ? splat<dv>(filly) : t;
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u)
      if (u < nrow) stg<dv, NTS>(po + u * nx, op2<OP>(tx[u], tx[u + 1]));
  } else {  // Y first, then X on the intermediate
#pragma unroll
    for (int u = 0; u <= SEG; ++u) {
      if (rowfill[u]) { pr[u] = splat<dv>(filly); nb[u] = filly; }
    }
#pragma unroll
    for (int u = 0; u < SEG; ++u) {
      if (u < nrow) {
        const dv ty = op2<OP>(pr[u], pr[u + 1]);
        const real tn = op2<OP>(nb[u], nb[u + 1]);
        stg<dv, NTS>(po + u * nx, opx(ty, fill_edge ? fillx : tn));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// synthetic fields (splitmix64 finaliser), bit-identical to oracle/refimpl.py:synthetic
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_fill_synthetic(real* __restrict__ out, int64_t n, u64 seed, u64 offset,
                                                          double scale, double shift) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    u64 z = (u64)i + offset + seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    double uu = (double)(z >> 11) * 0x1.0p-53;
    out[i] = (real)(uu * scale + shift);  // formed in f64, rounded once (oracle: synthetic(...).astype(dtype))
  }
}

// ------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }  // lane vector

inline int check_grid(u64 nblocks) {
  if (nblocks == 0 || nblocks > 0x7fffffffull) return fail(XG_ERR_UNSUPPORTED, "launch of %llu blocks exceeds grid limits", nblocks);
  return 0;
}

#define XG_LAUNCH_CHECK()                                                              \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) return fail(XG_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_)); \
  } while (0)

inline unsigned march_lds() {
  int kb = tune().march_lds_kb;
  if (kb < 0) kb = 0;
  if (kb > 160) kb = 160;
  return (unsigned)kb * 1024u;
}

// dispatch on (OP, V, MET, NT) -> template instance
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

// linear-order kernels: split the rows into launches of < 2^31 items
constexpr u64 MAX_ITEMS = 0x7fffff00ull;

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

template <int OP, int V, int MET>
int launch_contig(const StencilCall& c) {
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
  const u32 ZB_ROWS = 16;
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
      hipLaunchKernelGGL((k_stencil_contig<OP, V, MET, true>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)row0, nrows, nblk, fper, zb, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
    else
      hipLaunchKernelGGL((k_stencil_contig<OP, V, MET, false>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)row0, nrows, nblk, fper, zb, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
  }
  return 0;
}

template <int OP, int V, int MET>
int launch_seg(const StencilCall& c) {
  constexpr int SEG = 4;
  const u64 ntile = (u64)((c.g.inner + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 nseg = (u64)((c.g.n_out + SEG - 1) / SEG);
  const Chunk noch = make_chunk(0, 1, 0);
  const Chunk ck = make_chunk(ntile, nseg, ntile > (u64)tune().seg_max_tiles ? (u32)tune().zchunk : 0u);
  const u64 per_outer = ck.on ? (u64)ck.nchunk * ck.ch.d * nseg : ntile * nseg;  // waves per outer index
  if (per_outer > MAX_ITEMS) return launch_march<OP, V, MET>(c);  // (never the case below 2^31 cells per outer index)
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  // z-banding: a single outer dim along which every metric is broadcast, one launch
  const u32 ZB_SEGS = 4;
  const bool zb_ok = !ck.on && MET != 0 && tune().zband && c.g.n_outer == 1 && (!c.m_in || c.mi.outer[0] == 0) &&
                     (!c.m_out || c.mo.outer[0] == 0);
  if (zb_ok) {
    const u64 padded_segs = ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS;
    const u64 waves = padded_segs * (u64)c.g.outer * ntile;
    ZBand zb = make_zband(true, (u64)c.g.outer, nseg, ZB_SEGS);
    if (zb.on && waves <= MAX_ITEMS) {
      const u32 nblk = (u32)((waves + WPB - 1) / WPB);
      const u32 grid = ((nblk + 7) / 8) * 8;
      if (tune().nt_store)
        hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, true, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)0, (u32)c.g.outer, nblk, fnt, fns, zb, noch, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
      else
        hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, false, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, (int64_t)0, (u32)c.g.outer, nblk, fnt, fns, zb, noch, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
      return 0;
    }
  }
  const ZBand zoff = make_zband(false, 0, 0, 1);
  for (int64_t o0 = 0; o0 < c.g.outer; o0 += (int64_t)outer_per) {
    const u32 nouter = (u32)((c.g.outer - o0 < (int64_t)outer_per) ? c.g.outer - o0 : (int64_t)outer_per);
    const u32 nblk = (u32)(((u64)nouter * per_outer + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
    if (tune().nt_store)
      hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, true, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, o0, nouter, nblk, fnt, fns, zoff, ck, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
    else
      hipLaunchKernelGGL((k_stencil_strided_seg<OP, V, MET, false, SEG>), dim3(grid), dim3(BLOCK), 0, c.st, c.in, c.out, c.g, o0, nouter, nblk, fnt, fns, zoff, ck, c.pad_lo, c.bc, c.fill, c.halo, c.m_in, c.mi, c.m_out, c.mo);
  }
  return 0;
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
    default: return stencil_vec<XG_OP_MAX>(V, met, kind, c);
  }
}

inline u32 ceil_div_u32(int64_t a, int64_t b) { return (u32)((a + b - 1) / b); }
inline bool in_stride_inner_is_one(const int64_t* istride, int ndim) { return istride[ndim - 1] == 1; }

// A lane vector of NV elements takes its metric values at a constant step from the first one; that
// holds when the NV elements share one row of the innermost coalesced dim.  For NV == 2 the step is
// computed exactly per lane, so only wider vectors (float) need the innermost extent to divide.
inline bool vec_metric_ok(const Geo& g, bool metrics) {
  if (!metrics || NV <= 2 || g.n_inner == 0) return true;
  return g.inner_shape[g.n_inner - 1] % NV == 0;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

#ifdef XG_PRIMARY
int xg_version(void) { return XG_ABI_VERSION; }

int xg_last_error(char* buf, int n) {
  const char* g_err = xg_internal_errbuf();
  int len = (int)strlen(g_err);
  if (buf && n > 0) {
    int c = len < n - 1 ? len : n - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return len;
}

int xg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int xg_set_device(int device) { XG_HIP(hipSetDevice(device)); return 0; }
int xg_malloc(void** ptr, uint64_t bytes) { XG_HIP(hipMalloc(ptr, bytes)); return 0; }
int xg_free(void* ptr) { XG_HIP(hipFree(ptr)); return 0; }
int xg_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream) {
  XG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
int xg_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream) {
  XG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
int xg_stream_sync(void* stream) { XG_HIP(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int xg_event_create(void** ev) { XG_HIP(hipEventCreate((hipEvent_t*)ev)); return 0; }
int xg_event_record(void* ev, void* stream) { XG_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
int xg_event_elapsed_ms(void* start, void* stop, float* ms) {
  XG_HIP(hipEventSynchronize((hipEvent_t)stop));
  XG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}
int xg_event_destroy(void* ev) { XG_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }
#endif  // XG_PRIMARY

static int stencil1d_impl(int op, const real* in, const real* halo, real* out, const int64_t* shape, int ndim,
                          int axis, int64_t n_out, int pad_lo, int pad_hi, int bc, real fill, const real* m_in,
                          const int64_t* m_in_strides, const real* m_out, const int64_t* m_out_strides,
                          void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if (op < XG_OP_DIFF || op > XG_OP_MAX) return fail(XG_ERR_INVALID, "unknown op %d", op);
  if ((pad_lo | pad_hi) & ~1) return fail(XG_ERR_INVALID, "pad widths must be 0 or 1, got (%d,%d)", pad_lo, pad_hi);
  if (bc < XG_BC_NONE || bc > XG_BC_HALO) return fail(XG_ERR_INVALID, "unknown boundary mode %d", bc);
  if (bc == XG_BC_HALO && !halo) return fail(XG_ERR_INVALID, "XG_BC_HALO without a halo buffer");
  if (bc == XG_BC_HALO && m_in) return fail(XG_ERR_UNSUPPORTED, "pre-gathered halos cannot be combined with an input metric");
  if ((m_in && !m_in_strides) || (m_out && !m_out_strides)) return fail(XG_ERR_INVALID, "metric without strides");
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

int XG_FN(xg_cumsum1d)(const real* in, real* out, const int64_t* shape, int ndim, int axis, int reverse,
                    int skipna, int trim_lo, int trim_hi, int pad_lo, int pad_hi, int bc, real fill,
                    const real* m_in, const int64_t* m_in_strides, const real* m_out,
                    const int64_t* m_out_strides, void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if ((trim_lo | trim_hi | pad_lo | pad_hi) & ~1) return fail(XG_ERR_INVALID, "trim/pad widths must be 0 or 1");
  if (bc < XG_BC_NONE || bc > XG_BC_EXTEND) return fail(XG_ERR_INVALID, "unknown boundary mode %d", bc);
  if ((pad_lo || pad_hi) && bc == XG_BC_NONE) return fail(XG_ERR_INVALID, "halo cells requested but no boundary mode given");
  if ((m_in && !m_in_strides) || (m_out && !m_out_strides)) return fail(XG_ERR_INVALID, "metric without strides");
  if (axis < 0 || axis >= ndim) return fail(XG_ERR_INVALID, "axis out of range");
  const int64_t n = shape[axis];
  const int64_t n_out = n - trim_lo - trim_hi + pad_lo + pad_hi;
  if (n - trim_lo - trim_hi < 1) return fail(XG_ERR_INVALID, "nothing left after trimming (n=%lld)", (long long)n);
  Geo g; MIdx mi, mo;
  int rc = build_geo(shape, ndim, axis, n_out, m_in ? m_in_strides : nullptr, m_out ? m_out_strides : nullptr, &g, &mi, &mo);
  if (rc) return rc;
  if (g.outer == 0 || g.inner == 0) return XG_OK;
  ScanArgs a = {reverse ? 1 : 0, skipna ? 1 : 0, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill};
  const int met = (m_out ? 1 : 0) | (m_in ? 2 : 0);
  hipStream_t st = (hipStream_t)stream;
  if (g.inner == 1) {
    const u64 nblocks = (u64)g.outer;
    if ((rc = check_grid(nblocks + 8))) return rc;
    const bool periodic_halo = (pad_lo || pad_hi) && bc == XG_BC_PERIODIC;
    if (tune().scan_vec && !periodic_halo && n_out % NV == 0 && n_out >= 2 * NV && aligned16(out) && nblocks < 0x7ffffff0ull) {
      const u32 nrows = (u32)nblocks, grid = ((nrows + 7) / 8) * 8;
      const bool nts = tune().nt_store;
#define XG_M(M) do { if (nts) hipLaunchKernelGGL((k_cumsum_contig_vec<M, true>), dim3(grid), dim3(BLOCK), 0, st, in, out, g, nrows, a, m_in, mi, m_out, mo); \
                     else hipLaunchKernelGGL((k_cumsum_contig_vec<M, false>), dim3(grid), dim3(BLOCK), 0, st, in, out, g, nrows, a, m_in, mi, m_out, mo); } while (0)
      switch (met) { case 0: XG_M(0); break; case 1: XG_M(1); break; case 2: XG_M(2); break; default: XG_M(3); }
#undef XG_M
    } else {
#define XG_M(M) hipLaunchKernelGGL((k_cumsum_contig<M>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, a, m_in, mi, m_out, mo)
      switch (met) { case 0: XG_M(0); break; case 1: XG_M(1); break; case 2: XG_M(2); break; default: XG_M(3); }
#undef XG_M
    }
  } else {
    int V = (aligned16(in) && aligned16(out) && (g.inner % NV == 0) && vec_metric_ok(g, met != 0)) ? NV : 1;
    // few, long columns (cumsum along Y of (Z,Y,X): ~2k wave-tasks for 1024 SIMDs): one element per
    // lane doubles (f64) / quadruples (f32) the number of independent marches
    const bool long_march = g.n_in >= 256;
    if (V > 1 && long_march && (u64)ceil_div_u32(g.inner, (int64_t)WAVE * V) * (u64)g.outer < (u64)tune().scan_narrow_below) V = 1;
    const u32 ntile = ceil_div_u32(g.inner, (int64_t)WAVE * V);
    const u64 ntask = (u64)ntile * (u64)g.outer;
    const u64 nblocks = (ntask + WPB - 1) / WPB;
    if ((rc = check_grid(nblocks))) return rc;
    const bool nts = tune().nt_store;
    // few columns, long march (cumsum along Y: ~2k waves for the whole chip): occupancy cannot hide
    // the latency, so keep 16 loads in flight per lane instead of 4 (measured with the narrow lanes
    // above: cumsum along Y f32 48 -> 55 %, sum along Y 59 -> 67 % f32 / 68 -> 71 % f64)
    const bool deep = long_march && ntask < (u64)tune().deep_waves;
#define XG_GO(V_, M, NTS) do { if (deep) hipLaunchKernelGGL((k_cumsum_strided<V_, M, false, NTS, 16>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, a, m_in, mi, m_out, mo); \
                               else hipLaunchKernelGGL((k_cumsum_strided<V_, M, false, NTS, 4>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, a, m_in, mi, m_out, mo); } while (0)
#define XG_M(V_, M) do { if (nts) XG_GO(V_, M, true); else XG_GO(V_, M, false); } while (0)
#define XG_V(V_) switch (met) { case 0: XG_M(V_, 0); break; case 1: XG_M(V_, 1); break; case 2: XG_M(V_, 2); break; default: XG_M(V_, 3); }
    if (V > 1) { XG_V(NV) } else { XG_V(1) }
#undef XG_V
#undef XG_M
#undef XG_GO
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_reduce1d)(const real* in, real* out, const int64_t* shape, int ndim, int axis, int skipna,
                    const real* w, const int64_t* w_strides, void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if (w && !w_strides) return fail(XG_ERR_INVALID, "weight without strides");
  Geo g; MIdx mw;
  int rc = build_geo(shape, ndim, axis, 1, w ? w_strides : nullptr, nullptr, &g, &mw, nullptr);
  if (rc) return rc;
  if (g.outer == 0 || g.inner == 0) return XG_OK;
  hipStream_t st = (hipStream_t)stream;
  if (g.n_in == 0) { XG_HIP(hipMemsetAsync(out, 0, sizeof(real) * g.outer * g.inner, st)); return XG_OK; }
  if (g.inner == 1) {
    const u64 nblocks = ((u64)g.outer + WPB - 1) / WPB;
    if ((rc = check_grid(nblocks))) return rc;
    const bool vec = aligned16(in) && (g.n_in % NV == 0) && g.n_in >= 4 * NV;  // every row starts 16-B aligned
    if (vec) {
      if (w) hipLaunchKernelGGL((k_reduce_contig<true, true>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw);
      else hipLaunchKernelGGL((k_reduce_contig<false, true>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw);
    } else {
      if (w) hipLaunchKernelGGL((k_reduce_contig<true, false>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw);
      else hipLaunchKernelGGL((k_reduce_contig<false, false>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw);
    }
  } else {
    int V = (aligned16(in) && aligned16(out) && (g.inner % NV == 0) && vec_metric_ok(g, w != nullptr)) ? NV : 1;
    const bool long_march = g.n_in >= 256;
    if (V > 1 && long_march && (u64)ceil_div_u32(g.inner, (int64_t)WAVE * V) * (u64)g.outer < (u64)tune().scan_narrow_below) V = 1;
    const u32 ntile = ceil_div_u32(g.inner, (int64_t)WAVE * V);
    const u64 ntask = (u64)ntile * (u64)g.outer;
    const u64 nblocks = (ntask + WPB - 1) / WPB;
    if ((rc = check_grid(nblocks))) return rc;
    const bool deep = long_march && ntask < (u64)tune().deep_waves;
#define XG_GO(V_, W_) do { if (deep) hipLaunchKernelGGL((k_reduce_strided<V_, W_, false, 16>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, skipna, w, mw); \
                           else hipLaunchKernelGGL((k_reduce_strided<V_, W_, false, 4>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, skipna, w, mw); } while (0)
    if (V > 1) { if (w) XG_GO(NV, true); else XG_GO(NV, false); }
    else { if (w) XG_GO(1, true); else XG_GO(1, false); }
#undef XG_GO
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

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
    const u64 nt = (u64)((Lrow + (NV - 1) + (int64_t)WAVE * NV - 1) / ((int64_t)WAVE * NV));
    const u64 waves = (u64)nrows64 * nt;
    if (waves < 0x7fffffffull) {
      const u64 nb = (waves + WPB - 1) / WPB;
      if ((rc = check_grid(nb))) return rc;
      const FastDiv fnt = make_fastdiv(nt);
      if (straight) hipLaunchKernelGGL((k_pad_rows<NV, false>), dim3((u32)nb), dim3(BLOCK), 0, st, in, out, p, (u32)nrows64, fnt);
      else hipLaunchKernelGGL((k_pad_rows<NV, true>), dim3((u32)nb), dim3(BLOCK), 0, st, in, out, p, (u32)nrows64, fnt);
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
    const u64 nt = (u64)((Lrow + (NV - 1) + (int64_t)WAVE * NV - 1) / ((int64_t)WAVE * NV));
    const u64 waves = (u64)nrows64 * nt;
    if (waves < 0x7fffffffull) {
      const u64 nb = (waves + WPB - 1) / WPB;
      if ((rc = check_grid(nb))) return rc;
      hipLaunchKernelGGL(k_gather_rows, dim3((u32)nb), dim3(BLOCK), 0, st, in, partner, out, tokens, g, (u32)nrows64, make_fastdiv(nt));
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
  const u64 nblocks = ((u64)cols + BLOCK - 1) / BLOCK;
  if ((rc = check_grid(nblocks))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int fast = tune().transform_fast;
  if (logarithmic) hipLaunchKernelGGL((k_transform_linear<true>), dim3((u32)nblocks), dim3(BLOCK), 0, st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast);
  else hipLaunchKernelGGL((k_transform_linear<false>), dim3((u32)nblocks), dim3(BLOCK), 0, st, phi, theta, target, out, g, mt, mg, mask_edges, bypass_checks, fast);
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

int XG_FN(xg_binary)(int op, const real* a, const int64_t* a_strides, const real* b, const int64_t* b_strides,
                  real* out, const int64_t* shape, int ndim, void* stream) {
  if (!a || !b || !out || (ndim > 0 && (!shape || !a_strides || !b_strides))) return fail(XG_ERR_INVALID, "NULL argument");
  if (op < XG_BIN_MUL || op > XG_BIN_SUB) return fail(XG_ERR_INVALID, "unknown binary op %d", op);
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
  if ((rc = check_grid(nblocks))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
#define XG_GO(O, V_, NTS) hipLaunchKernelGGL((k_binary<O, V_, NTS>), dim3((u32)nblocks), dim3(BLOCK), 0, st, a, b, out, g, zb)
#define XG_O(O) do { if (V > 1) { if (nts) XG_GO(O, NV, true); else XG_GO(O, NV, false); } else { if (nts) XG_GO(O, 1, true); else XG_GO(O, 1, false); } } while (0)
  switch (op) { case XG_BIN_MUL: XG_O(XG_BIN_MUL); break; case XG_BIN_DIV: XG_O(XG_BIN_DIV); break; case XG_BIN_ADD: XG_O(XG_BIN_ADD); break; default: XG_O(XG_BIN_SUB); }
#undef XG_O
#undef XG_GO
  XG_LAUNCH_CHECK();
  return XG_OK;
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
  constexpr int SEG = 4;
  const u64 ntile = (u64)((nx + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 nseg = (u64)((ny + SEG - 1) / SEG);
  const u64 per_outer = ntile * nseg;
  if (per_outer > MAX_ITEMS) return fail(XG_ERR_UNSUPPORTED, "extent too large for the vorticity kernel");
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
  const u32 ZB_SEGS = 4;
  ZBand zb = make_zband(false, 0, 0, 1);
  u64 outer_step = outer_per;
  if (area && area_bcast_all && tune().zband && outer >= 2) {
    const u64 padded_segs = ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS;
    if (padded_segs * (u64)outer * ntile <= MAX_ITEMS) {
      zb = make_zband(true, (u64)outer, nseg, ZB_SEGS);
      if (zb.on) outer_step = (u64)outer;  // one launch over all levels
    }
  }
  for (int64_t o0 = 0; o0 < outer; o0 += (int64_t)outer_step) {
    const u32 nouter = (u32)((outer - o0 < (int64_t)outer_step) ? outer - o0 : (int64_t)outer_step);
    const u64 waves = zb.on ? ((nseg + ZB_SEGS - 1) / ZB_SEGS) * ZB_SEGS * (u64)outer * ntile : (u64)nouter * per_outer;
    const u32 nblk = (u32)((waves + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
#define XG_GO(V_, A_, NTS) do { if (div) hipLaunchKernelGGL((k_divergence<V_, A_, NTS, SEG>), dim3(grid), dim3(BLOCK), 0, st, u, v, area, out, o0, nouter, nblk, ny, nx, fnt, fns, zb, bc_x, fill_x, bc_y, fill_y, ai, a_sy, a_sx, halo_x, halo_y); \
                                else hipLaunchKernelGGL((k_vorticity<V_, A_, NTS, SEG>), dim3(grid), dim3(BLOCK), 0, st, u, v, area, out, o0, nouter, nblk, ny, nx, fnt, fns, zb, bc_x, fill_x, bc_y, fill_y, ai, a_sy, a_sx, halo_x, halo_y); } while (0)
#define XG_A(V_, A_) do { if (nts) XG_GO(V_, A_, true); else XG_GO(V_, A_, false); } while (0)
    if (V > 1) { if (area) XG_A(NV, true); else XG_A(NV, false); }
    else { if (area) XG_A(1, true); else XG_A(1, false); }
#undef XG_A
#undef XG_GO
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
                       const int64_t* mx_strides, const real* my, const int64_t* my_strides, void* stream) {
  if (!a || !out_x || !out_y || !shape || (mode == 1 && (!u || !v))) return fail(XG_ERR_INVALID, "NULL array argument");
  if (ndim < 2 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [2,%d]", ndim, XG_MAX_NDIM);
  if (bc_x < XG_BC_PERIODIC || bc_x > XG_BC_EXTEND || bc_y < XG_BC_PERIODIC || bc_y > XG_BC_EXTEND)
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
  const int V = al ? NV : 1;
  constexpr int SEG = 4;
  const u64 ntile = (u64)((nx + (int64_t)WAVE * V - 1) / ((int64_t)WAVE * V));
  const u64 nseg = (u64)((ny + SEG - 1) / SEG);
  const u64 per_outer = ntile * nseg;
  if (per_outer > MAX_ITEMS) return fail(XG_ERR_UNSUPPORTED, "extent too large for the fused two-output kernel");
  const FastDiv fnt = make_fastdiv(ntile), fns = make_fastdiv(nseg);
  const u64 outer_per = MAX_ITEMS / per_outer;
  hipStream_t st = (hipStream_t)stream;
  const bool nts = tune().nt_store;
  for (int64_t o0 = 0; o0 < outer; o0 += (int64_t)outer_per) {
    const u32 nouter = (u32)((outer - o0 < (int64_t)outer_per) ? outer - o0 : (int64_t)outer_per);
    const u32 nblk = (u32)(((u64)nouter * per_outer + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
#define XG_GO(V_, M_, NTS) hipLaunchKernelGGL((k_pair2d<V_, M_, NTS, SEG>), dim3(grid), dim3(BLOCK), 0, st, a, u, v, out_x, out_y, o0, nouter, nblk, ny, nx, fnt, fns, bc_x, fill_x, bc_y, fill_y, mx, aix, mx_sy, mx_sx, my, aiy, my_sy, my_sx)
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

int XG_FN(xg_stencil2d)(int op, const real* in, real* out, const int64_t* shape, int ndim, int order,
                     int padx_lo, int padx_hi, int bc_x, real fill_x, int pady_lo, int pady_hi, int bc_y,
                     real fill_y, void* stream) {
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
  if (nx % NV || !aligned16(in) || !aligned16(out)) return fail(XG_ERR_UNSUPPORTED, "fused 2-D stencil needs an X extent that is a multiple of the 16-byte lane vector");
  constexpr int SEG = 4;
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
    const u32 nblk = (u32)(((u64)nouter * per_outer + WPB - 1) / WPB);
    const u32 grid = ((nblk + 7) / 8) * 8;
#define XG_GO(O, NTS) hipLaunchKernelGGL((k_stencil2d<O, NTS, SEG>), dim3(grid), dim3(BLOCK), 0, st, in, out, o0, nouter, nblk, ny, nx, fnt, fns, order, padx_lo, bc_x, fill_x, pady_lo, bc_y, fill_y)
#define XG_O(O) do { if (nts) XG_GO(O, true); else XG_GO(O, false); } while (0)
    switch (op) { case XG_OP_DIFF: XG_O(XG_OP_DIFF); break; case XG_OP_INTERP: XG_O(XG_OP_INTERP); break; case XG_OP_MIN: XG_O(XG_OP_MIN); break; default: XG_O(XG_OP_MAX); }
#undef XG_O
#undef XG_GO
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_fill_synthetic)(real* out, int64_t n, uint64_t seed, uint64_t offset, double scale, double shift, void* stream) {
  if (!out && n > 0) return fail(XG_ERR_INVALID, "NULL output");
  if (n <= 0) return XG_OK;
  u64 nblocks = ((u64)n + BLOCK - 1) / BLOCK;
  if (nblocks > 256ull * 32) nblocks = 256ull * 32;  // grid-stride above 8192 blocks
  hipLaunchKernelGGL(k_fill_synthetic, dim3((u32)nblocks), dim3(BLOCK), 0, (hipStream_t)stream, out, n, (u64)seed, (u64)offset, scale, shift);
  XG_LAUNCH_CHECK();
  return XG_OK;
}

}  // extern "C"
