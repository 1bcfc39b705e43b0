// xg_common.hpp -- shared device / host helpers of the xgcm_amd HIP library (included by every xg_*.hip).
//
// Design (see DESIGN.md section 3 for the measurements behind each rule): every op is HBM-bound
// (<= 3 flop per 16 B), so
//  (1) every cell is read once and written once: the boundary halo (periodic / fill / extend) is
//      index arithmetic inside the kernel, never a padded copy; metric multiply / divide ride along;
//  (2) lanes run along the contiguous (last) dimension with 16-byte accesses whatever the op axis
//      is, ONE vector per lane, thread id == linear memory order (a copy written this way streams
//      at 80 % of the 8 TB/s spec; 2-8 tiles per thread or grid-stride loops lose 10-35 %);
//  (3) the set of rows in flight stays compact: along a strided axis a wave owns ONE row (two in the fused
//      two-component kernels) and re-reads the halo row from its XCD's L2; whole-plane rows (the Z axis) are
//      cut into column chunks; only the scans and reductions, whose sums are sequential by contract, march
//      whole columns;
//  (4) workgroup b runs on XCD b % 8, each XCD has its own L2: the linear work sequence is cut into
//      8 contiguous bands, one per XCD, so halo-row re-reads and broadcast metrics hit that XCD's
//      L2 ("banding", "z-banding") -- speed only, never correctness;
//  (5) per-item index math is 32-bit with multiply-shift division (FastDiv) and wave-uniform parts
//      on the scalar unit; launches are split on the host so item counts stay below 2^31;
//  (6) the instruction stream is budgeted like bandwidth: ~800 SIMD cycles of HBM time per wave-item,
//      4 cycles per wave64 VALU instruction -- inner loops carry no per-element bounds checks.
// No MFMA, no LDS tiling of the field (nothing is reused), LDS only for cross-wave scan carries.
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off (bitwise parity with numpy forbids
// FMA contraction of a*m - b*m and reciprocal-based division).

#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/xgcm_hip.h"

// The file is compiled up to four times into the same shared library: once with real = double (exports *_f64
// plus the type-independent helpers), once with -DXG_F32 (real = float, exports *_f32 only) and -- the units that
// serve integer arrays: stencil, scan, pad, the elementwise binary op -- once with -DXG_I64 -fwrapv (real = int64_t,
// exports *_i64): numpy keeps integer arrays integral through diff / min / max / cumsum / pad (xgcm/gridops.py:23-24,
// 123-126,172-175,227-278; xgcm/padding.py:610-615) and wraps modulo 2^bits, so the same kernels run on two's-complement
// int64 lanes (narrower and unsigned types are widened / narrowed by xg_convert).  Differences of that build: no
// metrics (a metric is a float: numpy promotes before the operator, the host converts first), XG_OP_INTERP returns the
// wrapped SUM l + r (the host halves it after the cast to float64, `(a + b) / 2.0`), NaN handling is a no-op,
// XG_OP_MINU / XG_OP_MAXU compare the lanes as unsigned (uint64 / uint32 arrays).  The units whose integer results keep
// the array's width (stencil, pad, binary -- not the scans / sums, which numpy accumulates in 64 bits) are compiled a
// fourth time with -DXG_I32 -fwrapv (real = int32_t, exports *_i32): int32 / uint32 arrays run on their own 4-byte lanes
// at the float32 byte rate, and the narrower types widen to 4 instead of 8 bytes.
// XG_INT: an integer build (either width); XG_REAL4: a build with 4-byte elements (float or int32).
#if defined(XG_I64)
typedef int64_t real;
typedef uint64_t ureal;
#define XG_FN(name) name##_i64
#define XG_INT 1
#elif defined(XG_I32)
typedef int32_t real;
typedef uint32_t ureal;
#define XG_FN(name) name##_i32
#define XG_INT 1
#define XG_REAL4 1
#elif defined(XG_F32)
typedef float real;
#define XG_FN(name) name##_f32
#define XG_REAL4 1
#else
typedef double real;
#define XG_FN(name) name##_f64
#define XG_PRIMARY 1
#endif

// thread-local error text shared by every translation unit (hidden: not part of the ABI); defined
// once, in xg_runtime.hip
#define XG_ERRBUF_LEN 512
extern "C" __attribute__((visibility("hidden"))) char* xg_internal_errbuf(void);

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

// tunables: ONE table for the whole library (defined in xg_runtime.hip), initialised from XG_* environment
// variables at first use and changeable afterwards through xg_set_tunable() -- A/B measurements interleave
// variants inside one process that way (a device drifts by several percent while it warms up, so variants
// timed one process after the other cannot be compared)
struct Tune {
  int seg;       // rows marched per wave-task along a strided stencil axis
  int nt_store;  // non-temporal stores (+2-4 %)
  int nt_load;   // non-temporal loads of streamed-once inputs (reductions, scans, the contiguous stencil's row)
  int seg_max_tiles;  // rows of at most this many 64-lane tiles use the banded short-segment kernel
  int scan_narrow_below; // marching scans with fewer wave-tasks than this use one element per lane
  int pad_rows;          // row-wise generic pad (wave-uniform row logic); 0: one thread per cell
  int pad_nt;            // row-wise pad: 1 non-temporal stores | 2 non-temporal loads of straight rows | 4 XCD-banded order
  int transform_lds_kb;  // LDS budget of the cell-major conservative kernel (0: always the register-tile kernel)
  int transform_win;   // conservative transform: sliding window of accumulators in LDS (K9d) for up to 64 bins
  int transform_fast;  // streaming path of the linear transform for well-formed columns (0: always the exact search)
  int transform_stage; // linear transform: 3 = ring of output rows in LDS, written out as complete rows (2: only the
                       // level table in LDS; 1: whole-column tile; 0: direct stores)
  int transform_ring;  // rows of that ring (8 / 16 / 32)
  int transform_cwin;  // conservative transform: accumulators per lane in the wave's LDS window (8 / 16)
  int zchunk;         // x-tiles per column chunk when the short-segment kernel serves whole-plane rows (0: march)
  int zband;          // band-major row order when all metrics are broadcast along the slowest dim
  int zb_rows;        // rows per band
  int scan_block;     // workgroup size of the contiguous-axis scan (64 / 128 / 256 / 512 / 1024; 0: 256, 128 for float32 rows)
  int strided_gen;    // flat NV-group kernel for misaligned rows of a strided stencil axis
  int march_band;     // XCD-banded wave order in the column-marching scans / reductions
  int scan_vec;       // aligned-output-group scan for cumsum along the contiguous axis
  int scan_dpp;       // its wave scan through DPP row shifts / broadcasts instead of __shfl_up
  int contig_gen;     // pair-wise general path for odd / length-changing rows on the contiguous axis
  int deep_waves;     // marching scans/reductions with fewer wave-tasks than this keep 16 loads in flight
  int contig_rw;      // rows per wave-task of the row-wave contiguous-axis metric kernel K1r (0: flat K1)
  int rw_zshare;      // K1r: the rows of a wave-task are one row of consecutive LEVELS sharing the metric vector
  int met_zk;         // K2S with two metrics, z-banded: outer levels per wave-task sharing the metric rows (1 / 2 / 4)
  int met_zk1;        // the same with ONE metric (derivative: a divisor only)
  int vec_zk;         // fused vorticity / divergence with an area, z-banded: levels per wave-task sharing the area rows
  int seg_ys;         // plain strided-axis stencil with short rows: y-stacked workgroups, the lower row of a pair through LDS (K2Sy)
  int met_ys;         // two-axis kernel with metrics, X first: y-stacked workgroups, the shared intermediate row through LDS (K8y)
  int met_zk2;        // two-axis kernel with metrics: levels per wave-task sharing the three metric planes' rows (2 / 4)
  int nb_dpp;         // contiguous-axis stencils: the value beside a lane's vector from the neighbouring lane (DPP) instead of an 8-byte load
  int vec_zb_rows;    // fused vorticity / divergence with an area: rows per band (the area rows of a band live in the XCD's L2)
  int vec_nt;         // fused vorticity: non-temporal loads of bit 0 the v rows (left neighbour by lane shuffle), bit 1 the inner u rows
  int contig_rw_mi;   // K1r rows per wave-task when an input metric rides along too (three metric loads per row)
  int met_seg;        // rows per wave-task of the strided-axis kernel K2S with two metrics (1 / 2 / 4)
  int met_seg1;       // the same with ONE metric
  int met_scalar;     // K2S: metrics that do not vary along the lanes (drF(Z)) through scalar loads
  int scan_pipe;      // rolling-window loads in the marching scans / reductions (0: batches of U; 2: short marches too)
  int scan_u;         // loads in flight per lane of a long march (8 / 16 / 24 / 32)
  int scan_pace;      // experiment: workgroup barrier per window in the pipelined marching scan
  int scan_chain;     // long strided-axis scans as a chained flat launch (K5c); 2: whenever the march has >= 2 chunks
  int scan_chain_w;   // K5c: levels' worth of columns that advance side by side inside one XCD band
  int scan_chain_tmaj; // chained kernels with a metric shared by the outer indices: columns numbered x-tile-major, a sub-band = all levels of a few x-tiles
  int scan_chain_spin; // polls of a hand-off slot before a chunk gives up (the launch is then redone by the march, in stream)
  int met_ys1, met_ys2; // K2Sm: strided-axis metric stencils with y-stacked workgroups, 10 * rows + levels per wave (one / two metrics; 0: K2S)
  int transform_lean; // linear transform, ring + shared level table: the lean streaming loop (targets validated once, 32-bit cursors, pointer-stepped columns)
  int pad_tpw;        // row-wise pad: consecutive 64-lane tiles of a row per wave-task (the row logic is paid once per task)
  int bin_idx32;      // elementwise binary op: index decomposition with 32-bit multiply-shift divisions (0: 64-bit divisions)
  int reduce_ldsw_u;  // K4L: rows per block (8 / 16)
  int march_ofast;    // marching weighted reductions whose weights are shared by the outer indices: outer indices fastest in the work order
  int reduce_sk;      // contiguous-axis reductions: kernels specialised for the plain sums (skipna False / True)
  int reduce_ru;      // contiguous-axis reductions: 4 independent 16-B loads per lane before the first addition
  int reduce_wfast;   // contiguous-axis weighted reductions: unit-stride, row-aligned weights as one vector load per lane
  int reduce_wg;      // contiguous-axis reductions: one WORKGROUP per row (its 4 waves read adjacent 1-KB pieces) instead of one wave
  int scan_sh1;       // contiguous-axis scan, 4-byte elements, inputs shifted by one cell: aligned vector + one narrow load
  int reduce_zmarch;  // K4Z: long weighted reductions with level-shared weights as a march of ZL levels per wave (100 * ZL + U: 312 = 3 levels, 12 row steps in flight; + 1000: float32 too -- the default; 0: K4L)
  int reduce_ldsw;    // K4L: long weighted reductions with level-shared weights as a march whose weight rows go through LDS once per workgroup (0: chained K4cz)
  int reduce_zl;      // K4cz: outer indices per task of the chained weighted reduction sharing the weight rows (1 / 2 / 4)
  int dbg;            // A/B switches that do NOT change results (bit 2: the linear transform's division inside its loop)
  int march_lds_kb;   // optional dynamic LDS request for the column-marching kernels, only to cap residency
                      // (experiment: +8 % on a bare march in tools/probes/streambench.hip, but -20 % on the real
                      // kernels whose index/metric math then has too few waves to hide behind) => default 0
};
extern "C" __attribute__((visibility("hidden"))) Tune* xg_internal_tune(void);
inline const Tune& tune() { return *xg_internal_tune(); }

// workspace of the chained scans / reductions (K5c, K4c: xg_scan.hip), one per (device, stream), owned by xg_runtime.hip:
//   slots   `slot_bytes` of running-sum slots, all zero between launches (the kernels clean up after themselves);
//   ticket  8 ticket counters, 128 B apart (same);
//   poison  two device words of the stream: [0] set by a wave that gives up waiting for its predecessor chunk (it
//           passes NaN on with a valid epoch, so its column is visibly poisoned and nobody waits behind it), [1] the
//           rescue kernel's count of finished workgroups.  EVERY chained launch is followed, on the same stream, by its
//           marching twin with poison[0] as its run-if word: zero (always, in practice) and every workgroup leaves at
//           once; non-zero and it redoes the whole call from the untouched inputs, scrubs the slots and clears the word.
//           A damaged result therefore never reaches a consumer, host or device, eager or under graph replay;
//   gave_up two host-mapped words for reporting only: [0] sticky, some wave has given up (the library then plans
//           marches until xg_chain_rearm()), [1] number of launches the rescue kernel has redone (xg_chain_status()).
// xg_internal_chain_ok(): 1 when workgroups whose ids agree modulo 8 share an XCD on this device (probed once) -- the
// chain passes its sums through that XCD's L2 -- and no wave has given up since the last re-arm.
struct ChainWs { void* slots; u64 slot_bytes; u32* ticket; u32* gave_up; u32* poison; };
extern "C" __attribute__((visibility("hidden"))) int xg_internal_chain_ws(void* stream, u64 slot_bytes, ChainWs* ws);
extern "C" __attribute__((visibility("hidden"))) int xg_internal_chain_ok(void);

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
// "faces" (round 4): the metrics of a (Z, face, Y, X) field -- MITgcm's LLC / cubed-sphere layout -- change from face to
// face and are shared by the levels only: two outer dims, the metric broadcast along the SLOWER one.  The banded index y
// then runs over nf x (rows or segments of one face) -- the (face, Y) block of a level is contiguous -- and the kernels
// split it again (zband_face); the levels a wave-task shares its metric rows across are `nf` outer indices apart.
struct ZBand {
  u32 on, Z, B, Y;       // Y = rows (or segments) per level, B = rows (segments) per band
  FastDiv per_band, fB;  // divisors Z*B and B
  u32 nf;                // faces under each level (1: none)
  FastDiv fper;          // rows (segments) per face
};
inline ZBand make_zband(bool on, u64 Z, u64 Y, u32 B, u64 nf = 1) {
  ZBand z;
  memset(&z, 0, sizeof(z));
  z.per_band = make_fastdiv(1);
  z.fB = make_fastdiv(1);
  z.fper = make_fastdiv(1);
  z.nf = 1;
  if (!on || Z < 2 || Z * (u64)B > 0x7fffffffull || nf < 1 || nf > 0xffffull || Y % nf) return z;
  z.on = 1; z.Z = (u32)Z; z.B = B; z.Y = (u32)Y;
  z.per_band = make_fastdiv(Z * B);
  z.fB = make_fastdiv(B);
  z.nf = (u32)nf;
  z.fper = make_fastdiv(Y / nf);
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

// (level group, banded row index over nf faces) -> first flattened outer index of the task, the step between its
// levels, the row index inside the face; `zl` = first level of the group
__device__ __forceinline__ void zband_face(const ZBand& zb, u32 zl, u32& y, u32& first, u32& step) {
  if (zb.nf <= 1) { first = zl; step = 1; return; }
  const u32 f = fdiv(y, zb.fper);
  y -= f * zb.fper.d;
  first = zl * zb.nf + f;
  step = zb.nf;
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
// the 8-byte lane of the long marches (few, long columns): one double, or TWO floats -- a float march with one
// element per lane moves 256 B per wave and row, too narrow for the memory system (sum along Y f32: 55 % against
// 80 % for f64); HV elements keep the bytes per wave-row the same in both builds
constexpr int HV = 8 / (int)sizeof(real);
#ifdef XG_REAL4
typedef real hv __attribute__((ext_vector_type(2)));
template <> struct VecT<2> { typedef hv type; };
#endif

template <typename T, bool NT>
__device__ __forceinline__ T ldg(const real* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
  return *reinterpret_cast<const T*>(p);
}
template <typename T> __device__ __forceinline__ void stg_drop(real* p, T v);
template <typename T, bool NT>
__device__ __forceinline__ void stg(real* p, T v) {
#ifdef XG_STG_DROP_ALL  // experiment builds only: every 16-B non-temporal store of the unit with `sc1 nt`
  if constexpr (NT && sizeof(T) == 16) { stg_drop<T>(p, v); return; }
#endif
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
  else *reinterpret_cast<T*>(p) = v;
}
// the store of a kernel that measured faster with dropped output lines (16-B lane vectors; narrower stores stay `nt`)
template <typename T, bool NT>
__device__ __forceinline__ void stg_s(real* p, T v) {
  if constexpr (NT && sizeof(T) == 16) stg_drop<T>(p, v);
  else stg<T, NT>(p, v);
}
// 16-B store with `sc1 nt`: written through AND dropped from the XCD's L2 (plain `nt` keeps the line).  For outputs that
// nothing in the launch reads again, in kernels whose L2 has better things to hold; measured per kernel (see the callers).
// Inline asm ends with `s_nop 1`: the compiler pads nothing after an asm store whose source registers it may overwrite next.
template <typename T>
__device__ __forceinline__ void stg_drop(real* p, T v) {
  static_assert(sizeof(T) == 16, "16-byte lane vectors only");
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
  const u32x4_ w = __builtin_bit_cast(u32x4_, v);
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}

// what a chained scan passes on after giving up on a hand-off (the launch is redone by the rescue kernel either way)
__device__ __forceinline__ real poison_value() {
#ifdef XG_INT
  return real(0x7ff8dead7ff8deadll);
#else
  return real(__builtin_nan(""));
#endif
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
#ifndef XG_INT
__device__ __forceinline__ dv interp_left_of(dv tc, real tl) {
  dv o;
  o[0] = (tl + tc[0]) * real(0.5);
#pragma unroll
  for (int k = 1; k < NV; ++k) o[k] = (tc[k - 1] + tc[k]) * real(0.5);
  return o;
}
__device__ __forceinline__ real interp_left_of(real tc, real tl) { return (tl + tc) * real(0.5); }
#endif
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
#ifdef XG_INT
  if (OP == XG_OP_INTERP) return l + r;  // the wrapped sum; halved by the host after the conversion to float64
  if (OP == XG_OP_MIN) return l < r ? l : r;
  if (OP == XG_OP_MINU) return (ureal)l < (ureal)r ? l : r;  // the lanes of an unsigned array
  if (OP == XG_OP_MAXU) return (ureal)l > (ureal)r ? l : r;
  return l > r ? l : r;
#else
  if (OP == XG_OP_INTERP) return (l + r) * real(0.5);  // == (l + r) / 2.0 bit for bit
  if (OP == XG_OP_MIN) return (l < r || l != l) ? l : r;  // NaN-propagating like np.min
  return (l > r || l != l) ? l : r;
#endif
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
#ifdef XG_REAL4
__device__ __forceinline__ hv splat1(real f, hv*) { hv o; o[0] = f; o[1] = f; return o; }
#endif
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

// Is the metric of EVERY active lane of this wave one aligned V-vector in every row (elements side by side, row 0 aligned,
// row step a multiple of V)?  Kernels whose row loops load metrics ask once and run the loop with plain vector loads: the
// test inside `ldm` is a lane-divergent branch per load that cuts an unrolled row loop into fragments, every product then
// waits for its own metric load (float32's two-element lanes: cumint Y 0.59 against cumsum Y's 0.74 of 8 TB/s).
template <int V>
__device__ __forceinline__ bool met_vec_all(const real* m, int64_t base, int64_t step, int64_t axis) {
  if (V == 1) return false;
  const bool ok = step == 1 && (axis % V) == 0 && ((reinterpret_cast<uintptr_t>(m) / sizeof(real) + (uintptr_t)base) % V) == 0;
  return __all(ok) != 0;
}
// metric value(s) for a V-wide lane at metric offset `off` (second element `step` further on)
template <typename T> __device__ __forceinline__ T ldm(const real* m, int64_t off, int64_t step);
template <> __device__ __forceinline__ real ldm<real>(const real* m, int64_t off, int64_t) { return m[off]; }
#ifdef XG_REAL4
template <> __device__ __forceinline__ hv ldm<hv>(const real* m, int64_t off, int64_t step) {
  // metric contiguous along the lanes and 8-B aligned here: one dwordx2 load instead of two narrow ones
  if (step == 1 && ((reinterpret_cast<uintptr_t>(m) / sizeof(real) + (uintptr_t)off) & 1) == 0)
    return *reinterpret_cast<const hv*>(m + off);
  hv o;
  o[0] = m[off];
  o[1] = m[off + step];
  return o;
}
#endif
template <> __device__ __forceinline__ dv ldm<dv>(const real* m, int64_t off, int64_t step) {
  // metric contiguous along the lanes and 16-B aligned here: one dwordx4 load instead of NV narrow ones
  if (step == 1 && (((reinterpret_cast<uintptr_t>(m) / sizeof(real)) + (uintptr_t)off) & (NV - 1)) == 0)
    return *reinterpret_cast<const dv*>(m + off);
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = m[off + k * step];
  return o;
}

// The value the NEXT-LOWER lane (wave_shr:1) / NEXT-HIGHER lane (wave_shl:1) holds, moved on the VALU data path (DPP).
// A stencil along the contiguous axis needs, next to its own 16-B vector, the one element left / right of it: that is
// the neighbouring lane's last / first element -- already in registers -- instead of a second, 8-byte load per row whose
// 64 addresses walk over the same cache lines again.  Lane 0 (shr) / lane 63 (shl) have no source lane and read 0: the
// caller loads that one value itself.  Source lanes must be active.
template <int CTRL>
__device__ __forceinline__ real dpp_lane(real v) {
#ifdef XG_REAL4
  return __builtin_bit_cast(real, (u32)__builtin_amdgcn_update_dpp(0, (int)__builtin_bit_cast(u32, v), CTRL, 0xf, 0xf, false));
#else
  const u64 b = __builtin_bit_cast(u64, v);
  const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, CTRL, 0xf, 0xf, false);
  const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(real, (u64)lo | ((u64)hi << 32));
#endif
}
__device__ __forceinline__ real from_lane_below(real v) { return dpp_lane<0x138>(v); }  // wave_shr:1
__device__ __forceinline__ real from_lane_above(real v) { return dpp_lane<0x130>(v); }  // wave_shl:1

// the same with the workgroups of a launch cut into 8 contiguous bands, one per XCD (grid size = multiple of 8;
// wave ids beyond the work are rejected by the caller's range check)
__device__ __forceinline__ u64 banded_wave_id() {
  const u32 pb = (gridDim.x + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return (u64)lb * WPB + w;
}
__device__ __forceinline__ u64 wave_id() {
  // uniform per wave; readfirstlane keeps the task decomposition on the scalar unit
  u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return (u64)blockIdx.x * WPB + w;
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

// linear-order kernels: the host splits the work into launches of < 2^31 items
constexpr u64 MAX_ITEMS = 0x7fffff00ull;

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
