// xg_host.cpp -- HOST build of the C ABI of include/xgcm_hip.h: the same `extern "C"` symbols over HOST pointers,
// compiled with g++ into libxgcm_host.so ("CPU build of the ABI" for BASELINE
// config 1, "plumbing, no GPU").
//
// What it is for: an integrator without a GPU can load a library with the exact symbol table of libxgcm_hip.so and
// exercise a binding end to end (argument marshalling, shapes, strides, error paths) on small arrays.  What it is
// NOT: a fallback.  xgcm_amd never loads it -- `xgcm_amd._hip.load()` opens libxgcm_hip.so only and the device
// layer raises without a GPU; the only users are tests/ (tests/host_abi_device.py) and examples/.  It has its own
// straightforward loops (one output cell at a time, index arithmetic in the open), written against the header's
// semantics, not against the kernels or the oracle.  It serves the 1-D operators of SURVEY.md section 8(a):
// stencil (+ pre-gathered halos), cumsum, reduce, pad, the broadcasting binary op and the synthetic generator; the
// fused / topology / transform entry points exist and return XG_ERR_UNSUPPORTED.
//
// Build: g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off  (no FMA contraction: same bit contract as the kernels)

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/xgcm_hip.h"

namespace {

thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int unsupported(const char* what) {
  return fail(XG_ERR_UNSUPPORTED, "%s: not part of the host build of the ABI (1-D operators only)", what);
}

// a C-contiguous N-D array seen as (outer, n, inner) around `axis`
struct View {
  int64_t outer = 1, n = 1, inner = 1;
};
int make_view(const int64_t* shape, int ndim, int axis, View* v) {
  if (!shape) return fail(XG_ERR_INVALID, "NULL shape");
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  if (axis < 0 || axis >= ndim) return fail(XG_ERR_INVALID, "axis %d out of range for ndim %d", axis, ndim);
  for (int d = 0; d < ndim; ++d)
    if (shape[d] < 0) return fail(XG_ERR_INVALID, "negative extent");
  v->n = shape[axis];
  for (int d = 0; d < axis; ++d) v->outer *= shape[d];
  for (int d = axis + 1; d < ndim; ++d) v->inner *= shape[d];
  return XG_OK;
}

// offset of the cell (o, k, x) of the (outer, n', inner) view in an array addressed by per-dim element strides
// (0 = broadcast) -- the metric convention of the header
int64_t strided_offset(const int64_t* shape, const int64_t* strides, int ndim, int axis, int64_t o, int64_t k, int64_t x) {
  int64_t off = k * strides[axis];
  for (int d = ndim - 1; d > axis; --d) {
    off += (x % shape[d]) * strides[d];
    x /= shape[d];
  }
  for (int d = axis - 1; d >= 0; --d) {
    off += (o % shape[d]) * strides[d];
    o /= shape[d];
  }
  return off;
}

template <typename R>
R op2(int op, R l, R r) {
  switch (op) {
    case XG_OP_DIFF: return r - l;
    case XG_OP_INTERP:
      if constexpr (std::is_integral_v<R>) return l + r;  // _i64: the wrapped sum, halved by the caller in float64
      else return (l + r) / R(2);
    case XG_OP_MIN: return (l != l || r != r) ? (l != l ? l : r) : (l < r ? l : r);  // NaN-propagating like np.min
    case XG_OP_MINU:  // integer entry points: the lanes hold an unsigned array
      if constexpr (std::is_integral_v<R>) return std::make_unsigned_t<R>(l) < std::make_unsigned_t<R>(r) ? l : r;
      else return l;
    case XG_OP_MAXU:
      if constexpr (std::is_integral_v<R>) return std::make_unsigned_t<R>(l) > std::make_unsigned_t<R>(r) ? l : r;
      else return l;
    default: return (l != l || r != r) ? (l != l ? l : r) : (l > r ? l : r);
  }
}

// P[q] of the header: the padded, m_in-weighted input at padded index q along the axis (halo: pre-gathered values)
template <typename R>
int stencil1d(int op, const R* in, const R* halo, R* out, const int64_t* shape, int ndim, int axis, int64_t n_out,
              int pad_lo, int pad_hi, int bc, R fill, const R* m_in, const int64_t* mis, const R* m_out,
              const int64_t* mos) {
  if (!in || !out) return fail(XG_ERR_INVALID, "NULL array argument");
  if (op < XG_OP_DIFF || op > (std::is_integral_v<R> ? XG_OP_MAXU : XG_OP_MAX)) return fail(XG_ERR_INVALID, "unknown op %d", op);
  if ((pad_lo | pad_hi) & ~1) return fail(XG_ERR_INVALID, "pad widths must be 0 or 1, got (%d,%d)", pad_lo, pad_hi);
  if (bc < XG_BC_NONE || bc > XG_BC_HALO) return fail(XG_ERR_INVALID, "unknown boundary mode %d", bc);
  if ((m_in && !mis) || (m_out && !mos)) return fail(XG_ERR_INVALID, "metric without strides");
  if (std::is_integral_v<R> && (m_in || m_out)) return fail(XG_ERR_UNSUPPORTED, "integer stencils take no metrics: convert to float64 first");
  View v;
  if (int rc = make_view(shape, ndim, axis, &v)) return rc;
  if (n_out != v.n + pad_lo + pad_hi - 1)
    return fail(XG_ERR_INVALID, "n_out %lld != n_in %lld + %d + %d - 1", (long long)n_out, (long long)v.n, pad_lo, pad_hi);
  if ((pad_lo || pad_hi) && bc == XG_BC_NONE) return fail(XG_ERR_INVALID, "halo cells requested but no boundary mode given");
  if (v.n < 1) return fail(XG_ERR_INVALID, "empty stencil axis");
  std::vector<int64_t> oshape(shape, shape + ndim);
  oshape[axis] = n_out;
  const int nhalo = pad_lo + pad_hi;
  for (int64_t o = 0; o < v.outer; ++o)
    for (int64_t j = 0; j < n_out; ++j)
      for (int64_t x = 0; x < v.inner; ++x) {
        R side[2];
        for (int s = 0; s < 2; ++s) {
          int64_t q = j + s - pad_lo;  // index into the unpadded input
          if (q >= 0 && q < v.n) {
            R val = in[(o * v.n + q) * v.inner + x];
            if (m_in) val = val * m_in[strided_offset(shape, mis, ndim, axis, o, q, x)];
            side[s] = val;
          } else if (bc == XG_BC_FILL) {
            side[s] = fill;  // the fill halo is not weighted: the reference pads after the product
          } else if (bc == XG_BC_HALO) {
            side[s] = halo[(o * nhalo + (q < 0 ? 0 : pad_lo)) * v.inner + x];
          } else {
            const int64_t src = (bc == XG_BC_PERIODIC) ? (q < 0 ? v.n - 1 : 0) : (q < 0 ? 0 : v.n - 1);
            R val = in[(o * v.n + src) * v.inner + x];
            if (m_in) val = val * m_in[strided_offset(shape, mis, ndim, axis, o, src, x)];
            side[s] = val;
          }
        }
        R res = op2<R>(op, side[0], side[1]);
        if (m_out) res = res / m_out[strided_offset(oshape.data(), mos, ndim, axis, o, j, x)];
        out[(o * n_out + j) * v.inner + x] = res;
      }
  return XG_OK;
}

template <typename R>
int cumsum1d(const R* in, R* out, const int64_t* shape, int ndim, int axis, int reverse, int skipna, int trim_lo,
             int trim_hi, int pad_lo, int pad_hi, int bc, R fill, const R* m_in, const int64_t* mis, const R* m_out,
             const int64_t* mos) {
  if (!in || !out) return fail(XG_ERR_INVALID, "NULL array argument");
  if ((trim_lo | trim_hi | pad_lo | pad_hi) & ~1) return fail(XG_ERR_INVALID, "trim/pad widths must be 0 or 1");
  if (bc < XG_BC_NONE || bc > XG_BC_EXTEND) return fail(XG_ERR_INVALID, "unknown boundary mode %d", bc);
  if ((pad_lo || pad_hi) && bc == XG_BC_NONE) return fail(XG_ERR_INVALID, "halo cells requested but no boundary mode given");
  if ((m_in && !mis) || (m_out && !mos)) return fail(XG_ERR_INVALID, "metric without strides");
  if (std::is_integral_v<R> && (m_in || m_out)) return fail(XG_ERR_UNSUPPORTED, "integer scans take no metrics: convert to float64 first");
  View v;
  if (int rc = make_view(shape, ndim, axis, &v)) return rc;
  const int64_t kept = v.n - trim_lo - trim_hi;
  if (kept < 1) return fail(XG_ERR_INVALID, "nothing left after trimming (n=%lld)", (long long)v.n);
  const int64_t n_out = kept + pad_lo + pad_hi;
  std::vector<int64_t> oshape(shape, shape + ndim);
  oshape[axis] = n_out;
  std::vector<R> c(v.n);
  for (int64_t o = 0; o < v.outer; ++o)
    for (int64_t x = 0; x < v.inner; ++x) {
      R acc = 0;
      for (int64_t t = 0; t < v.n; ++t) {  // sequential in scan order, like numpy.cumsum / nancumsum
        const int64_t k = reverse ? v.n - 1 - t : t;
        R val = in[(o * v.n + k) * v.inner + x];
        if (m_in) val = val * m_in[strided_offset(shape, mis, ndim, axis, o, k, x)];
        if (skipna && val != val) val = 0;
        acc = (t == 0) ? val : acc + val;
        c[k] = acc;
      }
      for (int64_t j = 0; j < n_out; ++j) {  // pad acts on the trimmed CUMULATIVE values
        const int64_t t = j - pad_lo;        // index into the trimmed result
        R val;
        if (t >= 0 && t < kept) val = c[trim_lo + t];
        else if (bc == XG_BC_FILL) val = fill;
        else if (bc == XG_BC_PERIODIC) val = c[trim_lo + (t < 0 ? kept - 1 : 0)];
        else val = c[trim_lo + (t < 0 ? 0 : kept - 1)];
        if (m_out) val = val / m_out[strided_offset(oshape.data(), mos, ndim, axis, o, j, x)];
        out[(o * n_out + j) * v.inner + x] = val;
      }
    }
  return XG_OK;
}

template <typename R>
int reduce1d(const R* in, R* out, const int64_t* shape, int ndim, int axis, int skipna, const R* w, const int64_t* ws) {
  if (!in || !out) return fail(XG_ERR_INVALID, "NULL array argument");
  if (w && !ws) return fail(XG_ERR_INVALID, "weight without strides");
  if (skipna < 0 || skipna > 7) return fail(XG_ERR_INVALID, "skipna / count / mean mode %d not in [0,7]", skipna);
  if (std::is_integral_v<R> && (w || skipna > 1)) return fail(XG_ERR_UNSUPPORTED, "integer reductions are plain sums: weights and means are float");
  View v;
  if (int rc = make_view(shape, ndim, axis, &v)) return rc;
  // one sequential sum per output cell in the given mode (k = 0 .. n-1 in order: numpy's order over a non-last axis)
  auto sum_in_mode = [&](int64_t o, int64_t x, int mode) -> R {
    R acc = 0;
    for (int64_t k = 0; k < v.n; ++k) {
      R val = in[(o * v.n + k) * v.inner + x];
      if (mode >= 2) val = (mode == 3 || val == val) ? R(1) : R(0);
      if (w) val = val * w[strided_offset(shape, ws, ndim, axis, o, k, x)];
      if (mode && val != val) val = 0;
      acc = (k == 0) ? val : acc + val;
    }
    return acc;
  };
  for (int64_t o = 0; o < v.outer; ++o)
    for (int64_t x = 0; x < v.inner; ++x) {
      R res;
      if (skipna == 4) res = sum_in_mode(o, x, 1) / sum_in_mode(o, x, 2);       // NaN-skipping weighted mean
      else if (skipna == 5) res = sum_in_mode(o, x, 0) / sum_in_mode(o, x, 3);  // weighted mean, NaN propagates
      else if (skipna >= 6) {  // numerator and denominator sums side by side
        res = sum_in_mode(o, x, skipna == 6 ? 1 : 0);
        out[(v.outer + o) * v.inner + x] = sum_in_mode(o, x, skipna == 6 ? 2 : 3);
      } else res = sum_in_mode(o, x, skipna);
      out[o * v.inner + x] = res;
    }
  return XG_OK;
}

template <typename R>
int pad_nd(const R* in, R* out, const int64_t* shape, int ndim, const int64_t* lo, const int64_t* hi, const int* bc,
           const R* fill, const int* order) {
  if (!in || !out || !shape || !lo || !hi || !bc) return fail(XG_ERR_INVALID, "NULL argument");
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  // one axis at a time in application order, each step a fresh array (numpy.pad chain semantics)
  std::vector<int64_t> cur(shape, shape + ndim);
  int64_t total = 1;
  for (int d = 0; d < ndim; ++d) total *= cur[d];
  std::vector<R> a(in, in + total), b;
  for (int step = 0; step < ndim; ++step) {
    const int ax = order ? order[step] : step;
    if (ax < 0 || ax >= ndim) return fail(XG_ERR_INVALID, "bad axis %d in order", ax);
    if (lo[ax] == 0 && hi[ax] == 0) continue;
    if (lo[ax] < 0 || hi[ax] < 0) return fail(XG_ERR_INVALID, "negative pad width");
    if (bc[ax] < XG_BC_PERIODIC || bc[ax] > XG_BC_EXTEND) return fail(XG_ERR_INVALID, "axis %d is padded but has no boundary mode", ax);
    View v;
    if (int rc = make_view(cur.data(), ndim, ax, &v)) return rc;
    if (v.n == 0 && bc[ax] != XG_BC_FILL) return fail(XG_ERR_INVALID, "can't extend empty axis %d using modes other than 'constant'", ax);
    const int64_t n2 = v.n + lo[ax] + hi[ax];
    b.assign((size_t)(v.outer * n2 * v.inner), R(0));
    for (int64_t o = 0; o < v.outer; ++o)
      for (int64_t j = 0; j < n2; ++j) {
        int64_t q = j - lo[ax];
        bool constant = false;
        if (q < 0 || q >= v.n) {
          if (bc[ax] == XG_BC_FILL) constant = true;
          else if (bc[ax] == XG_BC_PERIODIC) q = ((q % v.n) + v.n) % v.n;  // numpy 'wrap'
          else q = q < 0 ? 0 : v.n - 1;                                    // numpy 'edge'
        }
        for (int64_t x = 0; x < v.inner; ++x)
          b[(size_t)((o * n2 + j) * v.inner + x)] = constant ? (fill ? fill[ax] : R(0)) : a[(size_t)((o * v.n + q) * v.inner + x)];
      }
    a.swap(b);
    cur[ax] = n2;
  }
  int64_t n_out = 1;
  for (int d = 0; d < ndim; ++d) n_out *= cur[d];
  if (n_out) memcpy(out, a.data(), sizeof(R) * (size_t)n_out);
  return XG_OK;
}

template <typename R>
int binary(int op, const R* a, const int64_t* sa, const R* b, const int64_t* sb, R* out, const int64_t* shape, int ndim) {
  if (!a || !b || !out || !shape || !sa || !sb) return fail(XG_ERR_INVALID, "NULL argument");
  if (op < XG_BIN_MUL || op > XG_BIN_SUB) return fail(XG_ERR_INVALID, "unknown binary op %d", op);
  if (std::is_integral_v<R> && op == XG_BIN_DIV) return fail(XG_ERR_UNSUPPORTED, "true division leaves the integer domain: convert to float64 first");
  if (ndim < 1 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [1,%d]", ndim, XG_MAX_NDIM);
  int64_t total = 1;
  for (int d = 0; d < ndim; ++d) total *= shape[d];
  for (int64_t i = 0; i < total; ++i) {
    int64_t r = i, oa = 0, ob = 0;
    for (int d = ndim - 1; d >= 0; --d) {
      const int64_t c = r % shape[d];
      r /= shape[d];
      oa += c * sa[d];
      ob += c * sb[d];
    }
    const R x = a[oa], y = b[ob];
    if constexpr (std::is_integral_v<R>) out[i] = op == XG_BIN_MUL ? x * y : op == XG_BIN_ADD ? x + y : x - y;
    else out[i] = op == XG_BIN_MUL ? x * y : op == XG_BIN_DIV ? x / y : op == XG_BIN_ADD ? x + y : x - y;
  }
  return XG_OK;
}

template <typename R>
int fill_synthetic(R* out, int64_t n, uint64_t seed, uint64_t offset, double scale, double shift) {
  if (!out && n > 0) return fail(XG_ERR_INVALID, "NULL output");
  for (int64_t i = 0; i < n; ++i) {
    uint64_t z = (uint64_t)i + offset + seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u = (double)(z >> 11) * 0x1.0p-53;
    out[i] = (R)(u * scale + shift);
  }
  return XG_OK;
}

// tunables: the names of the device library (xg_runtime.hip::TUNABLES), accepted and remembered so that bindings
// can be exercised; they have no effect on the host loops
const char* const KNOWN_TUNABLES[] = {"seg", "nt_store", "nt_load", "seg_max_tiles", "scan_narrow_below", "pad_rows", "pad_nt", "transform_lds_kb", "transform_win", "transform_fast", "transform_stage", "transform_ring", "transform_cwin", "transform_lean", "zchunk", "zband", "zb_rows", "scan_block", "strided_gen", "march_band", "scan_vec", "scan_dpp", "contig_gen", "deep_waves", "contig_rw", "rw_zshare", "met_zk", "met_zk1", "vec_zk", "vec_nt", "vec_zb_rows", "nb_dpp", "met_zk2", "met_ys", "met_ys1", "met_ys2", "seg_ys", "contig_rw_mi", "met_seg", "met_seg1", "met_scalar", "scan_pipe", "scan_u", "scan_pace", "scan_chain", "scan_chain_w", "scan_chain_spin", "scan_chain_tmaj", "reduce_zl", "pad_tpw", "bin_idx32", "reduce_ldsw_u", "march_ofast", "reduce_sk", "reduce_ru", "reduce_wfast", "reduce_wg", "scan_sh1", "reduce_ldsw", "reduce_zmarch", "dbg", "march_lds_kb"};
struct Knob { const char* name; int value; bool set; };
std::vector<Knob>& knobs() {
  static std::vector<Knob> k;
  if (k.empty())
    for (const char* n : KNOWN_TUNABLES) k.push_back({n, 0, false});
  return k;
}

// fused vorticity / divergence of the header (docs/ufunc_examples.md): the operator chain's arithmetic cell by cell --
// (dx - dy) / area resp. (dx + dy) / area with the two one-cell halos from bc_x / bc_y -- so that the host build serves the
// config-5 workload of bench.py's CPU dry run (tests/test_bench_dryrun.py); shapes (.., Y, X), area through broadcast strides
template <typename R>
int curl_or_div(bool curl, const R* u, const R* v, const R* area, const int64_t* as, R* out, const int64_t* shape, int ndim,
                int bc_x, R fill_x, int bc_y, R fill_y) {
  if (!u || !v || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if (ndim < 2 || ndim > XG_MAX_NDIM) return fail(XG_ERR_UNSUPPORTED, "ndim %d not in [2,%d]", ndim, XG_MAX_NDIM);
  if (area && !as) return fail(XG_ERR_INVALID, "metric without strides");
  for (int b : {bc_x, bc_y})
    if (b < XG_BC_PERIODIC || b > XG_BC_EXTEND) return fail(XG_ERR_INVALID, "boundary mode %d: periodic, fill or extend", b);
  const int64_t ny = shape[ndim - 2], nx = shape[ndim - 1];
  int64_t outer = 1;
  for (int d = 0; d < ndim - 2; ++d) outer *= shape[d];
  if (outer == 0 || ny == 0 || nx == 0) return XG_OK;
  // neighbour along one axis: index k + s (s = -1 for the curl's center->left, +1 for the divergence's left->center)
  auto nb = [](const R* row, int64_t k, int64_t n, int64_t stride, int s, int bc, R fill) -> R {
    const int64_t q = k + s;
    if (q >= 0 && q < n) return row[q * stride];
    if (bc == XG_BC_FILL) return fill;
    if (bc == XG_BC_PERIODIC) return row[((q % n + n) % n) * stride];
    return row[(q < 0 ? 0 : n - 1) * stride];
  };
  for (int64_t o = 0; o < outer; ++o) {
    int64_t rem = o, aoff = 0;  // the outer index decomposed for the area's broadcast strides
    for (int d = ndim - 3; d >= 0; --d) { const int64_t i = rem % shape[d]; rem /= shape[d]; if (area) aoff += i * as[d]; }
    const R *pu = u + o * ny * nx, *pv = v + o * ny * nx;
    R* po = out + o * ny * nx;
    for (int64_t j = 0; j < ny; ++j)
      for (int64_t i = 0; i < nx; ++i) {
        R r;
        if (curl) {
          const R dvdx = pv[j * nx + i] - nb(pv + j * nx, i, nx, 1, -1, bc_x, fill_x);
          const R dudy = pu[j * nx + i] - nb(pu + i, j, ny, nx, -1, bc_y, fill_y);
          r = dvdx - dudy;
        } else {
          const R dudx = nb(pu + j * nx, i, nx, 1, +1, bc_x, fill_x) - pu[j * nx + i];
          const R dvdy = nb(pv + i, j, ny, nx, +1, bc_y, fill_y) - pv[j * nx + i];
          r = dudx + dvdy;
        }
        po[j * nx + i] = area ? r / area[aoff + j * as[ndim - 2] + i * as[ndim - 1]] : r;
      }
  }
  return XG_OK;
}


}  // namespace

extern "C" {

int xg_version(void) { return XG_ABI_VERSION; }
int xg_last_error(char* buf, int n) {
  const int len = (int)strlen(g_err);
  if (buf && n > 0) {
    const int c = len < n - 1 ? len : n - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return len;
}
int xg_set_tunable(const char* name, int value) {
  if (!name) return fail(XG_ERR_INVALID, "NULL tunable name");
  for (auto& k : knobs())
    if (!strcmp(k.name, name)) { k.value = value; k.set = true; return XG_OK; }
  return fail(XG_ERR_INVALID, "unknown tunable '%s'", name);
}
int xg_get_tunable(const char* name, int* value) {
  if (!name || !value) return fail(XG_ERR_INVALID, "NULL argument");
  for (auto& k : knobs())
    if (!strcmp(k.name, name)) { *value = k.value; return XG_OK; }
  return fail(XG_ERR_INVALID, "unknown tunable '%s'", name);
}
int xg_device_count(void) { return 0; }  // the host build drives no device
int xg_set_device(int) { return fail(XG_ERR_UNSUPPORTED, "host build of the ABI: no device"); }
int xg_malloc(void** ptr, uint64_t bytes) {
  if (!ptr) return fail(XG_ERR_INVALID, "NULL argument");
  *ptr = malloc(bytes ? bytes : 1);
  return *ptr ? XG_OK : fail(XG_ERR_HIP, "out of host memory");
}
int xg_free(void* ptr) { free(ptr); return XG_OK; }
int xg_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void*) { if (bytes) memcpy(dst, src, bytes); return XG_OK; }
int xg_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void*) { if (bytes) memcpy(dst, src, bytes); return XG_OK; }
int xg_stream_sync(void*) { return XG_OK; }
int xg_pin_host(void* ptr, uint64_t bytes) { return (ptr && bytes) ? XG_OK : fail(XG_ERR_INVALID, "empty host range"); }  // nothing to lock
int xg_unpin_host(void*) { return XG_OK; }
int xg_stream_create(void** stream) {
  if (!stream) return fail(XG_ERR_INVALID, "NULL argument");
  *stream = nullptr;  // the host build runs everything on the calling thread
  return XG_OK;
}
int xg_stream_destroy(void*) { return XG_OK; }
// (no chained kernels on the host: nothing ever gives up)
int xg_chain_status(int* gave_up, int* redone) {
  if (gave_up) *gave_up = 0;
  if (redone) *redone = 0;
  return XG_OK;
}
int xg_chain_rearm(void) { return XG_OK; }
int xg_scatter_alloc(void** ptr, uint64_t bytes, uint64_t, int, uint64_t) {  // host build: plain memory (placement is an HBM matter)
  if (!ptr || !bytes) return fail(XG_ERR_INVALID, "NULL / empty request");
  *ptr = malloc(bytes);
  return *ptr ? XG_OK : fail(XG_ERR_HIP, "out of host memory");
}
int xg_scatter_free(void* ptr) { free(ptr); return XG_OK; }
int xg_scatter_stats(uint64_t* a, uint64_t* b, uint64_t* c) { if (a) *a = 0; if (b) *b = 0; if (c) *c = 0; return XG_OK; }
int xg_scatter_grade(void* p, uint64_t, double* ratio) { if (!p || !ratio) return fail(XG_ERR_INVALID, "NULL argument"); *ratio = 1.0; return XG_OK; }  // host memory: nothing to grade
int xg_scatter_grade_stats(uint64_t* a, uint64_t* b) { if (a) *a = 0; if (b) *b = 0; return XG_OK; }
void* xg_pool_alloc(ssize_t size, int, void*) { return size > 0 ? malloc((size_t)size) : nullptr; }
void xg_pool_free(void* ptr, ssize_t, int, void*) { free(ptr); }
int xg_bswap(void* data, uint64_t nelem, int elem_bytes, void*) {
  if (elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(XG_ERR_INVALID, "byte swap of %d-byte elements (2, 4 or 8)", elem_bytes);
  if (nelem && !data) return fail(XG_ERR_INVALID, "NULL buffer");
  unsigned char* p = (unsigned char*)data;
  for (uint64_t e = 0; e < nelem; ++e, p += elem_bytes)
    for (int b = 0; b < elem_bytes / 2; ++b) { unsigned char t = p[b]; p[b] = p[elem_bytes - 1 - b]; p[elem_bytes - 1 - b] = t; }
  return XG_OK;
}
int xg_mask_value(void* data, uint64_t nelem, int elem_bytes, double value, void*) {
  if (elem_bytes != 4 && elem_bytes != 8) return fail(XG_ERR_INVALID, "masking of %d-byte elements (float32 / float64 only)", elem_bytes);
  if (nelem && !data) return fail(XG_ERR_INVALID, "NULL buffer");
  if (elem_bytes == 4) { float* p = (float*)data; for (uint64_t i = 0; i < nelem; ++i) if (p[i] == (float)value) p[i] = NAN; }
  else { double* p = (double*)data; for (uint64_t i = 0; i < nelem; ++i) if (p[i] == value) p[i] = NAN; }
  return XG_OK;
}
int xg_event_create(void** ev) {
  if (!ev) return fail(XG_ERR_INVALID, "NULL argument");
  *ev = calloc(1, sizeof(double));
  return *ev ? XG_OK : fail(XG_ERR_HIP, "out of host memory");
}
int xg_event_record(void* ev, void*) {
  if (!ev) return fail(XG_ERR_INVALID, "NULL event");
  *(double*)ev = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return XG_OK;
}
int xg_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return fail(XG_ERR_INVALID, "NULL argument");
  *ms = (float)(*(double*)stop - *(double*)start);
  return XG_OK;
}
int xg_event_destroy(void* ev) { free(ev); return XG_OK; }

// the entry points whose results keep the lanes' width (every build: f64, f32, i64, i32)
#define XG_HOST_LANES(SFX, R)                                                                                         \
  int xg_stencil1d_##SFX(int op, const R* in, R* out, const int64_t* shape, int ndim, int axis, int64_t n_out,        \
                         int pad_lo, int pad_hi, int bc, R fill, const R* m_in, const int64_t* mis, const R* m_out,   \
                         const int64_t* mos, void*) {                                                                 \
    if (bc == XG_BC_HALO) return fail(XG_ERR_INVALID, "XG_BC_HALO needs xg_stencil1d_halo");                          \
    return stencil1d<R>(op, in, nullptr, out, shape, ndim, axis, n_out, pad_lo, pad_hi, bc, fill, m_in, mis, m_out,   \
                        mos);                                                                                         \
  }                                                                                                                   \
  int xg_stencil1d_halo_##SFX(int op, const R* in, const R* halo, R* out, const int64_t* shape, int ndim, int axis,   \
                              int64_t n_out, int pad_lo, int pad_hi, const R* m_out, const int64_t* mos, void*) {     \
    if (!halo && (pad_lo || pad_hi)) return fail(XG_ERR_INVALID, "NULL halo buffer");                                 \
    return stencil1d<R>(op, in, halo, out, shape, ndim, axis, n_out, pad_lo, pad_hi,                                  \
                        (pad_lo || pad_hi) ? XG_BC_HALO : XG_BC_NONE, R(0), nullptr, nullptr, m_out, mos);            \
  }                                                                                                                   \
  int xg_pad_##SFX(const R* in, R* out, const int64_t* shape, int ndim, const int64_t* lo, const int64_t* hi,         \
                   const int* bc, const R* fill, const int* order, void*) {                                           \
    return pad_nd<R>(in, out, shape, ndim, lo, hi, bc, fill, order);                                                  \
  }                                                                                                                   \
  int xg_binary_##SFX(int op, const R* a, const int64_t* sa, const R* b, const int64_t* sb, R* out,                   \
                      const int64_t* shape, int ndim, void*) {                                                        \
    return binary<R>(op, a, sa, b, sb, out, shape, ndim);                                                             \
  }                                                                                                                   \
  int xg_halo_put_##SFX(const R* halo, R* out, const int64_t* shape, int ndim, int axis, int pad_lo, int pad_hi, void*) { \
    View v;                                                                                                           \
    if (int rc = make_view(shape, ndim, axis, &v)) return rc;                                                         \
    const int64_t nh = (int64_t)pad_lo + pad_hi;                                                                      \
    if (pad_lo < 0 || pad_hi < 0 || v.n < nh) return fail(XG_ERR_INVALID, "bad halo widths");                         \
    if (v.outer * nh * v.inner == 0) return XG_OK;                                                                    \
    if (!halo || !out) return fail(XG_ERR_INVALID, "NULL array argument");                                            \
    for (int64_t o = 0; o < v.outer; ++o)                                                                             \
      for (int64_t h = 0; h < nh; ++h)                                                                                \
        for (int64_t x = 0; x < v.inner; ++x)                                                                         \
          out[(o * v.n + (h < pad_lo ? h : v.n - nh + h)) * v.inner + x] = halo[(o * nh + h) * v.inner + x];          \
    return XG_OK;                                                                                                     \
  }                                                                                                                   \
  int xg_gather_##SFX(const R*, const R*, R*, const int64_t*, const int64_t*, const int64_t*, int, const int*,        \
                      const int*, const int64_t*, const int64_t*, int64_t, const R*, int, void*) {                    \
    return unsupported("xg_gather");                                                                                  \
  }

// scans and sums (numpy accumulates integers in 64 bits: no i32 build)
#define XG_HOST_SUMS(SFX, R)                                                                                          \
  int xg_cumsum1d_##SFX(const R* in, R* out, const int64_t* shape, int ndim, int axis, int reverse, int skipna,       \
                        int trim_lo, int trim_hi, int pad_lo, int pad_hi, int bc, R fill, const R* m_in,              \
                        const int64_t* mis, const R* m_out, const int64_t* mos, void*) {                              \
    return cumsum1d<R>(in, out, shape, ndim, axis, reverse, skipna, trim_lo, trim_hi, pad_lo, pad_hi, bc, fill,       \
                       m_in, mis, m_out, mos);                                                                        \
  }                                                                                                                   \
  int xg_reduce1d_##SFX(const R* in, R* out, const int64_t* shape, int ndim, int axis, int skipna, const R* w,        \
                        const int64_t* ws, void*) {                                                                   \
    return reduce1d<R>(in, out, shape, ndim, axis, skipna, w, ws);                                                    \
  }

// the float-only entry points (synthetic generator; fused / transform stubs)
#define XG_HOST_FLOAT(SFX, R)                                                                                         \
  int xg_stencil1d_halo_w_##SFX(int op, const R* in, const R* halo, R* out, const int64_t* shape, int ndim, int axis, \
                                int64_t n_out, int pad_lo, int pad_hi, const R* m_in, const int64_t* mis,             \
                                const R* m_out, const int64_t* mos, void*) {                                          \
    if (!halo && (pad_lo || pad_hi)) return fail(XG_ERR_INVALID, "NULL halo buffer");                                 \
    return stencil1d<R>(op, in, halo, out, shape, ndim, axis, n_out, pad_lo, pad_hi,                                  \
                        (pad_lo || pad_hi) ? XG_BC_HALO : XG_BC_NONE, R(0), m_in, mis, m_out, mos);                   \
  }                                                                                                                   \
  int xg_fill_synthetic_##SFX(R* out, int64_t n, uint64_t seed, uint64_t offset, double scale, double shift, void*) { \
    return fill_synthetic<R>(out, n, seed, offset, scale, shift);                                                     \
  }                                                                                                                   \
  int xg_transform_linear_##SFX(const R*, const R*, const int64_t*, const R*, const int64_t*, int64_t, R*,            \
                                const int64_t*, int, int, int, int, int, void*) {                                     \
    return unsupported("xg_transform_linear");                                                                        \
  }                                                                                                                   \
  int xg_transform_conservative_##SFX(const R*, const R*, const int64_t*, const R*, int64_t, R*, const int64_t*, int, \
                                      int, void*) {                                                                   \
    return unsupported("xg_transform_conservative");                                                                  \
  }                                                                                                                   \
  int xg_vorticity_##SFX(const R* u, const R* v, const R* area, const int64_t* as, R* out, const int64_t* shape,      \
                         int ndim, int bc_x, R fill_x, int bc_y, R fill_y, void*) {                                   \
    return curl_or_div<R>(true, u, v, area, as, out, shape, ndim, bc_x, fill_x, bc_y, fill_y);                        \
  }                                                                                                                   \
  int xg_divergence_##SFX(const R* u, const R* v, const R* area, const int64_t* as, R* out, const int64_t* shape,     \
                          int ndim, int bc_x, R fill_x, int bc_y, R fill_y, void*) {                                  \
    return curl_or_div<R>(false, u, v, area, as, out, shape, ndim, bc_x, fill_x, bc_y, fill_y);                       \
  }                                                                                                                   \
  int xg_gradient_##SFX(const R*, R*, R*, const int64_t*, int, int, R, int, R, const R*, const int64_t*, const R*,    \
                        const int64_t*, void*) {                                                                      \
    return unsupported("xg_gradient");                                                                                \
  }                                                                                                                   \
  int xg_flux_##SFX(const R*, const R*, const R*, R*, R*, const int64_t*, int, int, R, int, R, void*) {               \
    return unsupported("xg_flux");                                                                                    \
  }                                                                                                                   \
  int xg_gradient_halo_##SFX(const R*, const R*, const R*, R*, R*, const int64_t*, int, int, R, int, R, const R*,     \
                             const int64_t*, const R*, const int64_t*, void*) {                                       \
    return unsupported("xg_gradient_halo");                                                                           \
  }                                                                                                                   \
  int xg_flux_halo_##SFX(const R*, const R*, const R*, const R*, const R*, R*, R*, const int64_t*, int, int, R, int,  \
                         R, void*) {                                                                                  \
    return unsupported("xg_flux_halo");                                                                               \
  }                                                                                                                   \
  int xg_vorticity_halo_##SFX(const R*, const R*, const R*, const R*, const R*, const int64_t*, R*, const int64_t*,   \
                              int, int, R, int, R, void*) {                                                           \
    return unsupported("xg_vorticity_halo");                                                                          \
  }                                                                                                                   \
  int xg_divergence_halo_##SFX(const R*, const R*, const R*, const R*, const R*, const int64_t*, R*, const int64_t*,  \
                               int, int, R, int, R, void*) {                                                          \
    return unsupported("xg_divergence_halo");                                                                         \
  }                                                                                                                   \
  int xg_stencil2d_##SFX(int, const R*, R*, const int64_t*, int, int, int, int, int, R, int, int, int, R, void*) {    \
    return unsupported("xg_stencil2d");                                                                               \
  }                                                                                                                   \
  int xg_stencil2d_metric_##SFX(int, const R*, R*, const int64_t*, int, int, int, int, int, R, int, int, int, R,      \
                                const R*, const R*, const R*, void*) {                                                \
    return unsupported("xg_stencil2d_metric");                                                                        \
  }

XG_HOST_LANES(f64, double)
XG_HOST_LANES(f32, float)
XG_HOST_LANES(i64, int64_t)  // built with -fwrapv: overflow wraps like numpy's integer arithmetic
XG_HOST_LANES(i32, int32_t)
XG_HOST_SUMS(f64, double)
XG_HOST_SUMS(f32, float)
XG_HOST_SUMS(i64, int64_t)
XG_HOST_FLOAT(f64, double)
XG_HOST_FLOAT(f32, float)

// strided N-d copy (see the header): an odometer over the index space, bytes moved with memcpy
int xg_copy_nd(const void* src, const int64_t* ss, void* dst, const int64_t* ds, const int64_t* shape, int ndim, int eb, void*) {
  if (!shape || !ss || !ds) return fail(XG_ERR_INVALID, "NULL shape / stride argument");
  if (ndim < 0 || ndim > XG_MAX_NDIM) return fail(XG_ERR_INVALID, "ndim %d not in [0,%d]", ndim, XG_MAX_NDIM);
  if (eb != 1 && eb != 2 && eb != 4 && eb != 8) return fail(XG_ERR_INVALID, "element size %d not 1, 2, 4 or 8", eb);
  int64_t total = 1;
  for (int d = 0; d < ndim; ++d) {
    if (shape[d] < 0) return fail(XG_ERR_INVALID, "negative extent");
    if (shape[d] > 1 && ds[d] < 0) return fail(XG_ERR_INVALID, "destination strides must be positive");
    if (shape[d] > 1 && ds[d] == 0) return fail(XG_ERR_INVALID, "destination stride 0 on a dim of extent %lld (cells written more than once)", (long long)shape[d]);
    total *= shape[d];
  }
  if (total == 0) return XG_OK;
  if (!src || !dst) return fail(XG_ERR_INVALID, "NULL array argument");
  int64_t idx[XG_MAX_NDIM] = {0};
  const char* s = static_cast<const char*>(src);
  char* o = static_cast<char*>(dst);
  for (int64_t i = 0; i < total; ++i) {
    int64_t so = 0, dof = 0;
    for (int d = 0; d < ndim; ++d) { so += idx[d] * ss[d]; dof += idx[d] * ds[d]; }
    memcpy(o + dof * eb, s + so * eb, (size_t)eb);
    for (int d = ndim - 1; d >= 0; --d) {
      if (++idx[d] < shape[d]) break;
      idx[d] = 0;
    }
  }
  return XG_OK;
}

// numpy `astype` between the storage dtype and the compute dtype (see the header); one element at a time
// IEEE binary16 <-> double without compiler support (gcc 11 has no _Float16): one round-to-nearest-even from the double
static uint16_t half_from_double(double d) {
  uint64_t b;
  memcpy(&b, &d, 8);
  const uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
  int64_t e = (int64_t)((b >> 52) & 0x7ff);
  uint64_t m = b & 0xfffffffffffffull;
  if (e == 0x7ff) return (uint16_t)(sign | 0x7c00u | (m ? (0x200u | (uint16_t)(m >> 42)) : 0u));  // inf / quiet NaN
  e = e - 1023 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  int shift = 42;
  if (e <= 0) {  // subnormal half (or zero): the implicit one joins the mantissa, the shift grows
    if (e < -10) return sign;
    m |= 1ull << 52;
    shift = 42 + (int)(1 - e);
    e = 0;
  }
  uint64_t q = m >> shift;
  const uint64_t rem = m & ((1ull << shift) - 1), half = 1ull << (shift - 1);
  if (rem > half || (rem == half && (q & 1))) ++q;  // a carry out of the mantissa bumps the exponent (up to inf): correct
  return (uint16_t)(sign | (uint16_t)(((uint64_t)e << 10) + q));
}
static double half_to_double(uint16_t h) {
  const int e = (h >> 10) & 31, m = h & 0x3ff;
  double v;
  if (e == 0) v = std::ldexp((double)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = std::ldexp((double)(m | 0x400), e - 25);
  return (h & 0x8000) ? -v : v;
}
int xg_convert(const void* src, int st, void* dst, int dt, uint64_t n, int via, double scale, int flags, void*) {
  if (st < XG_T_BOOL || st > XG_T_F16 || dt < XG_T_BOOL || dt > XG_T_F16) return fail(XG_ERR_INVALID, "unknown element type (%d -> %d)", st, dt);
  if (via < -1 || via > XG_T_U64) return fail(XG_ERR_INVALID, "via_type %d is not an integer type", via);
  if (flags & ~1) return fail(XG_ERR_INVALID, "unknown flags %d", flags);
  if (n == 0) return XG_OK;
  if (!src || !dst) return fail(XG_ERR_INVALID, "NULL array argument");
  const bool sf = st >= XG_T_F32, df = dt >= XG_T_F32;
  if (sf && via != -1) return fail(XG_ERR_INVALID, "via_type applies to integer sources only");
  auto bytes = [](int t) { return (t == XG_T_BOOL || t == XG_T_I8 || t == XG_T_U8) ? 1 : (t == XG_T_I16 || t == XG_T_U16 || t == XG_T_F16) ? 2 : (t == XG_T_I32 || t == XG_T_U32 || t == XG_T_F32) ? 4 : 8; };
  if ((flags & 1) && (bytes(st) != 8 || bytes(dt) != 8 || sf || df)) return fail(XG_ERR_INVALID, "the sign-bit flip needs 64-bit integer types on both sides");
  auto wrap = [](int64_t w, int t) -> int64_t {
    switch (t) {
      case XG_T_BOOL: return w != 0;
      case XG_T_I8: return (int8_t)w;   case XG_T_I16: return (int16_t)w;  case XG_T_I32: return (int32_t)w;
      case XG_T_U8: return (uint8_t)w;  case XG_T_U16: return (uint16_t)w; case XG_T_U32: return (uint32_t)w;
      default: return w;
    }
  };
  const int logical = via != -1 ? via : st;
  for (uint64_t i = 0; i < n; ++i) {
    int64_t w = 0;
    double f = 0.0;
    switch (st) {  // load
      case XG_T_BOOL: w = ((const uint8_t*)src)[i] != 0; break;
      case XG_T_I8: w = ((const int8_t*)src)[i]; break;     case XG_T_I16: w = ((const int16_t*)src)[i]; break;
      case XG_T_I32: w = ((const int32_t*)src)[i]; break;   case XG_T_I64: w = ((const int64_t*)src)[i]; break;
      case XG_T_U8: w = ((const uint8_t*)src)[i]; break;    case XG_T_U16: w = ((const uint16_t*)src)[i]; break;
      case XG_T_U32: w = ((const uint32_t*)src)[i]; break;  case XG_T_U64: w = (int64_t)((const uint64_t*)src)[i]; break;
      case XG_T_F32: f = ((const float*)src)[i]; break;     case XG_T_F16: f = half_to_double(((const uint16_t*)src)[i]); break;
      default: f = ((const double*)src)[i]; break;
    }
    if (!sf) w = wrap(w, via);
    if (df) {  // float destination
      if (dt == XG_T_F16) {  // one rounding into binary16, then the scale in binary16 (a power of two: exact unless subnormal)
        const double v = sf ? f : (logical == XG_T_U64 ? (double)(uint64_t)w : (double)w);
        const uint16_t h = half_from_double(v);
        ((uint16_t*)dst)[i] = scale == 1.0 ? h : half_from_double(half_to_double(h) * half_to_double(half_from_double(scale)));
      } else if (dt == XG_T_F32) {
        const float v = sf ? (float)f : (logical == XG_T_U64 ? (float)(uint64_t)w : (float)w);
        ((float*)dst)[i] = v * (float)scale;
      } else {
        const double v = sf ? f : (logical == XG_T_U64 ? (double)(uint64_t)w : (double)w);
        ((double*)dst)[i] = v * scale;
      }
      continue;
    }
    if (sf) w = (dt == XG_T_U64) ? (int64_t)(uint64_t)f : (int64_t)f;
    if (dt == XG_T_BOOL) { ((uint8_t*)dst)[i] = sf ? (f != 0.0) : (w != 0); continue; }
    if (flags & 1) w ^= (int64_t)0x8000000000000000ull;
    switch (dt) {
      case XG_T_I8: ((int8_t*)dst)[i] = (int8_t)w; break;     case XG_T_I16: ((int16_t*)dst)[i] = (int16_t)w; break;
      case XG_T_I32: ((int32_t*)dst)[i] = (int32_t)w; break;  case XG_T_I64: ((int64_t*)dst)[i] = w; break;
      case XG_T_U8: ((uint8_t*)dst)[i] = (uint8_t)w; break;   case XG_T_U16: ((uint16_t*)dst)[i] = (uint16_t)w; break;
      case XG_T_U32: ((uint32_t*)dst)[i] = (uint32_t)w; break; default: ((uint64_t*)dst)[i] = (uint64_t)w; break;
    }
  }
  return XG_OK;
}

}  // extern "C"
