/*
 * xgcm_hip.h -- C ABI of libxgcm_hip.so: MI355X (gfx950) kernels for the xgcm staggered-grid
 * 1-D stencil hot path (Grid.diff / interp / min / max / cumsum / derivative / integrate).
 *
 * Every array argument is a DEVICE pointer to a C-contiguous float64 array described by
 * (shape[ndim], axis); the caller owns all buffers; kernels never write their inputs; `out`
 * must not alias an input.  All calls are asynchronous on `stream` (a hipStream_t passed as
 * void*, NULL = the null stream), re-entrant and thread-safe.  Return value: 0 on success,
 * a negative xg_status otherwise (message via xg_last_error, thread-local).  Semantic
 * validation that the reference expresses as Python exceptions (missing boundary condition,
 * unknown position pair ...) stays in the host layer so messages match the reference.
 *
 * "Metric" arguments are optional (NULL = absent) float64 arrays addressed through
 * per-dimension ELEMENT strides given in the coordinate system of the array they weight
 * (0 = broadcast along that dim), exactly the broadcasting xarray performs for `da * metric`.
 *
 * Reference interfaces replaced (paths relative to the xgcm source tree):
 *   xg_stencil1d_f64   xgcm/grid_ufunc.py:885-904 (pad-then-apply) + xgcm/padding.py:575-616
 *                      (_pad_basic) + xgcm/gridops.py:23-215 (diff/interp/min/max bodies)
 *                      + xgcm/grid.py:804-808,830-832 (metric_weighted) + :1576-1578 (derivative)
 *   xg_stencil1d_halo_f64  the same operators on grids with face connections / a north fold
 *                      (halo values pre-gathered by xg_gather_f64 instead of a padded copy)
 *   xg_cumsum1d_f64    xgcm/grid.py:1295-1414 (Grid.cumsum per-axis body) and the 8 cumsum
 *                      grid ufuncs xgcm/gridops.py:221-278
 *   xg_reduce1d_f64    xgcm/grid.py:1598-1605 (Grid.integrate: (da*weight).sum(dim))
 *   xg_pad_f64         xgcm/padding.py:765-871 (pad) for user grid-ufuncs of any width
 *   xg_gather_f64      xgcm/padding.py:260-572 (_pad_face_connections) and :619-762 (_fold_north_halo,
 *                      _pad_fold): halos of complex topologies as one gather through a token map
 *   xg_halo_put_f64    xgcm/grid.py:1385-1395 (Grid.cumsum pads the cumulative field) on such topologies, in place
 *   xg_transform_linear_f64 / xg_transform_conservative_f64
 *                      xgcm/transform.py:15-41 (_interp_1d_linear) and :88-142
 *                      (_interp_1d_conservative), the numba gufuncs behind Grid.transform
 *   xg_binary_f64      the xarray broadcasting `*`, `/`, `+`, `-` around the ops
 *                      (xgcm/grid.py:808,832,1578,1600; get_metric products :614-617)
 *   xg_stencil2d_f64   Grid.interp/diff/min/max over two axes (xgcm/grid.py:798-828), one pass
 *   xg_stencil2d_metric_f64   the same with metric_weighted on both axes (xgcm/grid.py:804-828)
 *   xg_gradient_f64 / xg_flux_f64  the "Gradient" and "Advection" grid ufuncs of docs/ufunc_examples.md, fused
 *   xg_divergence_f64  the chained (diff(u,X) + diff(v,Y)) / area of docs/ufunc_examples.md, fused
 *   xg_vorticity_f64   the chained (diff(v,X) - diff(u,Y)) / area of docs/ufunc_examples.md
 *                      (one fused pass instead of three apply_ufunc passes, grid.py:798-800 TODO)
 *   xg_*_i64           the same bodies on integer arrays, which numpy keeps integral and wraps
 *                      (xgcm/gridops.py:23-24,123-126,172-175,227-278; xgcm/padding.py:610-615)
 *   xg_convert         numpy's dtype promotion / `astype` around them (int * float metric: xgcm/grid.py:804-808)
 */
#ifndef XGCM_HIP_H
#define XGCM_HIP_H

#include <stdint.h>
#include <sys/types.h> /* ssize_t */

#ifdef __cplusplus
extern "C" {
#endif

#define XG_ABI_VERSION 1
#define XG_MAX_NDIM 8

typedef enum xg_status {
  XG_OK = 0,
  XG_ERR_INVALID = -1,     /* bad argument (shape/axis/widths inconsistent, NULL pointer ...) */
  XG_ERR_UNSUPPORTED = -2, /* valid request this build cannot serve (ndim > XG_MAX_NDIM ...)  */
  XG_ERR_HIP = -3          /* HIP runtime error; text in xg_last_error                        */
} xg_status;

/* raw two-point bodies, xgcm/gridops.py:23-24,76-77,123-126,172-175 */
/* XG_OP_MINU / XG_OP_MAXU: integer entry points (*_i64, *_i32) only -- min / max of lanes that hold an UNSIGNED array
 * (uint64 / uint32 compare differently from the two's-complement lanes they are stored in). */
typedef enum xg_op { XG_OP_DIFF = 0, XG_OP_INTERP = 1, XG_OP_MIN = 2, XG_OP_MAX = 3, XG_OP_MINU = 4, XG_OP_MAXU = 5 } xg_op;

/* boundary modes, xgcm/padding.py:15-19.  XG_BC_NONE is only legal when no halo cell is read. */
/* XG_BC_HALO (internal to xg_stencil1d_halo_*): halo values were gathered beforehand. */
typedef enum xg_bc { XG_BC_NONE = 0, XG_BC_PERIODIC = 1, XG_BC_FILL = 2, XG_BC_EXTEND = 3, XG_BC_HALO = 4 } xg_bc;

typedef enum xg_binop { XG_BIN_MUL = 0, XG_BIN_DIV = 1, XG_BIN_ADD = 2, XG_BIN_SUB = 3 } xg_binop;

/* ---- library / device management ------------------------------------------------------ */
int xg_version(void);
/* copies the calling thread's last error text (NUL-terminated) into buf; returns its length */
int xg_last_error(char* buf, int n);
/* Launch-shape tunables (rows per wave-task, band heights, window lengths ...; names in INTEGRATION.md): each
 * starts from the environment variable XG_<NAME> or its measured default; set / read one at run time.  They
 * change speed only: results are bit-identical under every setting, with one documented exception -- sums along the
 * CONTIGUOUS axis are re-associated by contract (1e-12 relative), and `scan_dpp` / `scan_vec` / `scan_block` select
 * among associations there.  (`dbg` holds A/B switches of the measurement tools, not a user setting.)
 * THE ONE EXCEPTION TO THE THREADING RULE BELOW: process-wide and not synchronised -- set them from one thread while no
 * other thread is inside the library (a measurement tool's job, not an operator's).
 *
 * Threading (the reference's ufuncs run concurrently on dask scheduler threads, xgcm/grid.py:786-789,
 * xgcm/grid_ufunc.py:966-984): every other entry point may be called from any number of host threads at once, on distinct
 * streams or on one stream.  The library keeps no per-call host state; its only shared state is the lazily created
 * per-(device, stream) hand-off workspace of the chained scans, handed out under a mutex, and the error text, which is
 * thread-local (xg_last_error returns the CALLING thread's).  tests/test_gpu_threads.py drives stencils, chained and
 * marching scans, weighted reductions, fused vorticity and a failing call from 6 threads and compares every result with
 * the single-threaded one bit for bit. */
int xg_set_tunable(const char* name, int value);
int xg_get_tunable(const char* name, int* value);
int xg_device_count(void);
int xg_set_device(int device);
int xg_malloc(void** ptr, uint64_t bytes);
int xg_free(void* ptr);
int xg_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream);
int xg_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream);
int xg_stream_sync(void* stream);
/* A device buffer whose PHYSICAL backing is `chunk_bytes`-sized allocations taken alternately from `groups` regions of the
 * HBM that a transient allocation of `spacer_bytes` pushed apart, mapped into one contiguous virtual range (HIP virtual
 * memory management).  Kernels whose stores are spread over their whole output (scans along Z, marches, fills) write
 * 15-20 % faster into such a buffer than into one contiguous physical block (tools/placement_probe.py).
 * chunk_bytes 0: 64 MiB.  Free with xg_scatter_free only. */
int xg_scatter_alloc(void** ptr, uint64_t bytes, uint64_t chunk_bytes, int groups, uint64_t spacer_bytes);
int xg_scatter_free(void* ptr);
/* The same as an allocator PLUG-IN with the signature PyTorch's `torch.cuda.memory.CUDAPluggableAllocator` expects
 * (`void* alloc(ssize_t, int device, stream)`, `void free(void*, ssize_t, int device, stream)`): xgcm_amd.device feeds a
 * torch MemPool with it and allocates operator outputs of 256 MB or more there.  Chunk size: XG_SCATTER_CHUNK_MB (64). */
/* scattered buffers made so far, bytes of them alive, and requests xg_pool_alloc had to serve with plain hipMalloc */
int xg_scatter_stats(uint64_t* buffers_made, uint64_t* live_bytes, uint64_t* pool_fallbacks);
/* Not every scattered buffer is a good one (round 6: of ten alive in one process four to seven carry a scan along Z at 1.62 -
 * 1.69 ms, the others at 1.86 - 2.04, each the same every time).  xg_scatter_grade times a write-only fill that walks 64 equal
 * slices of the buffer side by side -- the scan's store pattern -- against a flat fill of the same bytes (about 2 ms for 5 GB;
 * the buffer's contents are overwritten): `ratio` 1.03 - 1.09 = good, 1.17 - 1.38 = bad.  xg_pool_alloc grades every buffer
 * of 1 GiB or more it creates, parks bad ones while it tries again (up to 8 times; XG_SCATTER_TRIES) and keeps the best
 * (XG_SCATTER_GRADE_PCT, 112; 0 switches grading off); xg_scatter_grade_stats: requests graded, buffers rejected on the way. */
int xg_scatter_grade(void* ptr, uint64_t bytes, double* ratio);
int xg_scatter_grade_stats(uint64_t* graded, uint64_t* rejected);
void* xg_pool_alloc(ssize_t size, int device, void* stream);
void xg_pool_free(void* ptr, ssize_t size, int device, void* stream);
/* page-lock / release a range of HOST memory in place (hipHostRegister): asynchronous copies to and from it then run at
 * the link rate and overlap with kernels (xgcm_amd/streaming.py locks the record blocks of a host array ahead of its
 * copies).  A range that cannot be locked (a read-only file mapping ...) returns XG_ERR_HIP and leaves NO pending HIP
 * error behind; such memory still copies, through the runtime's pageable path. */
int xg_pin_host(void* ptr, uint64_t bytes);
int xg_unpin_host(void* ptr);
/* a stream of the library's own (xgcm_amd.graphs.capture records on one: the chained kernels keep their workspace per
 * stream, so a captured graph never shares it with another capture or with eager calls) */
int xg_stream_create(void** stream);
int xg_stream_destroy(void* stream); /* synchronises the DEVICE (a graph captured on the stream may be replaying elsewhere), releases the stream's chained-kernel workspace, destroys */
/* The long strided-axis scans / weighted reductions run as CHAINED flat launches (chunks of a column hand their running
 * sum on through one XCD's L2).  A chunk that gives up waiting for its predecessor (scan_chain_spin polls) cannot
 * damage a result: every chained launch is followed on the same stream by its marching twin, which runs only if a
 * chunk gave up and then redoes the whole call from the untouched inputs -- callers always read correct data, in
 * stream order.  The event is reported here: *gave_up = 1 once any chunk has given up (sticky; the library plans
 * marching kernels from then on), *redone = number of launches redone so far.  xg_chain_rearm() clears the sticky
 * word (a device whose workgroup -> XCD mapping had not been confirmed is probed again at its next chained launch).
 * Either pointer may be NULL. */
int xg_chain_status(int* gave_up, int* redone);
int xg_chain_rearm(void);
/* hipEvent helpers so hosts without a HIP binding can time kernels on `stream` */
int xg_event_create(void** ev);
int xg_event_record(void* ev, void* stream);
int xg_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int xg_event_destroy(void* ev);
/* Reverse the byte order of `nelem` elements of 2, 4 or 8 bytes in place (device buffer, 16-byte aligned): blocks read raw
 * from big-endian files -- MITgcm's MDS .data, NetCDF-3 -- and any numpy array of non-native byte order handed to an
 * operator (`np.fromfile(..., ">f4")`; the reference's numpy bodies accept them, xgcm/gridops.py:23-24,76-77, and
 * `np.pad` keeps the dtype, xgcm/padding.py:610-615) cross PCIe as raw bytes and are swapped on the GPU.  The reference
 * otherwise gets decoded native arrays from xarray's backends (xgcm/grid.py:786-818 walks their dask chunks). */
int xg_bswap(void* data, uint64_t nelem, int elem_bytes, void* stream);
/* Cells of a float32 / float64 device buffer (elem_bytes 4 / 8) that equal `value` become NaN, in place: the _FillValue /
 * missing_value of a file variable, which xarray's mask_and_scale decoding turns into NaN before the reference sees the
 * array (so that skipna reductions and cumsum skip land / missing cells). */
int xg_mask_value(void* data, uint64_t nelem, int elem_bytes, double value, void* stream);

/* ---- fused pad + two-point stencil along one axis -------------------------------------- */
/* out[.., i, ..] = OP(P[i], P[i+1]) / m_out,  P = pad(in * m_in, (pad_lo, pad_hi), bc, fill)
 * along `axis`;  pad_lo, pad_hi in {0,1};  n_out must equal shape[axis] + pad_lo + pad_hi - 1.
 * `out` has `shape` with shape[axis] replaced by n_out.  m_in_strides are per-dim strides in
 * the INPUT coordinate system, m_out_strides in the OUTPUT coordinate system.
 * The fill halo is NOT multiplied by m_in (the reference pads after the product). */
int xg_stencil1d_f64(int op, const double* in, double* out, const int64_t* shape, int ndim,
                     int axis, int64_t n_out, int pad_lo, int pad_hi, int bc, double fill,
                     const double* m_in, const int64_t* m_in_strides, const double* m_out,
                     const int64_t* m_out_strides, void* stream);

/* Same operator on a complex topology (face connections xgcm/padding.py:260-572, north fold
 * :619-762): the halo cells do not follow from `in` by a wrap / clamp / constant rule, so the
 * caller gathers them first (xg_gather_f64 over the halo cells only) into `halo`, an array shaped
 * like `in` with the op axis shortened to pad_lo + pad_hi (low halo first).  The kernels are the
 * ones of xg_stencil1d_f64; `in` is read once, no padded copy of the field is ever made. */
int xg_stencil1d_halo_f64(int op, const double* in, const double* halo, double* out,
                          const int64_t* shape, int ndim, int axis, int64_t n_out, int pad_lo,
                          int pad_hi, const double* m_out, const int64_t* m_out_strides,
                          void* stream);

/* The same with an INPUT metric (`metric_weighted` operators on a complex topology, xgcm/grid.py:804-808: the reference
 * multiplies, THEN pads the product through the topology): `in` is multiplied by m_in inside the kernel; `halo` holds the
 * halo cells of the PRODUCT in * m_in, which the caller forms from two halo-slab gathers -- gather(in) * gather(m_in), the
 * very operands the reference multiplies -- and which are not weighted again.  One pass over the field instead of a
 * product pass plus the operator. */
int xg_stencil1d_halo_w_f64(int op, const double* in, const double* halo, double* out,
                            const int64_t* shape, int ndim, int axis, int64_t n_out, int pad_lo,
                            int pad_hi, const double* m_in, const int64_t* m_in_strides,
                            const double* m_out, const int64_t* m_out_strides, void* stream);

/* ---- prefix sum along one axis with the reference's trim/pad folded in ------------------ */
/* c = inclusive cumsum of (in * m_in) along axis (from the high end if `reverse`; NaN counted
 * as 0 if `skipna`);  t = c[trim_lo : n - trim_hi];  out = pad(t, (pad_lo, pad_hi), bc, fill)
 * / m_out.  Output length along axis: n - trim_lo - trim_hi + pad_lo + pad_hi.
 * trim_*, pad_* in {0,1}.  The pad acts on the CUMULATIVE values (xgcm/grid.py:1385-1391). */
int xg_cumsum1d_f64(const double* in, double* out, const int64_t* shape, int ndim, int axis,
                    int reverse, int skipna, int trim_lo, int trim_hi, int pad_lo, int pad_hi,
                    int bc, double fill, const double* m_in, const int64_t* m_in_strides,
                    const double* m_out, const int64_t* m_out_strides, void* stream);

/* ---- weighted sum along one axis -------------------------------------------------------- */
/* out = sum_k (in * w)[.., k, ..] with `axis` removed from the output shape; NaN products
 * count as 0 if `skipna`.  Along a non-last axis the sum runs sequentially k = 0..n-1 per
 * output cell (bit-identical to numpy); along the last axis it is a lane-strided tree.
 * `skipna` also selects the two denominators of a weighted mean (xarray's
 * `da.weighted(w).mean`, xgcm/grid.py:1681-1685) in the same single pass: 2 = sum of the weights of
 * the valid (non-NaN) cells of `in`, 3 = sum of the weights of all cells -- and the mean itself, numerator
 * and denominator marching together so that `in` is read ONCE: 4 = (mode 1) / (mode 2), the NaN-skipping
 * weighted mean of Grid.average; 5 = (mode 0) / (mode 3).  Same sums in the same order as the separate
 * modes, one IEEE division at the end: bit-identical to computing the two sums apart and dividing.
 * 6 / 7 = the two sums of 4 / 5 written side by side instead of divided: `out` holds 2 N cells, numerators in
 * out[0 : N], denominators in out[N : 2N] (N = cells of the reduced array) -- the first, full-size pass of a
 * mean over SEVERAL dims; the caller reduces both halves over the remaining dims and divides. */
int xg_reduce1d_f64(const double* in, double* out, const int64_t* shape, int ndim, int axis,
                    int skipna, const double* w, const int64_t* w_strides, void* stream);

/* ---- generic N-D pad (any widths), axes applied in `order` like the reference's loop ---- */
/* out shape[d] = shape[d] + lo[d] + hi[d].  order[ndim] lists axes in application order
 * (NULL = 0..ndim-1); later-applied axes see the halos of earlier ones (numpy.pad chain). */
int xg_pad_f64(const double* in, double* out, const int64_t* shape, int ndim, const int64_t* lo,
               const int64_t* hi, const int* bc, const double* fill, const int* order,
               void* stream);

/* ---- halo gather through a token map (north fold, face connections) --------------------- */
/* `mapped[d]` marks the dims the padding procedure touches (face dim, the padded axes' dims);
 * `tokens` holds one int64 per cell of the PADDED mapped dims (row-major, out dim order),
 * shared by every index of the unmapped dims:
 *     |t| in [1, 2^62):  source element k = |t| - 1, counted row-major over the mapped dims of
 *                        `in` (k < P_in) or of `partner` (k - P_in, in partner's own dim order)
 *     |t| >= 2^62:       fills[|t| - 2^62]
 *     t < 0:             the value is negated (vector component across a fold / reversed link)
 * Interior cells (lo[d] <= c[d] < lo[d] + in_shape[d] on every mapped dim) copy the input and
 * never read the map.  `partner` (nullable) is the other vector component; its dim k
 * corresponds to out dim partner_perm[k]; unmapped dims must agree in length.
 * out_shape[d] == in_shape[d] on unmapped dims. */
int xg_gather_f64(const double* in, const double* partner, double* out, const int64_t* in_shape,
                  const int64_t* partner_shape, const int64_t* out_shape, int ndim,
                  const int* mapped, const int* partner_perm, const int64_t* lo,
                  const int64_t* tokens, int64_t n_tokens, const double* fills, int n_fills,
                  void* stream);

/* Halo cells written IN PLACE: `out` (shape[ndim], its interior already there) receives along `axis` the slab `halo`
 * -- shaped like `out` with the axis shortened to pad_lo + pad_hi, low halo first -- in its first pad_lo and last pad_hi
 * cells.  Grid.cumsum on a connected axis (xgcm/grid.py:1385-1395 pads the CUMULATIVE field through the topology): the
 * scan writes the padded layout in one pass (xg_cumsum1d with XG_BC_FILL as a placeholder), the halo cells are gathered
 * from that buffer (xg_gather over the halo cells only) and put in place -- the padded copy the reference makes is gone. */
int xg_halo_put_f64(const double* halo, double* out, const int64_t* shape, int ndim, int axis,
                    int pad_lo, int pad_hi, void* stream);

/* ---- vertical coordinate transform (xgcm/transform.py:15-142) ----------------------------- */
/* Per column along `axis` of `phi` (shape[ndim], C-contiguous):
 *   linear:       out[.., i, ..] = numpy.interp(target[i], theta[:], phi[:]) with the reference's
 *                 pre-steps: optional log of theta/target (`logarithmic`), flip when theta decreases
 *                 (first vs last non-NaN value; skipped if `bypass_checks`), NaN outside
 *                 [nanmin(theta), nanmax(theta)] if `mask_edges`.  numpy's search (carried guess,
 *                 probes, bisection) is reproduced, so NaN / duplicate thetas give numpy's answer.
 *   conservative: phi (n cells) is spread over the bins [bins[j], bins[j+1]) by the overlap of
 *                 each cell's theta range (theta has n + 1 vertices along `axis`), contributions
 *                 added in cell order; bins increasing (the host flips decreasing targets).
 * `theta` / `target` are addressed through element strides per dim of `shape` (0 = broadcast;
 * the entry at `axis` is the step between levels).  `out` has `shape` with `axis` -> m levels
 * (linear) or n_edges - 1 bins (conservative): the new dim stays where the axis was. */
int xg_transform_linear_f64(const double* phi, const double* theta, const int64_t* theta_strides,
                            const double* target, const int64_t* target_strides, int64_t m,
                            double* out, const int64_t* shape, int ndim, int axis, int mask_edges,
                            int bypass_checks, int logarithmic, void* stream);
int xg_transform_conservative_f64(const double* phi, const double* theta,
                                  const int64_t* theta_strides, const double* bins,
                                  int64_t n_edges, double* out, const int64_t* shape, int ndim,
                                  int axis, void* stream);

/* ---- broadcasting elementwise arithmetic ------------------------------------------------ */
/* out[idx] = a[idx . a_strides] OP b[idx . b_strides] over `shape` (out C-contiguous). */
int xg_binary_f64(int op, const double* a, const int64_t* a_strides, const double* b,
                  const int64_t* b_strides, double* out, const int64_t* shape, int ndim,
                  void* stream);

/* ---- fused relative vorticity on a C-grid ---------------------------------------------- */
/* out[..,j,i] = ((v[..,j,i] - v[..,j,i-1]) - (u[..,j,i] - u[..,j-1,i])) / area[..,j,i]
 * for arrays of identical `shape` (.., Y, X), ndim >= 2; the i-1 / j-1 halos follow bc_x/bc_y
 * (center->left diffs, padding_width (1,0)).  area (nullable) uses broadcast strides. */
int xg_vorticity_f64(const double* u, const double* v, const double* area,
                     const int64_t* area_strides, double* out, const int64_t* shape, int ndim,
                     int bc_x, double fill_x, int bc_y, double fill_y, void* stream);

/* ---- fused horizontal divergence (docs/ufunc_examples.md "Divergence") -------------------- */
/* out[..,j,i] = ((u[..,j,i+1] - u[..,j,i]) + (v[..,j+1,i] - v[..,j,i])) / area: both differences
 * left -> center (padding_width (0,1)), halos per bc_x / bc_y as in xg_stencil1d; same argument
 * list as xg_vorticity_f64; bit-identical to (diff(u,X) + diff(v,Y)) / area run operator by operator. */
int xg_divergence_f64(const double* u, const double* v, const double* area,
                     const int64_t* area_strides, double* out, const int64_t* shape, int ndim,
                     int bc_x, double fill_x, int bc_y, double fill_y, void* stream);

/* ---- fused gradient and advective flux (docs/ufunc_examples.md "Gradient", "Advection") ------ */
/* One centre field in, two staggered fields out (both center -> left, padding_width (1,0) on X and Y):
 *   gradient: out_x = (a[j,i] - a[j,i-1]) / mx,   out_y = (a[j,i] - a[j-1,i]) / my   (mx / my optional metrics
 *             at the output positions, broadcast strides; NULL = plain differences)
 *   flux:     out_x = u * (t[j,i-1] + t[j,i]) / 2, out_y = v * (t[j-1,i] + t[j,i]) / 2
 * bit-identical to the operator chains (diff / derivative, interp then multiply), the field read once. */
int xg_gradient_f64(const double* a, double* out_x, double* out_y, const int64_t* shape, int ndim,
                    int bc_x, double fill_x, int bc_y, double fill_y, const double* mx,
                    const int64_t* mx_strides, const double* my, const int64_t* my_strides,
                    void* stream);
int xg_flux_f64(const double* u, const double* v, const double* t, double* out_x, double* out_y,
                const int64_t* shape, int ndim, int bc_x, double fill_x, int bc_y, double fill_y,
                void* stream);

/* gradient / flux on a complex topology: an axis whose mode is XG_BC_HALO takes the one-cell halo of
 * the centre field from a pre-gathered slab, halo_x (..., Y) = the column left of i = 0, halo_y
 * (..., X) = the row below j = 0 (xg_gather_f64 over the halo-only plane), like xg_vorticity_halo_f64. */
int xg_gradient_halo_f64(const double* a, const double* halo_x, const double* halo_y, double* out_x,
                         double* out_y, const int64_t* shape, int ndim, int bc_x, double fill_x,
                         int bc_y, double fill_y, const double* mx, const int64_t* mx_strides,
                         const double* my, const int64_t* my_strides, void* stream);
int xg_flux_halo_f64(const double* u, const double* v, const double* t, const double* halo_x,
                     const double* halo_y, double* out_x, double* out_y, const int64_t* shape,
                     int ndim, int bc_x, double fill_x, int bc_y, double fill_y, void* stream);

/* The two fused operators on a complex topology (face connections, north fold): an axis whose
 * boundary mode is XG_BC_HALO takes its one-cell halo from a pre-gathered slab (xg_gather_f64 over
 * the halo cells, vector-component rules included) instead of the array itself:
 *   vorticity:  halo_x[outer, Y] = v left of column 0,   halo_y[outer, X] = u below row 0
 *   divergence: halo_x[outer, Y] = u right of the last column, halo_y[outer, X] = v above the last row
 * (NULL for an axis with an ordinary mode). */
int xg_vorticity_halo_f64(const double* u, const double* v, const double* halo_x, const double* halo_y,
                          const double* area, const int64_t* area_strides, double* out,
                          const int64_t* shape, int ndim, int bc_x, double fill_x, int bc_y,
                          double fill_y, void* stream);
int xg_divergence_halo_f64(const double* u, const double* v, const double* halo_x,
                           const double* halo_y, const double* area, const int64_t* area_strides,
                           double* out, const int64_t* shape, int ndim, int bc_x, double fill_x,
                           int bc_y, double fill_y, void* stream);

/* ---- the same two-point operator along the last TWO axes in one pass -------------------- */
/* out = OP_second(pad(OP_first(pad(in)))) for (.., Y, X) arrays, order 0: X then Y, 1: Y then X;
 * replaces two sequential apply_as_grid_ufunc passes of Grid.interp/diff/min/max(da, [ax1, ax2])
 * (xgcm/grid.py:798-828; the TODO at :798-800 asks for this fusion).  Length-preserving pads
 * ((1,0) or (0,1)) on both axes, nx even; bit-identical to two xg_stencil1d_f64 calls. */
int xg_stencil2d_f64(int op, const double* in, double* out, const int64_t* shape, int ndim, int order,
                     int padx_lo, int padx_hi, int bc_x, double fill_x, int pady_lo, int pady_hi,
                     int bc_y, double fill_y, void* stream);
/* The same with `metric_weighted` on both axes and one metric set (Grid.interp(da, ["X", "Y"], metric_weighted=("X", "Y")):
 * per axis the reference multiplies by the metric at the current position, applies the operator and divides by the
 * metric at the new position, xgcm/grid.py:804-828).  Three contiguous (ny, nx) planes shared by all outer indices:
 * m_in at the input positions, m_mid at the positions between the two axes (divisor of the first step and factor of
 * the second), m_out at the output positions.  Bit-identical to the two metric-carrying xg_stencil1d_f64 calls. */
int xg_stencil2d_metric_f64(int op, const double* in, double* out, const int64_t* shape, int ndim, int order,
                            int padx_lo, int padx_hi, int bc_x, double fill_x, int pady_lo, int pady_hi,
                            int bc_y, double fill_y, const double* m_in, const double* m_mid,
                            const double* m_out, void* stream);

/* ---- synthetic fields, bit-identical to oracle/refimpl.py:synthetic --------------------- */
/* out[i] = u * scale + shift,  u = (splitmix64_mix(i + offset + seed*0x9E3779B97F4A7C15) >> 11)
 * * 2^-53,  i = 0..n-1. */
int xg_fill_synthetic_f64(double* out, int64_t n, uint64_t seed, uint64_t offset, double scale,
                          double shift, void* stream);


/* ---- float32 variants ------------------------------------------------------------------- */
/* Same semantics, same argument order; arrays, metrics and fill values are float.  The reference
 * computes in the input's dtype (numpy), so float32 fields (e.g. MITgcm / LLC4320 output) stay
 * float32 and results are bit-identical to numpy's float32 arithmetic.  Lanes move 16 bytes in both
 * builds (2 doubles / 4 floats), so float32 runs at the same byte rate = twice the f64 cell rate. */
int xg_stencil1d_f32(int op, const float* in, float* out, const int64_t* shape, int ndim, int axis,
                     int64_t n_out, int pad_lo, int pad_hi, int bc, float fill, const float* m_in,
                     const int64_t* m_in_strides, const float* m_out, const int64_t* m_out_strides,
                     void* stream);
int xg_stencil1d_halo_f32(int op, const float* in, const float* halo, float* out,
                          const int64_t* shape, int ndim, int axis, int64_t n_out, int pad_lo,
                          int pad_hi, const float* m_out, const int64_t* m_out_strides, void* stream);
int xg_stencil1d_halo_w_f32(int op, const float* in, const float* halo, float* out,
                            const int64_t* shape, int ndim, int axis, int64_t n_out, int pad_lo,
                            int pad_hi, const float* m_in, const int64_t* m_in_strides,
                            const float* m_out, const int64_t* m_out_strides, void* stream);
int xg_cumsum1d_f32(const float* in, float* out, const int64_t* shape, int ndim, int axis,
                    int reverse, int skipna, int trim_lo, int trim_hi, int pad_lo, int pad_hi,
                    int bc, float fill, const float* m_in, const int64_t* m_in_strides,
                    const float* m_out, const int64_t* m_out_strides, void* stream);
int xg_reduce1d_f32(const float* in, float* out, const int64_t* shape, int ndim, int axis,
                    int skipna, const float* w, const int64_t* w_strides, void* stream);
int xg_pad_f32(const float* in, float* out, const int64_t* shape, int ndim, const int64_t* lo,
               const int64_t* hi, const int* bc, const float* fill, const int* order, void* stream);
int xg_gather_f32(const float* in, const float* partner, float* out, const int64_t* in_shape,
                  const int64_t* partner_shape, const int64_t* out_shape, int ndim,
                  const int* mapped, const int* partner_perm, const int64_t* lo,
                  const int64_t* tokens, int64_t n_tokens, const float* fills, int n_fills,
                  void* stream);
int xg_halo_put_f32(const float* halo, float* out, const int64_t* shape, int ndim, int axis, int pad_lo, int pad_hi,
                    void* stream);
int xg_transform_linear_f32(const float* phi, const float* theta, const int64_t* theta_strides,
                            const float* target, const int64_t* target_strides, int64_t m,
                            float* out, const int64_t* shape, int ndim, int axis, int mask_edges,
                            int bypass_checks, int logarithmic, void* stream);
int xg_transform_conservative_f32(const float* phi, const float* theta,
                                  const int64_t* theta_strides, const float* bins, int64_t n_edges,
                                  float* out, const int64_t* shape, int ndim, int axis,
                                  void* stream);
int xg_binary_f32(int op, const float* a, const int64_t* a_strides, const float* b,
                  const int64_t* b_strides, float* out, const int64_t* shape, int ndim,
                  void* stream);
int xg_vorticity_f32(const float* u, const float* v, const float* area,
                     const int64_t* area_strides, float* out, const int64_t* shape, int ndim,
                     int bc_x, float fill_x, int bc_y, float fill_y, void* stream);
int xg_divergence_f32(const float* u, const float* v, const float* area,
                     const int64_t* area_strides, float* out, const int64_t* shape, int ndim,
                     int bc_x, float fill_x, int bc_y, float fill_y, void* stream);
int xg_gradient_f32(const float* a, float* out_x, float* out_y, const int64_t* shape, int ndim,
                    int bc_x, float fill_x, int bc_y, float fill_y, const float* mx,
                    const int64_t* mx_strides, const float* my, const int64_t* my_strides, void* stream);
int xg_flux_f32(const float* u, const float* v, const float* t, float* out_x, float* out_y,
                const int64_t* shape, int ndim, int bc_x, float fill_x, int bc_y, float fill_y,
                void* stream);
int xg_gradient_halo_f32(const float* a, const float* halo_x, const float* halo_y, float* out_x,
                         float* out_y, const int64_t* shape, int ndim, int bc_x, float fill_x, int bc_y,
                         float fill_y, const float* mx, const int64_t* mx_strides, const float* my,
                         const int64_t* my_strides, void* stream);
int xg_flux_halo_f32(const float* u, const float* v, const float* t, const float* halo_x,
                     const float* halo_y, float* out_x, float* out_y, const int64_t* shape, int ndim,
                     int bc_x, float fill_x, int bc_y, float fill_y, void* stream);
int xg_vorticity_halo_f32(const float* u, const float* v, const float* halo_x, const float* halo_y,
                          const float* area, const int64_t* area_strides, float* out,
                          const int64_t* shape, int ndim, int bc_x, float fill_x, int bc_y,
                          float fill_y, void* stream);
int xg_divergence_halo_f32(const float* u, const float* v, const float* halo_x, const float* halo_y,
                           const float* area, const int64_t* area_strides, float* out,
                           const int64_t* shape, int ndim, int bc_x, float fill_x, int bc_y,
                           float fill_y, void* stream);
int xg_stencil2d_f32(int op, const float* in, float* out, const int64_t* shape, int ndim, int order,
                     int padx_lo, int padx_hi, int bc_x, float fill_x, int pady_lo, int pady_hi,
                     int bc_y, float fill_y, void* stream);
int xg_stencil2d_metric_f32(int op, const float* in, float* out, const int64_t* shape, int ndim, int order,
                     int padx_lo, int padx_hi, int bc_x, float fill_x, int pady_lo, int pady_hi,
                     int bc_y, float fill_y, const float* m_in, const float* m_mid,
                            const float* m_out, void* stream);
/* value formed in float64 exactly as the _f64 variant, then rounded once to float */
int xg_fill_synthetic_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, double scale,
                          double shift, void* stream);

/* ---- integer variants (int64 / int32 lanes, two's complement, wrap-around) ------------------ */
/* numpy keeps integer arrays integral through diff / min / max / cumsum / pad and wraps modulo 2^bits
 * (xgcm/gridops.py:23-24,123-126,172-175,227-278 run in the array's own dtype; xgcm/padding.py:610-615: numpy.pad keeps
 * it and casts the fill value).  Same kernels, same argument order as the _f64 entry points, with these differences:
 *   - arithmetic is modulo 2^64 (_i64) or 2^32 (_i32); int64 / uint64 arrays run on _i64 and int32 / uint32 arrays on
 *     _i32 as they are (an unsigned array shares its bits with the signed lanes for diff / cumsum / pad; its min / max
 *     are XG_OP_MINU / XG_OP_MAXU); narrower and bool arrays are widened to int32 lanes by xg_convert, computed, and
 *     narrowed back (wrap modulo 2^bits).  Scans and sums exist as _i64 only: numpy accumulates every integer dtype
 *     in 64 bits;
 *   - XG_OP_INTERP returns the wrapped SUM l + r: `(a[1:] + a[:-1]) / 2.0` leaves the integer domain, the host casts
 *     the sum to the array's dtype width and halves it in float64 (xg_convert with via_type and scale 0.5);
 *   - metric / weight arguments must be NULL and reductions are plain sums (skipna 0 / 1, no NaN exists): a metric is
 *     float, numpy promotes `int * float64` BEFORE the operator, so the host converts the field first and runs _f64;
 *   - XG_BIN_DIV is refused (true division is float);  fill values are int64 (numpy.pad's cast, done by the host). */
int xg_stencil1d_i64(int op, const int64_t* in, int64_t* out, const int64_t* shape, int ndim, int axis,
                     int64_t n_out, int pad_lo, int pad_hi, int bc, int64_t fill, const int64_t* m_in,
                     const int64_t* m_in_strides, const int64_t* m_out, const int64_t* m_out_strides,
                     void* stream);
int xg_stencil1d_halo_i64(int op, const int64_t* in, const int64_t* halo, int64_t* out,
                          const int64_t* shape, int ndim, int axis, int64_t n_out, int pad_lo,
                          int pad_hi, const int64_t* m_out, const int64_t* m_out_strides, void* stream);
int xg_cumsum1d_i64(const int64_t* in, int64_t* out, const int64_t* shape, int ndim, int axis,
                    int reverse, int skipna, int trim_lo, int trim_hi, int pad_lo, int pad_hi,
                    int bc, int64_t fill, const int64_t* m_in, const int64_t* m_in_strides,
                    const int64_t* m_out, const int64_t* m_out_strides, void* stream);
int xg_reduce1d_i64(const int64_t* in, int64_t* out, const int64_t* shape, int ndim, int axis,
                    int skipna, const int64_t* w, const int64_t* w_strides, void* stream);
int xg_pad_i64(const int64_t* in, int64_t* out, const int64_t* shape, int ndim, const int64_t* lo,
               const int64_t* hi, const int* bc, const int64_t* fill, const int* order, void* stream);
int xg_gather_i64(const int64_t* in, const int64_t* partner, int64_t* out, const int64_t* in_shape,
                  const int64_t* partner_shape, const int64_t* out_shape, int ndim,
                  const int* mapped, const int* partner_perm, const int64_t* lo,
                  const int64_t* tokens, int64_t n_tokens, const int64_t* fills, int n_fills,
                  void* stream);
int xg_halo_put_i64(const int64_t* halo, int64_t* out, const int64_t* shape, int ndim, int axis, int pad_lo, int pad_hi,
                    void* stream);
int xg_binary_i64(int op, const int64_t* a, const int64_t* a_strides, const int64_t* b,
                  const int64_t* b_strides, int64_t* out, const int64_t* shape, int ndim,
                  void* stream);

/* the same on int32 lanes (int32 / uint32 arrays as they are, bool / 8 / 16-bit arrays widened to 4 bytes): the entry
 * points whose result keeps the array's width */
int xg_stencil1d_i32(int op, const int32_t* in, int32_t* out, const int64_t* shape, int ndim, int axis,
                     int64_t n_out, int pad_lo, int pad_hi, int bc, int32_t fill, const int32_t* m_in,
                     const int64_t* m_in_strides, const int32_t* m_out, const int64_t* m_out_strides,
                     void* stream);
int xg_stencil1d_halo_i32(int op, const int32_t* in, const int32_t* halo, int32_t* out,
                          const int64_t* shape, int ndim, int axis, int64_t n_out, int pad_lo,
                          int pad_hi, const int32_t* m_out, const int64_t* m_out_strides, void* stream);
int xg_pad_i32(const int32_t* in, int32_t* out, const int64_t* shape, int ndim, const int64_t* lo,
               const int64_t* hi, const int* bc, const int32_t* fill, const int* order, void* stream);
int xg_gather_i32(const int32_t* in, const int32_t* partner, int32_t* out, const int64_t* in_shape,
                  const int64_t* partner_shape, const int64_t* out_shape, int ndim,
                  const int* mapped, const int* partner_perm, const int64_t* lo,
                  const int64_t* tokens, int64_t n_tokens, const int32_t* fills, int n_fills,
                  void* stream);
int xg_halo_put_i32(const int32_t* halo, int32_t* out, const int64_t* shape, int ndim, int axis, int pad_lo, int pad_hi,
                    void* stream);
int xg_binary_i32(int op, const int32_t* a, const int64_t* a_strides, const int32_t* b,
                  const int64_t* b_strides, int32_t* out, const int64_t* shape, int ndim,
                  void* stream);

/* ---- strided N-d copy: the data movement around the operators ------------------------------ */
/* dst[i0, i1, ...] = src[i0, i1, ...] for every index of `shape`, with ELEMENT strides per dim on both sides: the
 * materialisation of a transposed / flipped / broadcast / sliced view, which the reference leaves to numpy behind
 * `DataArray.transpose` (xgcm/grid_ufunc.py:56-103), `[..., ::-1]` (xgcm/transform.py:180-192) and `xr.concat`.
 * Source strides may be negative (flip: `src` then points at the element with index 0, i.e. the END of the flipped
 * range) or 0 (broadcast); destination strides must be positive and must not make two indices share a cell.
 * elem_bytes 1, 2, 4 or 8 (the bytes are moved, never interpreted).  Rows that run along the same unit-stride dim on both
 * sides move as 16-byte lane vectors, true transposes through 32 x 32 LDS tiles, anything else one element per lane. */
int xg_copy_nd(const void* src, const int64_t* src_strides, void* dst, const int64_t* dst_strides,
               const int64_t* shape, int ndim, int elem_bytes, void* stream);

/* ---- element type conversion (numpy `astype`) --------------------------------------------- */
typedef enum xg_dtype {
  XG_T_BOOL = 0, XG_T_I8 = 1, XG_T_I16 = 2, XG_T_I32 = 3, XG_T_I64 = 4,
  XG_T_U8 = 5, XG_T_U16 = 6, XG_T_U32 = 7, XG_T_U64 = 8, XG_T_F32 = 9, XG_T_F64 = 10,
  XG_T_F16 = 11 /* IEEE binary16: a STORAGE type -- float16 arrays are widened to float32 lanes and results narrowed */
} xg_dtype;
/* dst[i] = (dst_type) src[i] for n contiguous elements, with C / numpy `astype` rules: integer -> integer wraps modulo
 * 2^bits, integer -> float rounds to nearest, float -> integer truncates toward zero, bool reads / stores `!= 0`,
 * float -> narrower float rounds once to nearest-even (float16 overflows to inf, keeps subnormals).
 *   via_type  XG_T_* integer type or -1: an integer source value is first wrapped to that type's width and signedness --
 *             the value the narrow dtype would hold -- so a result computed on int64 lanes leaves as numpy's narrow-dtype
 *             arithmetic would have it (interp of an int8 array = convert(int64 sums, via int8, to f64, scale 0.5));
 *   scale     multiplies float destinations after the conversion (1.0: none; a power of two is exact);
 *   flags     bit 0, 64-bit integer types on both sides: flip the sign bit on the way (uint64 order <-> int64 order, for
 *             min / max of uint64 arrays through the signed kernels).
 * This is also numpy's promotion before `int_array * float64_metric` (xgcm/grid.py:804-808,1600).  src == dst is allowed
 * when the element sizes agree. */
int xg_convert(const void* src, int src_type, void* dst, int dst_type, uint64_t n, int via_type,
               double scale, int flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XGCM_HIP_H */
