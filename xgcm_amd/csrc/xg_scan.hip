// xg_scan.hip -- prefix sums (K5, K6, K6v) and weighted reductions (K4, K4b) along one axis
// Part of libxgcm_hip.so; compiled twice (real = double / -DXG_F32), see xg_common.hpp.

#include "xg_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// K5: cumsum along a STRIDED axis (one lane = one column pair, sequential => bit-exact with
// numpy.cumsum / nancumsum), trim/pad table folded into the output index, halo cells written
// from registers at the end of the march.
// Tried in round 3 and removed again (kernel in the history, commit "K5L level-major Z scan"): the same scan LEVEL-MAJOR --
// one generation of waves, each holding the running sums of 24 x-tiles in registers (rows through buffer descriptors,
// straight-line levels with exact vmcnt waits), so that the whole chip sweeps one level at a time like a copy.  Bit-exact,
// but no faster than this march in any state of the device: 0.69-0.72 against 0.72-0.76 of 8 TB/s for one record, equal
// from 4 records per launch on (profiles/history/r03c_*, r03i_ab_cumZ_records.jsonl) -- the number of DRAM streams and the
// generation tail are not what holds the march back.  The stand-alone probe tools/probes/levels_probe.hip (14 shapes: 8-32 tiles
// per wave, groups of 2 / 4, loads 1-5 groups ahead; profiles/history/r03p_*): slow-kind box 0.675-0.680 whatever the depth against
// 0.653 for the march and 0.82 for a copy; fast-kind box 0.738 against 0.766; the same waves loading only 0.83, storing
// only 0.77 -- i.e. neither prefetch depth nor occupancy is the lever, and the two directions together cost 8 % more than
// apart.  What the same measurements do show: the rate of THIS kernel on
// ONE box falls from 0.76 (a 1.7 ms launch) to 0.65 (8 records, 16 ms) with the length of the busy period, i.e. the
// "slow boxes" of round 2 are the sustained, power-managed state of every box.
// ------------------------------------------------------------------------------------------
struct ScanArgs {
  int reverse, skipna, trim_lo, trim_hi, pad_lo, pad_hi, bc;
  real fill;
};

#ifdef XG_I64
__device__ __forceinline__ real nan0(real v) { return v; }  // integers hold no NaN
#else
__device__ __forceinline__ real nan0(real v) { return (v != v) ? real(0) : v; }
#endif
__device__ __forceinline__ dv nan0(dv v) {
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = nan0(v[k]);
  return o;
}

#ifdef XG_F32
__device__ __forceinline__ hv nan0(hv v) { hv o; o[0] = nan0(v[0]); o[1] = nan0(v[1]); return o; }
#endif
// reduction input modes beyond plain / NaN-skipping sums (xg_reduce1d `skipna` argument):
// 2: every valid (non-NaN) cell counts as 1, NaN cells as 0 -> sum of the weights of the valid cells,
// 3: every cell counts as 1 -> sum of the weights (the two denominators of a weighted mean)
#ifdef XG_I64
__device__ __forceinline__ real as_count(real, int) { return real(1); }
#else
__device__ __forceinline__ real as_count(real v, int mode) { return (mode == 3 || v == v) ? real(1) : real(0); }
#endif
#ifdef XG_F32
__device__ __forceinline__ hv as_count(hv v, int mode) { hv o; o[0] = as_count(v[0], mode); o[1] = as_count(v[1], mode); return o; }
#endif
__device__ __forceinline__ dv as_count(dv v, int mode) {
  dv o;
#pragma unroll
  for (int k = 0; k < NV; ++k) o[k] = as_count(v[k], mode);
  return o;
}

// PIPE: the U loads of a lane form a rolling window -- every consumed row is replaced by the load of the row U
// steps ahead, so U - 1 loads stay in flight through the whole march instead of draining at every batch of U
// (few, long columns: cumsum along Y of (Z,Y,X) has ~4 waves per SIMD, occupancy cannot hide the drain).
template <int V, int MET, bool NTL, bool NTS, int U, bool PIPE>
__device__ __forceinline__ void cumsum_strided_body(
    const real* __restrict__ in, real* __restrict__ out, const Geo& g, u32 ntile, const ScanArgs& a,
    const real* __restrict__ m_in, const MIdx& mi, const real* __restrict__ m_out, const MIdx& mo, int band) {
  typedef typename VecT<V>::type T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  // U independent loads in flight per lane (the scan chain only consumes them)

  const u64 w = (band & 1) ? banded_wave_id() : wave_id();
  const bool pace = (band & 2) != 0;  // experiment: the 4 waves of a workgroup (adjacent x-tiles) re-align every window
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
  // the input metric of a row is loaded WITH the row (same window / batch): left inside `step` it was one
  // dependent L2 round trip per row of the march (cumint along Y with dy(Y,X): 40 % of 8 TB/s)
  auto step = [&](int64_t idx, T v, T w) {
    if (HAS_MI) v = v * w;
    if (a.skipna) v = nan0(v);
    acc = started ? acc + v : v;
    started = true;
    if (idx == first_kept) c_first = acc;
    if (idx == last_kept) c_last = acc;
    if (idx >= first_kept && idx <= last_kept) put(idx + shift, acc);
  };
  // (the form of the metric load -- one aligned vector or element by element -- is decided once per wave: see met_vec_all)
  auto march = [&](auto wvec) {
  constexpr bool WV = decltype(wvec)::value;
  auto ldw = [&](int64_t idx) -> T {
    if constexpr (WV) return *reinterpret_cast<const T*>(m_in + mi_base + idx * mi.axis);
    else return HAS_MI ? ldm<T>(m_in, mi_base + idx * mi.axis, mi_step) : splat<T>(real(1));
  };
  int64_t t = 0;
  if (PIPE) {
    auto row = [&](int64_t k) -> int64_t { return a.reverse ? n - 1 - k : k; };  // k-th row in scan order
    constexpr int UW = HAS_MI ? U : 1;
    T v[U], wv[UW];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < n) {
        v[u] = ldg<T, NTL>(pin + row(u) * inner);
        if (HAS_MI) wv[u % UW] = ldw(row(u));
      }
    for (; t + 2 * U <= n; t += U) {  // steady state: consume row t + u, refill its slot with row t + U + u
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T x = v[u], xw = wv[u % UW];
        v[u] = ldg<T, NTL>(pin + row(t + U + u) * inner);
        if (HAS_MI) wv[u % UW] = ldw(row(t + U + u));
        step(row(t + u), x, xw);
      }
      if (pace) __builtin_amdgcn_s_barrier();
    }
    for (; t < n; t += U) {  // the last one or two windows: refills and consumes guarded (wave-uniform tests)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T x = v[u], xw = wv[u % UW];
        if (t + U + u < n) {
          v[u] = ldg<T, NTL>(pin + row(t + U + u) * inner);
          if (HAS_MI) wv[u % UW] = ldw(row(t + U + u));
        }
        if (t + u < n) step(row(t + u), x, xw);
      }
    }
  } else {
  for (; t + U <= n; t += U) {
    constexpr int UW = HAS_MI ? U : 1;
    T v[U], wv[UW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t idx = a.reverse ? n - 1 - (t + u) : t + u;
      v[u] = ldg<T, NTL>(pin + idx * inner);
      if (HAS_MI) wv[u % UW] = ldw(idx);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(a.reverse ? n - 1 - (t + u) : t + u, v[u], wv[u % UW]);
  }
  for (; t < n; ++t) {
    int64_t idx = a.reverse ? n - 1 - t : t;
    step(idx, ldg<T, NTL>(pin + idx * inner), ldw(idx));
  }
  }
  };
  if (V > 1 && HAS_MI && met_vec_all<V>(m_in, mi_base, mi_step, mi.axis)) march(std::true_type{});
  else march(std::false_type{});
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
// (K5p, round 3: the march as loader / storer wave PAIRS -- one wave keeps 3 groups of 4 / 5 / 8 rows in flight and drops each
// landed group into one of two LDS stages, the other adds and stores, one LDS-only barrier per group, vmcnt(16..20) waits in
// the steady state -- built into the library with every option of the scan, bit-exact on the whole suite, and measured in one
// process on a slow-kind box: cumsum Z 0.6735 (this march) against 0.662-0.663 for all three shapes, 4 records in one launch
// 0.748 against 0.740.  The probe's +2.5 points (tools/probes/marchprobe.hip, round 2) were over a single wave WITHOUT the rolling
// window this march has since.  Removed again; profiles/EXPERIMENTS.md.)
template <int V, int MET, bool NTL, bool NTS, int U, bool PIPE = false>
__global__ __launch_bounds__(BLOCK) void k_cumsum_strided(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo, int band) {
  cumsum_strided_body<V, MET, NTL, NTS, U, PIPE>(in, out, g, ntile, a, m_in, mi, m_out, mo, band);
}

// ------------------------------------------------------------------------------------------
// K5c: cumsum along a STRIDED axis for few, LONG columns (cumsum along Y of (Z, Y, X): 4 288 marches of 2 400 rows)
// as a CHAINED FLAT launch.  A marching wave lives for the whole column: every load it waits for sits behind its own
// earlier stores in the in-order vmcnt queue, and the chip touches every level at once (150 read + write streams);
// measured with tools/probes/marchprobe.hip the march stays at 62-71 % of 8 TB/s (box to box) where a flat launch over the
// same bytes reaches 73-80 %.  Here a wave-task is (column = (outer index, x-tile), chunk c of R rows): it loads its
// R rows at once, waits for the running sum that chunk c - 1 of its column published, adds its rows IN SEQUENCE
// (the march's order: bit-identical), publishes its own last sum BEFORE issuing its R stores, and ends.
//   order: an XCD owns a band of consecutive columns, cut into sub-bands of W columns; inside a sub-band the tasks
//     run chunk-major, so W chains advance side by side and one XCD sweeps ~W / ntile levels at a time;
//   tickets: a workgroup takes the next 4 tasks of its band from a counter, so every task a wave can wait for is
//     already running (no reliance on dispatch order); the last taker resets the counter for the next launch;
//   hand-off: one 16-B slot per lane and ring position (2 deep), {payload low, epoch, payload high, epoch} with
//     epoch = c + 1: valid at 8-B granularity, so a torn 16-B access cannot pair a fresh epoch with a stale half;
//     written with a plain store (acknowledged by the XCD's L2 -- the chain lives behind ONE L2: agent-scope stores
//     are acknowledged by memory behind the whole write stream, 2.9 instead of 1.9 ms; release / acquire fences cost
//     a buffer_wbl2 / buffer_inv per task, 17 ms), polled with sc1 loads (bypass the L1); 0.3 us per hand-off when
//     idle (tools/probes/pingpong.hip), ~1 us under load.  The last chunk of a column zeroes its two slots: the workspace
//     is all-zero between launches (no per-launch memset, safe under graph replay);
//   the spin is bounded (`scan_chain_spin` polls): a wave that gives up poisons its column with NaN and raises the
//     stream's poison word, on which the rescue kernel queued behind every chained launch redoes the call (chain_wait).
// (Round 3, after K4L's success with weights through LDS: the same for this scan -- the 4 waves of a workgroup = 4
// consecutive levels of one x-tile and chunk, the chunk's 32 metric rows fetched once per workgroup into LDS instead of once
// per wave, i.e. 40 instead of 64 loads per task: cumint Y 0.666 -> 0.673, nothing (profiles/history/r03v_*).  Not kept: the chain's
// pace is set by its hand-offs and stores, not by the metric loads.)
// ------------------------------------------------------------------------------------------
struct ChainArgs {
  u32 nchunk, cpx, ncol, W, nblk;  // chunks per column, columns per XCD band, columns, sub-band width, workgroups per band
  u32 srow;                        // slots per ring row = lanes per row of the array
  u32 spin;                        // polls of a slot before giving up
  u32 tmaj;                        // 0: columns numbered (outer index, x-tile); n > 0: (x-tile, outer index) with n outer indices
  u32* ticket;
  u32* gave_up;                    // host-mapped: [0] sticky report, [1] launches redone
  u32* poison;                     // device: [0] run-if word of this launch's rescue kernel, [1] its workgroup count
  void* slots;
  u64 nslot;                       // 16-B slots in the workspace block (the rescue kernel scrubs them all)
};
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ u32x4 chain_pack(T v, u32 ep);
template <typename T> __device__ __forceinline__ T chain_unpack(u32x4 s);
#ifdef XG_F32
// (scalar temporaries: __builtin_bit_cast applied directly to an element of an ext_vector reads element 0)
template <> __device__ __forceinline__ u32x4 chain_pack<hv>(hv v, u32 ep) {
  const float f0 = v[0], f1 = v[1];
  u32x4 o = {__float_as_uint(f0), ep, __float_as_uint(f1), ep};
  return o;
}
template <> __device__ __forceinline__ hv chain_unpack<hv>(u32x4 s) {
  const u32 b0 = s[0], b1 = s[2];
  hv o;
  o[0] = __uint_as_float(b0);
  o[1] = __uint_as_float(b1);
  return o;
}
#else
template <> __device__ __forceinline__ u32x4 chain_pack<real>(real v, u32 ep) {
  const u64 b = __builtin_bit_cast(u64, v);
  u32x4 o = {(u32)b, ep, (u32)(b >> 32), ep};
  return o;
}
template <> __device__ __forceinline__ real chain_unpack<real>(u32x4 s) {
  const u64 b = (u64)s[0] | ((u64)s[2] << 32);
  return __builtin_bit_cast(real, b);
}
#endif

// the next wave-task of this workgroup's XCD band: chunk c of column (o32, tile); false when the band is exhausted
__device__ __forceinline__ bool chain_task(const ChainArgs& ch, u32 ntile, u32& c, u32& o32, u32& tile) {
  __shared__ u32 s_ticket;
  const u32 xcd = blockIdx.x & 7;
  if (threadIdx.x == 0) {
    const u32 t = atomicAdd(&ch.ticket[xcd * 32], 1u);  // one 128-B line per counter: each lives in its own XCD's L2
    if (t == ch.nblk - 1) __hip_atomic_store(&ch.ticket[xcd * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ticket = t;
  }
  __syncthreads();
  const u32 q = __builtin_amdgcn_readfirstlane(s_ticket * WPB + (threadIdx.x >> 6));
  const u32 col_lo = xcd * ch.cpx;
  if (col_lo >= ch.ncol) return false;
  const u32 col_hi = (ch.ncol - col_lo < ch.cpx) ? ch.ncol : col_lo + ch.cpx;
  const u32 ncols = col_hi - col_lo;
  if (q >= ncols * ch.nchunk) return false;
  u32 j = q / (ch.nchunk * ch.W);
  const u32 nsub = (ncols + ch.W - 1) / ch.W;
  if (j >= nsub) j = nsub - 1;
  const u32 sub_lo = col_lo + j * ch.W;
  const u32 w = (col_hi - sub_lo < ch.W) ? col_hi - sub_lo : ch.W;
  const u32 ql = q - j * ch.nchunk * ch.W;
  // (the divisions run on the vector unit; readfirstlane tells the compiler their results are wave-uniform again)
  c = __builtin_amdgcn_readfirstlane(ql / w);
  const u32 col = sub_lo + (ql - c * w);
  if (ch.tmaj) {  // x-tile-major: the levels of one x-tile are neighbours in the sequence (a metric shared by the levels)
    tile = __builtin_amdgcn_readfirstlane(col / ch.tmaj);
    o32 = col - tile * ch.tmaj;
  } else {
    o32 = __builtin_amdgcn_readfirstlane(col / ntile);
    tile = col - o32 * ntile;
  }
  return true;
}

// one hand-off slot: spin (bounded) until the epoch of both halves is `want`.  A wave that gives up raises the
// stream's poison word -- the rescue kernel queued behind this launch then redoes the whole call with the march --
// and carries on with NaN under a VALID epoch: its own rows and everything downstream in the column are visibly
// poisoned, and no later chunk spins behind it.
template <typename T>
__device__ __forceinline__ T chain_wait(const u32x4* src, u32 want, const ChainArgs& ch) {
  u32x4 got;
  u32 tries = 0;
  do {
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(got) : "v"(src) : "memory");
  } while ((got[1] != want || got[3] != want) && ++tries < ch.spin);
  if (got[1] != want || got[3] != want) {
    __hip_atomic_store(ch.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ch.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return splat<T>(poison_value());
  }
  return chain_unpack<T>(got);
}

// The rescue kernels: the marching twin of a chained launch, queued right behind it on the same stream with the
// stream's poison word as run-if.  Zero (in practice always): every workgroup leaves after one scalar load.  Non-zero:
// the inputs are untouched, so the march redoes the whole call (same additions in the same order: the bits the chain
// would have produced), every thread helps zeroing the hand-off slots -- a chunk that published after its successor
// had given up leaves a stale epoch behind -- and the last workgroup to finish clears the word and counts the event.
struct Rescue { u32* poison; u32* gave_up; u32x4* slots; u64 nslot; };
__device__ __forceinline__ bool rescue_begin(const Rescue& rs) {
  if (__hip_atomic_load(rs.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return false;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < rs.nslot; i += (u64)gridDim.x * BLOCK) rs.slots[i] = zero;
  return true;
}
__device__ __forceinline__ void rescue_end(const Rescue& rs) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const u32 done = atomicAdd(&rs.poison[1], 1u);
    if (done == gridDim.x - 1) {  // every workgroup has read poison[0] (at its start) and finished its columns
      __hip_atomic_store(&rs.poison[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&rs.poison[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&rs.gave_up[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
template <int MET>
__global__ __launch_bounds__(BLOCK) void k_cumsum_rescue(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo, int band, Rescue rs) {
  if (!rescue_begin(rs)) return;
  cumsum_strided_body<HV, MET, true, true, 8, true>(in, out, g, ntile, a, m_in, mi, m_out, mo, band);
  rescue_end(rs);
}

template <int MET, int R>
__global__ __launch_bounds__(BLOCK) void k_cumsum_chain(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo, ChainArgs ch) {
  constexpr int V = HV;
  typedef typename VecT<V>::type T;
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  u32 c, o32, tile;
  if (!chain_task(ch, ntile, c, o32, tile)) return;
  const int64_t o = o32;
  const int lane = threadIdx.x & 63;
  const u32 lx = tile * WAVE + lane;  // lane index along the row: the slot index
  const int64_t x = (int64_t)lx * V;
  if (x >= g.inner) return;
  const int64_t inner = g.inner, n = g.n_in;
  // row base = wave-uniform 64-bit pointer, lane part = 32-bit offset: the loads / stores take the scalar-base form and
  // the R addresses cost no vector registers (they were 64 of this kernel's 141: 3 waves per SIMD instead of 6)
  const u32 xo = lx * V;
  const real* pin = in + (o * n) * inner;
  real* pout = out + (o * g.n_out) * inner;

  int64_t mi_base = 0, mo_base = 0, mi_step = 0, mo_step = 0;
  if (HAS_MI) {
    mi_base = outer_off(g, mi, o) + inner_off(g, mi, x);
    mi_step = (V > 1) ? inner_off(g, mi, x + 1) - inner_off(g, mi, x) : 0;
  }
  if (HAS_MO) {
    mo_base = outer_off(g, mo, o) + inner_off(g, mo, x);
    mo_step = (V > 1) ? inner_off(g, mo, x + 1) - inner_off(g, mo, x) : 0;
  }
  const int64_t k0 = (int64_t)c * R;           // first row of the chunk in scan order
  const int rows = (n - k0 < R) ? (int)(n - k0) : R;  // wave-uniform; < R in the last chunk only
  auto row = [&](int r) -> int64_t { return a.reverse ? n - 1 - (k0 + r) : k0 + r; };
  T v[R];
  if (rows == R) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ldg<T, true>(pin + row(r) * inner + xo);
    if (HAS_MI) {
      if (V > 1 && met_vec_all<V>(m_in, mi_base, mi_step, mi.axis)) {  // the load form decided once, outside the row loop
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = v[r] * *reinterpret_cast<const T*>(m_in + mi_base + row(r) * mi.axis);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = v[r] * ldm<T>(m_in, mi_base + row(r) * mi.axis, mi_step);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      v[r] = splat<T>(real(0));
      if (r < rows) {
        v[r] = ldg<T, true>(pin + row(r) * inner + xo);
        if (HAS_MI) v[r] = v[r] * ldm<T>(m_in, mi_base + row(r) * mi.axis, mi_step);
      }
    }
  }
  if (a.skipna) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = nan0(v[r]);
  }
  // the running sum of everything before this chunk
  u32x4* ring = reinterpret_cast<u32x4*>(ch.slots);
  T acc = splat<T>(real(0));
  if (c > 0) {
    const u32x4* src = ring + ((size_t)o32 * 2 + ((c - 1) & 1)) * ch.srow + lx;
    acc = chain_wait<T>(src, c, ch);
  }
  // the chunk's cumulative values, in the march's order (the first row of the column is assigned, not added to 0)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < rows) {
      acc = (c > 0 || r > 0) ? acc + v[r] : v[r];
      v[r] = acc;
    }
  }
  if (c + 1 < ch.nchunk) {
    ring[((size_t)o32 * 2 + (c & 1)) * ch.srow + lx] = chain_pack<T>(acc, c + 1);
  } else if (ch.nchunk > 1) {  // end of the column: leave the ring as it was found
    const u32x4 zero = {0u, 0u, 0u, 0u};
    ring[((size_t)o32 * 2 + 0) * ch.srow + lx] = zero;
    ring[((size_t)o32 * 2 + 1) * ch.srow + lx] = zero;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the hand-off leaves before the R stores below
  // outputs: trim / pad table folded into the output index, halo cells written by the task that holds their value
  // (xgcm/grid.py:1385-1391; numpy.pad semantics)
  auto put = [&](int64_t jo, T val) {
    if (HAS_MO) val = val / ldm<T>(m_out, mo_base + jo * mo.axis, mo_step);
    stg<T, true>(pout + jo * inner + xo, val);
  };
  const int64_t first_kept = a.trim_lo, last_kept = n - 1 - a.trim_hi;
  const int64_t shift = a.pad_lo - a.trim_lo;
  const bool wrap = a.bc == XG_BC_PERIODIC, ext = a.bc == XG_BC_EXTEND;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < rows) {
      const int64_t idx = row(r);
      if (idx >= first_kept && idx <= last_kept) put(idx + shift, v[r]);
      if (idx == first_kept) {
        if (a.pad_lo && ext) put(0, v[r]);
        if (a.pad_hi && wrap) put(g.n_out - 1, v[r]);
      }
      if (idx == last_kept) {
        if (a.pad_lo && wrap) put(0, v[r]);
        if (a.pad_hi && ext) put(g.n_out - 1, v[r]);
      }
    }
  }
  if (c == 0 && a.bc == XG_BC_FILL) {
    if (a.pad_lo) put(0, splat<T>(a.fill));
    if (a.pad_hi) put(g.n_out - 1, splat<T>(a.fill));
  }
}

// Inclusive wave scan with DPP row shifts / row broadcasts (VALU data path) instead of __shfl_up (ds_bpermute, the LDS
// crossbar): 4 shifts inside each row of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows
// 2 / 3.  Lanes without a source add 0.  (A different association than the shuffle ladder: contiguous-axis sums are
// re-associated by contract, 1e-12.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ real dpp_take(real v) {
#ifdef XG_F32
  return __uint_as_float((u32)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROW_MASK, 0xf, false));
#else
  const u64 b = __builtin_bit_cast(u64, v);
  const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, CTRL, ROW_MASK, 0xf, false);
  const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  const u64 r = (u64)lo | ((u64)hi << 32);
  return __builtin_bit_cast(real, r);
#endif
}
__device__ __forceinline__ real wave_scan_dpp(real s) {
  s += dpp_take<0x111, 0xf>(s);  // row_shr:1
  s += dpp_take<0x112, 0xf>(s);  // row_shr:2
  s += dpp_take<0x114, 0xf>(s);  // row_shr:4
  s += dpp_take<0x118, 0xf>(s);  // row_shr:8
  s += dpp_take<0x142, 0xa>(s);  // row_bcast:15 -> rows 1, 3
  s += dpp_take<0x143, 0xc>(s);  // row_bcast:31 -> rows 2, 3
  return s;
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
  const int n = (int)g.n_in, no = (int)g.n_out;  // the host sends rows of 2^31 cells or more elsewhere
  const real* prow = in + row * n;
  real* orow = out + row * no;
  int64_t mi_base = 0, mo_base = 0;
  if (HAS_MI) mi_base = outer_off(g, mi, row);
  if (HAS_MO) mo_base = outer_off(g, mo, row);
  const int first_kept = a.trim_lo, last_kept = n - 1 - a.trim_hi;
  const int shift = a.pad_lo - a.trim_lo;
  auto put = [&](int j, real v) {
    if (HAS_MO) v = v / m_out[mo_base + (int64_t)j * mo.axis];
    orow[j] = v;
  };
  auto fetch = [&](int k) -> real {  // k-th input in scan order (32-bit in-row indices: the loop is VALU-bound)
    real v = real(0);
    if (k < n) {
      const int idx = a.reverse ? n - 1 - k : k;
      v = prow[idx];
      if (HAS_MI) v = v * m_in[mi_base + (int64_t)idx * mi.axis];
      if (a.skipna) v = nan0(v);
    }
    return v;
  };
  real carry = real(0);
  int buf = 0;
  real cur = fetch(tid);
  for (int base = 0; base < n; base += BLOCK, buf ^= 1) {
    const int k = base + tid;
    const int idx = a.reverse ? n - 1 - k : k;
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
        if (a.pad_hi && a.bc == XG_BC_PERIODIC) put(no - 1, c);
      }
      if (idx == last_kept) {
        if (a.pad_lo && a.bc == XG_BC_PERIODIC) put(0, c);
        if (a.pad_hi && a.bc == XG_BC_EXTEND) put(no - 1, c);
      }
    }
  }
  if (tid == 0 && a.bc == XG_BC_FILL) {
    if (a.pad_lo) put(0, a.fill);
    if (a.pad_hi) put(no - 1, a.fill);
  }
}

// ------------------------------------------------------------------------------------------
// K6v: cumsum along the CONTIGUOUS axis, any row length.  The output ARRAY is 16-B
// aligned, a row of it need not be (N + 1 outputs of center -> outer): the row starts `lead` cells before
// a 16-B boundary and ends `tail` cells after one.  Threads own the aligned groups of NV consecutive
// OUTPUTS in between (one 16-B store each) and fetch the NV inputs behind them with narrow consecutive
// loads (input = output index - shift, so it may be misaligned: served by L1, the trick that made K1g
// fast; one 16-B load when the groups coincide); a thread scans its group, waves scan the group totals
// with shuffles, wave totals go through LDS, the running carry stays in a register.  The <= NV-1 lead
// and tail cells are summed by every thread (they seed / follow the carry) and stored by thread 0.
// Inputs that map outside the output range (at most one, when the trim is on the side the scan starts
// from) seed the carry.  A `periodic` halo cell repeats the kept cell at the OTHER end of the row, known only
// when the scan gets there: whoever stores that source cell stores the halo too, and the group that holds the
// halo cell stores its other cells one by one (no cell is written twice).  Re-associated sum => 1e-12 parity like K6.
// ------------------------------------------------------------------------------------------
template <int MET, bool NTS, int BS>
__global__ __launch_bounds__(BS) void k_cumsum_contig_vec(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 nrows, ScanArgs a,
    const real* __restrict__ m_in, MIdx mi, const real* __restrict__ m_out, MIdx mo, ZBand zb, u32 nwork, int flags) {
  const bool dpp = (flags & 1) != 0;  // wave scan on DPP; bit 1: the shifted-by-one input path (scan_sh1)
  constexpr bool HAS_MO = (MET & 1) != 0, HAS_MI = (MET & 2) != 0;
  constexpr int NW = BS / WAVE;
  __shared__ real wtot[2][NW];
  const u32 pb = (nwork + 7) >> 3;
  u32 row = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);  // XCD banding over the work sequence
  if (row >= nwork) return;
  if (zb.on) {  // metrics broadcast along the slow outer dim (cumint along X with dx(Y, X)): all levels of a band of rows
                // before the next band, so the band's metric rows are served by the XCD's L2 instead of the fabric
    u32 z, y;
    if (!zband_map(zb, row, z, y)) return;
    row = z * zb.Y + y;
  }
  if (row >= nrows) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = (int)g.n_in, no = (int)g.n_out;  // the host selects this kernel for rows below 2^31 cells only
  const real* prow = in + (int64_t)row * n;
  real* orow = out + (int64_t)row * no;
  int64_t mi_base = 0, mo_base = 0;
  if (HAS_MI) mi_base = outer_off(g, mi, row);
  if (HAS_MO) mo_base = outer_off(g, mo, row);
  const int shift = a.pad_lo - a.trim_lo;
  const int lead = (int)((NV - (int)(((u64)row * (u64)no) % NV)) % NV);  // cells before the row's first 16-B boundary
  const int groups = (no - lead) / NV;
  const int gend = lead + groups * NV;                                   // first tail cell
  auto fetch = [&](int idx) -> real {  // weighted, NaN-cleaned input or 0 outside the row
    if (idx < 0 || idx >= n) return real(0);
    real v = prow[idx];
    if (HAS_MI) v = v * m_in[mi_base + (int64_t)idx * mi.axis];
    if (a.skipna) v = nan0(v);
    return v;
  };
  auto put1 = [&](int j, real v) {  // scalar store of output cell j
    if (HAS_MO) v = v / m_out[mo_base + (int64_t)j * mo.axis];
    orow[j] = v;
  };
  // halo cells of the padded cumulative result (fill / extend only): j = 0 and j = no - 1; an `extend` halo
  // repeats its inner neighbour (j = 1 resp. no - 2), which may live in another part of the row
  const bool lo_halo = a.pad_lo != 0, hi_halo = a.pad_hi != 0, fillmode = (a.bc == XG_BC_FILL);
  const bool wrap = (a.bc == XG_BC_PERIODIC) && (lo_halo || hi_halo);
  const int src_lo = wrap && lo_halo ? no - 1 - a.pad_hi : -1;  // output cell whose value the j = 0 halo repeats
  const int src_hi = wrap && hi_halo ? a.pad_lo : -1;           // ... and the j = no - 1 halo
  // sequential part in scan order over output cells [j0, j1] (either direction), all threads alike, thread 0
  // stores; returns the carry after it
  auto scalars = [&](real carry, int jfirst, int count) -> real {
    real prev = real(0);
    bool have_prev = false;
    for (int q = 0; q < count; ++q) {
      const int j = a.reverse ? jfirst - q : jfirst + q;
      carry += fetch(j - shift);
      real val = carry;
      const bool is_lo = lo_halo && j == 0, is_hi = hi_halo && j == no - 1;
      if (tid == 0) {
        if (is_lo || is_hi) {
          if (fillmode) put1(j, (real)a.fill);
          else if (wrap) {}  // stored with its source cell
          else if (have_prev && ((is_lo && a.reverse) || (is_hi && !a.reverse))) put1(j, prev);  // neighbour came just before
          // otherwise the neighbour comes next (stored below) or sits in a group (its owner stores the halo)
        } else {
          put1(j, val);
          if (wrap) {
            if (j == src_lo) put1(0, val);
            if (j == src_hi) put1(no - 1, val);
          } else {
            // the neighbour of a halo cell that was passed one step earlier in scan order
            if (!fillmode && lo_halo && j == 1 && !a.reverse && lead >= 2) put1(0, val);
            if (!fillmode && hi_halo && j == no - 2 && a.reverse && no - gend >= 2) put1(no - 1, val);
          }
        }
      }
      prev = val;
      have_prev = true;
    }
    return carry;
  };
  // the one input (if any) that precedes everything in scan order but maps outside [0, no)
  real carry = real(0);
  if (!a.reverse && shift < 0) carry = fetch(0);
  if (a.reverse && (n - 1 + shift) >= no) carry = fetch(n - 1);
  // cells before the groups in scan order
  if (!a.reverse) carry = scalars(carry, 0, lead);
  else carry = scalars(carry, no - 1, no - gend);
  int buf = 0;
  auto group_lo = [&](int t) -> int { return a.reverse ? lead + NV * (groups - 1 - t) : lead + NV * t; };
  // inputs of output group t.  All but the first / last group of a row lie inside the row: no per-element
  // bounds checks there (they were half of the loop's VALU work); one 16-B load when input and output groups
  // coincide (no shift, equal row lengths) and the arrays keep the alignment, else NV narrow loads served by L1.
  const bool vec_in = (shift == 0) && (n == no) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
  // the usual metric runs along the row with unit stride (dx(Y, X)): a row pointer + a 32-bit index instead of a
  // 64-bit multiply per element (this kernel has the highest VALU share of the library: 0.45)
  const bool mi_unit = HAS_MI && mi.axis == 1;
  const real* mi_row = m_in + mi_base;
  // ... and when its row keeps the output groups' alignment, ONE 16-B load per group instead of NV narrow ones
  const bool mi_al = mi_unit && ((reinterpret_cast<uintptr_t>(mi_row + lead) & 15u) == 0);
  // (Round 3 tried two more things here, both measured in one process and both REMOVED: for the shifted scan -- center ->
  // left, the Grid's default -- aligned vectors of field and metric with the one missing product taken from the
  // neighbouring lane by DPP: cumsum X 0.73 -> 0.62, cumint X 0.66 -> 0.56, the lane-0 fix-up and the extra moves cost more
  // than the misaligned narrow loads the L1 serves; and two vectors per thread and pass: 0.74 -> 0.70.
  // profiles/history/r03q_ab_scan_sh1.jsonl, r03q_ab_scan_gv_nosh1.jsonl)
  // shift == 1 (center -> left with its leading halo cell: the Grid's default): output group t takes the input cell BEFORE
  // its own aligned group and the group's first NV - 1 cells.  With 4-byte elements that is one aligned 16-B load + one
  // narrow load instead of four narrow ones (per array: the metric the same) -- float32 cumint X through the Grid ran at
  // 0.51 where the unshifted scan reaches 0.70.  (8-byte elements: two loads either way.  The round-3 variant took the
  // extra cell from the neighbouring LANE and paid for the fix-ups; this one reads it, an L1 hit.)
  // (with an input metric only: the plain scan's four narrow loads were no slower -- 0.93 against 1.00 ms with this path)
  const bool sh1_in = (NV == 4) && HAS_MI && (flags & 2) && (shift == 1) && (n == no) && ((reinterpret_cast<uintptr_t>(in) & 15u) == 0);
  const bool sh1_mi = sh1_in && mi_unit && ((reinterpret_cast<uintptr_t>(mi_row + lead) & 15u) == 0);
  auto load_group = [&](int t, real (&x)[NV]) {
    if (t >= groups) {
#pragma unroll
      for (int k = 0; k < NV; ++k) x[k] = real(0);
      return;
    }
    const int i0 = group_lo(t) - shift;
    if (sh1_in && i0 >= 0 && i0 + 1 + NV <= n) {
      const dv v = *reinterpret_cast<const dv*>(prow + i0 + 1);  // the output group's own (aligned) cells
      x[0] = prow[i0];
#pragma unroll
      for (int k = 1; k < NV; ++k) x[k] = v[k - 1];
      if (HAS_MI) {
        if (sh1_mi) {
          const dv mv = *reinterpret_cast<const dv*>(mi_row + i0 + 1);
          x[0] = x[0] * mi_row[i0];
#pragma unroll
          for (int k = 1; k < NV; ++k) x[k] = x[k] * mv[k - 1];
        } else {
#pragma unroll
          for (int k = 0; k < NV; ++k) x[k] = x[k] * (mi_unit ? mi_row[i0 + k] : m_in[mi_base + (int64_t)(i0 + k) * mi.axis]);
        }
      }
      if (a.skipna) {
#pragma unroll
        for (int k = 0; k < NV; ++k) x[k] = nan0(x[k]);
      }
      return;
    }
    if (i0 >= 0 && i0 + NV <= n) {
      if (vec_in) {
        const dv v = *reinterpret_cast<const dv*>(prow + i0);
#pragma unroll
        for (int k = 0; k < NV; ++k) x[k] = v[k];
      } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) x[k] = prow[i0 + k];
      }
      if (HAS_MI) {
        if (vec_in && mi_al) {
          const dv mv = *reinterpret_cast<const dv*>(mi_row + i0);
#pragma unroll
          for (int k = 0; k < NV; ++k) x[k] = x[k] * mv[k];
        } else {
#pragma unroll
          for (int k = 0; k < NV; ++k) x[k] = x[k] * (mi_unit ? mi_row[i0 + k] : m_in[mi_base + (int64_t)(i0 + k) * mi.axis]);
        }
      }
      if (a.skipna) {
#pragma unroll
        for (int k = 0; k < NV; ++k) x[k] = nan0(x[k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NV; ++k) x[k] = fetch(i0 + k);
    }
  };
  real xn[NV];  // the next pass's inputs are loaded before this pass's scan and barrier
  load_group(tid, xn);
  for (int base = 0; base < groups; base += BS, buf ^= 1) {
    const int t = base + tid;
    const bool act = t < groups;
    const int jlo = group_lo(t);
    real x[NV], l[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) x[k] = xn[k];
    load_group(t + BS, xn);
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
    if (dpp) s = wave_scan_dpp(s);
    else {
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        real u = __shfl_up(s, d, WAVE);
        if (lane >= d) s += u;
      }
    }
    real excl = dpp ? dpp_take<0x138, 0xf>(s) : __shfl_up(s, 1, WAVE);  // wave_shr:1
    if (lane == 0) excl = real(0);
    if (lane == WAVE - 1) wtot[buf][wv] = s;
    __syncthreads();
    real woff = real(0), tot = real(0);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
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
      if (wrap) {
        if (src_lo >= jlo && src_lo < jlo + NV) put1(0, res[src_lo - jlo]);
        if (src_hi >= jlo && src_hi < jlo + NV) put1(no - 1, res[src_hi - jlo]);
        const bool has_lo = lo_halo && jlo == 0, has_hi = hi_halo && jlo + NV == no;
        if (has_lo || has_hi) {  // the group around a halo cell: its other cells one by one
#pragma unroll
          for (int k = 0; k < NV; ++k)
            if (!(has_lo && k == 0) && !(has_hi && k == NV - 1)) put1(jlo + k, res[k]);
        } else {
          if (HAS_MO) res = res / ldm<dv>(m_out, mo_base + (int64_t)jlo * mo.axis, mo.axis);
          stg<dv, NTS>(orow + jlo, res);
        }
      } else {
        // halo cells inside this group sit next to a kept cell of the same group
        if (lo_halo && jlo == 0) res[0] = fillmode ? (real)a.fill : res[1];
        if (hi_halo && jlo + NV == no) res[NV - 1] = fillmode ? (real)a.fill : res[NV - 2];
        // an `extend` halo cell just outside this group (lead == 1 / one tail cell) repeats this group's edge cell
        if (!fillmode && lo_halo && jlo == 1) put1(0, res[0]);
        if (!fillmode && hi_halo && jlo + NV == no - 1) put1(no - 1, res[NV - 1]);
        if (HAS_MO) res = res / ldm<dv>(m_out, mo_base + (int64_t)jlo * mo.axis, mo.axis);
        stg<dv, NTS>(orow + jlo, res);
      }
    }
  }
  // cells after the groups in scan order
  if (!a.reverse) scalars(carry, gend, no - gend);
  else scalars(carry, lead - 1, lead);
}

// ------------------------------------------------------------------------------------------
// K4: weighted sum along a STRIDED axis: one lane per output column pair, sequential in k
// (bit-exact with numpy's reduction over a non-last axis).
// ------------------------------------------------------------------------------------------
// WMODE: 0 no weights, 1 a weight per cell of the row, 2 ONE weight per row (drF(Z) under (Z, Y, X): wave-uniform, read
// through the scalar cache; nothing rides in the window)
template <int V, int WMODE, bool NTL, int U, bool PIPE>
__device__ __forceinline__ void reduce_strided_body(
    const real* __restrict__ in, real* __restrict__ out, const Geo& g, u32 ntile, int skipna,
    const real* __restrict__ wgt, const MIdx& mw, int band) {
  typedef typename VecT<V>::type T;
  constexpr bool HAS_W = WMODE != 0, WU = WMODE == 2;
  // U independent loads in flight per lane
  const u64 w = band ? banded_wave_id() : wave_id();
  u32 tile;
  int64_t o;
  if (band & 2) {  // weights shared by all outer indices: outer indices fastest, an XCD band = all levels of a few x-tiles,
                   // marching down the same weight columns together (the weight rows are then L2 hits; see K4L)
    tile = (u32)(w / (u64)g.outer);
    o = (int64_t)(w - (u64)tile * (u64)g.outer);
    if (tile >= ntile) return;
  } else {
    tile = (u32)(w % ntile);
    o = (int64_t)(w / ntile);
    if (o >= g.outer) return;
  }
  if (WU) o = (int64_t)__builtin_amdgcn_readfirstlane((u32)o);  // (host: outer < 2^32) the division ran on the vector unit
  const int lane = threadIdx.x & 63;
  const int64_t x = ((int64_t)tile * WAVE + lane) * V;
  if (x >= g.inner) return;
  const int64_t inner = g.inner, n = g.n_in;
  const real* pin = in + (o * n) * inner + x;
  int64_t mb = 0, ms = 0;
  if (WU) mb = outer_off(g, mw, o);
  else if (HAS_W) {
    mb = outer_off(g, mw, o) + inner_off(g, mw, x);
    ms = (V > 1) ? inner_off(g, mw, x + 1) - inner_off(g, mw, x) : 0;
  }
  T acc = splat<T>(real(0)), den = splat<T>(real(0));
  bool started = false;
  // modes 4 / 5: weighted mean in ONE pass, numerator and denominator march together; 6 / 7: the same two sums
  // written side by side (out[0 : N] numerators, out[N : 2N] denominators) for means over several dims
  const bool pair = skipna >= 6;
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;
  // the weight of a row is loaded WITH the row (same window / batch), not inside `step` (see k_cumsum_strided)
  auto step = [&](int64_t k, T v, T wv) {
    if (mean) {  // the two sums of modes 1 / 0 (numerator) and 2 / 3 (denominator), same order, same bits
      T d = as_count(v, skipna == 4 ? 2 : 3);
      if (HAS_W) { d = d * wv; v = v * wv; }
      if (skipna == 4) v = nan0(v);
      d = nan0(d);
      den = started ? den + d : d;
    } else {
      if (skipna >= 2) v = as_count(v, skipna);
      if (HAS_W) v = v * wv;
      if (skipna) v = nan0(v);
    }
    acc = started ? acc + v : v;
    started = true;
  };
  // (the form of the per-cell weight load -- one aligned vector or element by element -- is decided once per wave: see met_vec_all)
  auto march = [&](auto wvec) {
  constexpr bool WV = decltype(wvec)::value;
  auto ldw = [&](int64_t k) -> T {
    if (WU) return splat<T>(wgt[mb + k * mw.axis]);
    if constexpr (WV) return *reinterpret_cast<const T*>(wgt + mb + k * mw.axis);
    else return HAS_W ? ldm<T>(wgt, mb + k * mw.axis, ms) : splat<T>(real(1));
  };
  int64_t k = 0;
  if (PIPE) {  // rolling window of U loads (see k_cumsum_strided)
    constexpr int UW = (HAS_W && !WU) ? U : 1;
    T v[U], wv[UW];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < n) {
        v[u] = ldg<T, NTL>(pin + (int64_t)u * inner);
        if (HAS_W && !WU) wv[u % UW] = ldw(u);
      }
    for (; k + 2 * U <= n; k += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T x = v[u], xw = WU ? ldw(k + u) : HAS_W ? wv[u % UW] : splat<T>(real(1));
        v[u] = ldg<T, NTL>(pin + (k + U + u) * inner);
        if (HAS_W && !WU) wv[u % UW] = ldw(k + U + u);
        step(k + u, x, xw);
      }
    }
    for (; k < n; k += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T x = v[u], xw = (WU && k + u < n) ? ldw(k + u) : (HAS_W && !WU) ? wv[u % UW] : splat<T>(real(1));
        if (k + U + u < n) {
          v[u] = ldg<T, NTL>(pin + (k + U + u) * inner);
          if (HAS_W && !WU) wv[u % UW] = ldw(k + U + u);
        }
        if (k + u < n) step(k + u, x, xw);
      }
    }
  } else {
  for (; k + U <= n; k += U) {
    constexpr int UW = (HAS_W && !WU) ? U : 1;
    T v[U], wv[UW];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = ldg<T, NTL>(pin + (k + u) * inner);
      if (HAS_W && !WU) wv[u % UW] = ldw(k + u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(k + u, v[u], WU ? ldw(k + u) : HAS_W ? wv[u % UW] : splat<T>(real(1)));
  }
  for (; k < n; ++k) step(k, ldg<T, NTL>(pin + k * inner), ldw(k));
  }
  };
  if (V > 1 && HAS_W && !WU && met_vec_all<V>(wgt, mb, ms, mw.axis)) march(std::true_type{});
  else march(std::false_type{});
  if (pair) {
    *reinterpret_cast<T*>(out + o * inner + x) = acc;
    *reinterpret_cast<T*>(out + (g.outer + o) * inner + x) = den;
  } else {
    *reinterpret_cast<T*>(out + o * inner + x) = mean ? acc / den : acc;
  }
}
template <int V, int WMODE, bool NTL, int U, bool PIPE = false>
__global__ __launch_bounds__(BLOCK) void k_reduce_strided(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, int skipna,
    const real* __restrict__ wgt, MIdx mw, int band) {
  reduce_strided_body<V, WMODE, NTL, U, PIPE>(in, out, g, ntile, skipna, wgt, mw, band);
}
// the rescue twin of the chained weighted reductions (see k_cumsum_rescue)
__global__ __launch_bounds__(BLOCK) void k_reduce_rescue(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, int skipna,
    const real* __restrict__ wgt, MIdx mw, int band, Rescue rs) {
  if (!rescue_begin(rs)) return;
  reduce_strided_body<HV, 1, true, 8, true>(in, out, g, ntile, skipna, wgt, mw, band);
  rescue_end(rs);
}

// K4L: the long WEIGHTED march with the weights in LDS.  `integrate` / `average` along Y of (Z, Y, X) with dy(Y, X): the
// weights are shared by the levels, and every way of sharing them through registers costs what it saves -- riding in the
// march's window they halve the bytes in flight (K4 with weights: 0.52 of 8 TB/s), held by a chunk task for two levels they
// need a cross-wave chain whose 16-byte hand-offs end up in HBM (K4cz: 0.58-0.63, traffic 1.22x).  Here the NWV waves of a
// workgroup march the SAME x-tile of NWV consecutive levels side by side, block of U rows after block of U rows: each wave
// keeps a rolling window of U field rows in registers like the unweighted march (sum along Y without weights: 0.80), and the
// block's U weight rows are fetched ONCE per workgroup -- every wave a share -- into a double-buffered LDS tile that all of
// them read; one barrier per block.  Sequential additions per column in row order: the bits of numpy / of K4.
// integrate Y 0.58 -> 0.69, average Y (two running sums, which doubled K4cz's hand-offs) 0.51 -> 0.68 of 8 TB/s in one process;
// HBM reads 1.22x (the weights once per workgroup, Infinity Cache hits), writes = the result.  No chain, no rescue twin.
template <int U, int NWV>
__global__ __launch_bounds__(NWV * WAVE) void k_reduce_ldsw(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, u32 nblk, int skipna,
    const real* __restrict__ wgt, MIdx mw, u32 ngrp) {
  constexpr int V = HV;
  typedef typename VecT<V>::type T;
  static_assert(U % NWV == 0, "every wave fetches U / NWV weight rows of a block");
  constexpr int SH = U / NWV;
  __shared__ T s_w[2][U][WAVE];
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;  // (whole workgroups leave: nobody is left waiting at a barrier)
  // ngrp != 0: level groups run fastest, so an XCD band = ALL level groups of a few x-tiles: their workgroups march down the
  // same weight columns together and each XCD fetches only its own eighth of the weight plane (tile-fastest order: every XCD
  // streams the whole plane once per level group it holds)
  u32 og, tile;
  if (ngrp) { tile = lb / ngrp; og = lb - tile * ngrp; }
  else { og = lb / ntile; tile = lb - og * ntile; }
  const u32 wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int64_t o = (int64_t)og * NWV + wv;
  const int64_t x = ((int64_t)tile * WAVE + lane) * V;
  const int64_t inner = g.inner, n = g.n_in;
  const bool live = o < g.outer && x < inner;  // a wave beyond the last level / a lane beyond the row still fetches weights
  const bool lane_ok = x < inner;
  const real* pin = in + (live ? o : 0) * n * inner + (lane_ok ? x : 0);
  const int64_t mb = lane_ok ? inner_off(g, mw, x) : 0;
  const int64_t ms = (V > 1 && lane_ok) ? inner_off(g, mw, x + 1) - mb : 0;
  const bool pair = skipna >= 6;
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;
  T acc = splat<T>(real(0)), den = splat<T>(real(0));
  bool started = false;
  auto step = [&](T v, T wv_) {  // k_reduce_strided's `step`, same operations in the same order
    if (mean) {
      T d = as_count(v, skipna == 4 ? 2 : 3);
      d = d * wv_;
      v = v * wv_;
      if (skipna == 4) v = nan0(v);
      d = nan0(d);
      den = started ? den + d : d;
    } else {
      if (skipna >= 2) v = as_count(v, skipna);
      v = v * wv_;
      if (skipna) v = nan0(v);
    }
    acc = started ? acc + v : v;
    started = true;
  };
  // this wave's share of a block's weight rows: rows [b U + wv SH, + SH) -> registers -> LDS buffer b & 1
  T wreg[SH];
  auto fetch_w = [&](int64_t b) {
#pragma unroll
    for (int r = 0; r < SH; ++r) {
      int64_t k = b * U + (int64_t)wv * SH + r;
      if (k >= n) k = n - 1;  // (rows beyond the column: never consumed)
      wreg[r] = lane_ok ? ldm<T>(wgt, mb + k * mw.axis, ms) : splat<T>(real(0));
    }
  };
  auto park_w = [&](int64_t b) {
#pragma unroll
    for (int r = 0; r < SH; ++r) s_w[b & 1][wv * SH + r][lane] = wreg[r];
  };
  const int64_t nb = (n + U - 1) / U;
  fetch_w(0);
  park_w(0);
  T v[U];  // the rolling window of field rows: row k + U is requested when row k is consumed
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (live && u < n) v[u] = ldg<T, true>(pin + (int64_t)u * inner);
  __syncthreads();
  for (int64_t b = 0; b < nb; ++b) {
    if (b + 1 < nb) fetch_w(b + 1);  // in flight while this block is consumed
    const int64_t k0 = b * U;
    if (live) {
      if (k0 + 2 * U <= n) {  // steady state: no bounds tests
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const T xv = v[u];
          v[u] = ldg<T, true>(pin + (k0 + U + u) * inner);
          step(xv, s_w[b & 1][u][lane]);
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const T xv = v[u];
          if (k0 + U + u < n) v[u] = ldg<T, true>(pin + (k0 + U + u) * inner);
          if (k0 + u < n) step(xv, s_w[b & 1][u][lane]);
        }
      }
    }
    if (b + 1 < nb) park_w(b + 1);  // buffer (b + 1) & 1 was last read in block b - 1, before the barrier below of that block
    __syncthreads();
  }
  if (!live) return;
  if (pair) {
    *reinterpret_cast<T*>(out + o * inner + x) = acc;
    *reinterpret_cast<T*>(out + (g.outer + o) * inner + x) = den;
  } else {
    *reinterpret_cast<T*>(out + o * inner + x) = mean ? acc / den : acc;
  }
}

// K4Z (round 6): the long WEIGHTED march with the weights shared IN REGISTERS.  One wave marches the same x-tile of ZL
// consecutive levels at once: a row step is ZL field rows + ONE weight row, every load of U steps rides in one rolling
// window (U (ZL + 1) independent loads per lane; nothing waits on a barrier, nothing goes through LDS), and the weight row
// costs 1 / ZL of a load per field row instead of K4's whole one.  Few waves (75 levels x 57 tiles / 5 = 855: one per SIMD)
// with 36 - 48 loads in flight each instead of K4L's four-wave workgroups in lock step.  Sequential additions per column in
// row order: the bits of numpy / of K4.  Workgroup = ONE wave, so that the few waves spread over all the chip's SIMDs.
// MODE (compile time, so that the row loop carries no mode tests): 0 = sum(x * w), 1 = the same skipping NaN, 2 / 3 = the
// weights of the valid / of all cells (`skipna` 2 / 3), 4 / 5 = the one-pass mean over the valid / over all cells (`skipna`
// 4 / 5; 6 / 7 = the same with numerator and denominator stored side by side, a runtime flag read after the loop only).
// WV (decided once per wave, then a compile-time constant of the row loop): the lane's V weights are one aligned vector
// load in every row -- float32's two-element lanes otherwise test alignment and step at EVERY weight load, a divergent
// branch per row step that cut the loop into fragments (float32 average Y 0.37 of 8 TB/s, K4L 0.54; profiles/EXPERIMENTS.md).
// Accumulators start at -0.0: (-0.0) + v == v for every v (the first row is ASSIGNED, as numpy's reduction does: a column
// of -0.0 sums to -0.0), so no "first row" test rides in the loop.
template <int U, int ZL, int MODE>
__global__ __launch_bounds__(WAVE) void k_reduce_zmarch(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, u32 ngrp, u32 ntask, int skipna,
    const real* __restrict__ wgt, MIdx mw) {
  constexpr int V = HV;
  constexpr bool TWO = MODE >= 4;  // numerator and denominator march together
  typedef typename VecT<V>::type T;
  // XCD banding over (tile, level group) with the level groups fastest: the groups of one x-tile run side by side on ONE
  // XCD and find each other's weight rows in its L2
  const u32 pb = (ntask + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= ntask) return;
  const u32 tile = lb / ngrp, og = lb - tile * ngrp;
  const int lane = threadIdx.x & 63;
  const int64_t x = ((int64_t)tile * WAVE + lane) * V;
  const int64_t inner = g.inner, n = g.n_in;
  if (x >= inner) return;
  const int64_t o0 = (int64_t)og * ZL;
  const int nlev = (g.outer - o0 < ZL) ? (int)(g.outer - o0) : ZL;  // wave-uniform; < ZL in the last group only
  // row base = wave-uniform 64-bit pointer (scalar registers), lane part = one 32-bit offset: the loads take the scalar-base
  // form, the U (ZL + 1) addresses of the window cost no vector registers and no 64-bit vector arithmetic (as in K5c)
  const u32 xo = (u32)x;
  const real* pin[ZL];
#pragma unroll
  for (int l = 0; l < ZL; ++l) pin[l] = in + (o0 + (l < nlev ? l : 0)) * n * inner;  // (a level beyond the last re-reads level o0: never stored)
  const int64_t mb = inner_off(g, mw, x);
  const int64_t ms = (V > 1) ? inner_off(g, mw, x + 1) - mb : 0;
  T acc[ZL], den[TWO ? ZL : 1];
#pragma unroll
  for (int l = 0; l < ZL; ++l) acc[l] = splat<T>(real(-0.0));
#pragma unroll
  for (int l = 0; l < (TWO ? ZL : 1); ++l) den[l] = splat<T>(real(-0.0));
  auto step = [&](int l, T v, T wv_) {  // k_reduce_strided's `step`, same operations in the same order
    if (TWO) {
      T d = as_count(v, MODE == 4 ? 2 : 3);
      d = d * wv_;
      v = v * wv_;
      if (MODE == 4) v = nan0(v);
      d = nan0(d);
      den[TWO ? l : 0] = den[TWO ? l : 0] + d;
    } else if (MODE >= 2) {
      v = as_count(v, MODE);
      v = v * wv_;
      v = nan0(v);
    } else {
      v = v * wv_;
      if (MODE == 1) v = nan0(v);
    }
    acc[l] = acc[l] + v;
  };
  auto march = [&](auto wvec) {
    constexpr bool WV = decltype(wvec)::value;
    auto ldw = [&](int64_t k) -> T {
      if constexpr (WV) return *reinterpret_cast<const T*>(wgt + mb + k * mw.axis);
      else return ldm<T>(wgt, mb + k * mw.axis, ms);
    };
    T v[U][ZL], wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < n) {
        wv[u] = ldw(u);
#pragma unroll
        for (int l = 0; l < ZL; ++l) v[u][l] = ldg<T, true>(pin[l] + (int64_t)u * inner + xo);
      }
    int64_t k0 = 0;
    for (; k0 + 2 * U <= n; k0 += U) {  // steady state: no bounds tests; row k0 + u is consumed, row k0 + U + u requested
#pragma unroll
      for (int u = 0; u < U; ++u) {  // consume slot u, then refill it in place (no copies of the window's registers)
#pragma unroll
        for (int l = 0; l < ZL; ++l) step(l, v[u][l], wv[u]);
        wv[u] = ldw(k0 + U + u);
#pragma unroll
        for (int l = 0; l < ZL; ++l) v[u][l] = ldg<T, true>(pin[l] + (k0 + U + u) * inner + xo);
      }
    }
    for (; k0 < n; k0 += U) {  // the last one or two windows (wave-uniform tests)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (k0 + u < n) {
#pragma unroll
          for (int l = 0; l < ZL; ++l) step(l, v[u][l], wv[u]);
        }
        if (k0 + U + u < n) {
          wv[u] = ldw(k0 + U + u);
#pragma unroll
          for (int l = 0; l < ZL; ++l) v[u][l] = ldg<T, true>(pin[l] + (k0 + U + u) * inner + xo);
        }
      }
    }
  };
  if (V > 1) {
    // every active lane's V weights sit side by side, vector-aligned in row 0, and the row step keeps that alignment
    const bool lane_ok = ms == 1 && (mw.axis % V) == 0 && ((reinterpret_cast<uintptr_t>(wgt) / sizeof(real) + (uintptr_t)mb) % V) == 0;
    if (__all(lane_ok)) march(std::true_type{});
    else march(std::false_type{});
  } else {
    march(std::false_type{});
  }
  const bool pair = TWO && skipna >= 6;
#pragma unroll
  for (int l = 0; l < ZL; ++l) {
    if (l < nlev) {
      const int64_t o = o0 + l;
      if (pair) {
        *reinterpret_cast<T*>(out + o * inner + x) = acc[l];
        *reinterpret_cast<T*>(out + (g.outer + o) * inner + x) = den[TWO ? l : 0];
      } else {
        *reinterpret_cast<T*>(out + o * inner + x) = TWO ? acc[l] / den[TWO ? l : 0] : acc[l];
      }
    }
  }
}

// K4c: the long WEIGHTED march (integrate / average along Y of (Z, Y, X) with dy(Y, X)) as a chained flat launch, K5c
// without the stores: a marching wave keeps the weight of every in-flight row in registers next to the row, so its
// window is 8 rows (52 % of 8 TB/s); a chunk task holds 32 rows + 32 weight rows once and ends.  Same order of
// additions as the march (bit-identical); two running sums travel for the mean / pair modes.  The last chunk of a
// column writes the result and zeroes the column's slots.
template <bool HAS_W, int R>
__global__ __launch_bounds__(BLOCK) void k_reduce_chain(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, int skipna,
    const real* __restrict__ wgt, MIdx mw, ChainArgs ch) {
  constexpr int V = HV;
  typedef typename VecT<V>::type T;
  u32 c, o32, tile;
  if (!chain_task(ch, ntile, c, o32, tile)) return;
  const int64_t o = o32;
  const int lane = threadIdx.x & 63;
  const u32 lx = tile * WAVE + lane;
  const int64_t x = (int64_t)lx * V;
  if (x >= g.inner) return;
  const int64_t inner = g.inner, n = g.n_in;
  const u32 xo = lx * V;
  const real* pin = in + (o * n) * inner;
  int64_t mb = 0, ms = 0;
  if (HAS_W) {
    mb = outer_off(g, mw, o) + inner_off(g, mw, x);
    ms = (V > 1) ? inner_off(g, mw, x + 1) - inner_off(g, mw, x) : 0;
  }
  const bool pair = skipna >= 6;
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;
  const u32 np = mean ? 2u : 1u;  // running sums per lane
  const int64_t k0 = (int64_t)c * R;
  const int rows = (n - k0 < R) ? (int)(n - k0) : R;
  T v[R], d[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    v[r] = splat<T>(real(0));
    d[r] = splat<T>(real(0));
  }
  if (rows == R) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ldg<T, true>(pin + (k0 + r) * inner + xo);
    if (HAS_W) {
#pragma unroll
      for (int r = 0; r < R; ++r) d[r] = ldm<T>(wgt, mb + (k0 + r) * mw.axis, ms);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < rows) {
        v[r] = ldg<T, true>(pin + (k0 + r) * inner + xo);
        if (HAS_W) d[r] = ldm<T>(wgt, mb + (k0 + r) * mw.axis, ms);
      }
  }
  // the terms of every row (k_reduce_strided's `step`, same operations in the same order)
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const T wv = d[r];
    T x1 = v[r];
    if (mean) {
      T dd = as_count(x1, skipna == 4 ? 2 : 3);
      if (HAS_W) { dd = dd * wv; x1 = x1 * wv; }
      if (skipna == 4) x1 = nan0(x1);
      d[r] = nan0(dd);
    } else {
      if (skipna >= 2) x1 = as_count(x1, skipna);
      if (HAS_W) x1 = x1 * wv;
      if (skipna) x1 = nan0(x1);
    }
    v[r] = x1;
  }
  u32x4* ring = reinterpret_cast<u32x4*>(ch.slots);
  T acc = splat<T>(real(0)), den = splat<T>(real(0));
  if (c > 0) {
    const u32x4* src = ring + (((size_t)o32 * 2 + ((c - 1) & 1)) * ch.srow + lx) * np;
    acc = chain_wait<T>(src, c, ch);
    if (mean) den = chain_wait<T>(src + 1, c, ch);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r < rows) {
      const bool first = (c == 0 && r == 0);
      acc = first ? v[r] : acc + v[r];
      if (mean) den = first ? d[r] : den + d[r];
    }
  }
  if (c + 1 < ch.nchunk) {
    u32x4* dst = ring + (((size_t)o32 * 2 + (c & 1)) * ch.srow + lx) * np;
    dst[0] = chain_pack<T>(acc, c + 1);
    if (mean) dst[1] = chain_pack<T>(den, c + 1);
    return;
  }
  if (ch.nchunk > 1) {
    const u32x4 zero = {0u, 0u, 0u, 0u};
    for (u32 par = 0; par < 2; ++par)
      for (u32 p = 0; p < np; ++p) ring[(((size_t)o32 * 2 + par) * ch.srow + lx) * np + p] = zero;
  }
  if (pair) {
    *reinterpret_cast<T*>(out + o * inner + xo) = acc;
    *reinterpret_cast<T*>(out + (g.outer + o) * inner + xo) = den;
  } else {
    *reinterpret_cast<T*>(out + o * inner + xo) = mean ? acc / den : acc;
  }
}

// K4cz: K4c with the weights of a chunk loaded ONCE for ZL consecutive outer indices ("levels").  PMC on K4c
// (FETCH_SIZE / WRITE_SIZE passes, round 2): its HBM traffic is the minimum (the field once, the weights once per XCD), yet it stops at
// 57 % -- what it saturates is the L2 -> CU path, which carries the field AND, for every level again, the weights
// (9.8 TB/s; the unweighted sum moves 6.4).  A task = (x-tile, chunk of R rows, ZL levels): R weight rows + ZL x R field
// rows in registers, then level after level: wait for that level's running sum, add the R rows in order, publish.
// Same additions in the same order as the march: same bits.
template <int R, int ZL>
__global__ __launch_bounds__(BLOCK) void k_reduce_chain_z(
    const real* __restrict__ in, real* __restrict__ out, Geo g, u32 ntile, int skipna,
    const real* __restrict__ wgt, MIdx mw, ChainArgs ch) {
  constexpr int V = HV;
  typedef typename VecT<V>::type T;
  u32 c, og, tile;
  if (!chain_task(ch, ntile, c, og, tile)) return;
  const int lane = threadIdx.x & 63;
  const u32 lx = tile * WAVE + lane;
  const int64_t x = (int64_t)lx * V;
  if (x >= g.inner) return;
  const int64_t inner = g.inner, n = g.n_in;
  const u32 xo = lx * V;
  const u32 o_first = og * ZL;
  const int nlev = (g.outer - (int64_t)o_first < ZL) ? (int)(g.outer - o_first) : ZL;  // wave-uniform
  // (host: the weights do not depend on the outer index)
  const int64_t mb = inner_off(g, mw, x);
  const int64_t ms = (V > 1) ? inner_off(g, mw, x + 1) - inner_off(g, mw, x) : 0;
  const bool pair = skipna >= 6;
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;
  const u32 np = mean ? 2u : 1u;
  const int64_t k0 = (int64_t)c * R;
  const int rows = (n - k0 < R) ? (int)(n - k0) : R;
  T wv[R], v[ZL][R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t k = k0 + ((r < rows) ? r : rows - 1);  // a short last chunk repeats its last row (not added)
    wv[r] = ldm<T>(wgt, mb + k * mw.axis, ms);
  }
#pragma unroll
  for (int kz = 0; kz < ZL; ++kz) {
    const int64_t o = o_first + ((kz < nlev) ? kz : nlev - 1);
    const real* pin = in + (o * n) * inner;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t k = k0 + ((r < rows) ? r : rows - 1);
      v[kz][r] = ldg<T, true>(pin + k * inner + xo);
    }
  }
  u32x4* ring = reinterpret_cast<u32x4*>(ch.slots);
#pragma unroll
  for (int kz = 0; kz < ZL; ++kz) {
    if (kz >= nlev) break;
    const u32 o32 = o_first + kz;
    T acc = splat<T>(real(0)), den = splat<T>(real(0));
    if (c > 0) {
      const u32x4* src = ring + (((size_t)o32 * 2 + ((c - 1) & 1)) * ch.srow + lx) * np;
      acc = chain_wait<T>(src, c, ch);
      if (mean) den = chain_wait<T>(src + 1, c, ch);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < rows) {
        T x1 = v[kz][r];
        const T w1 = wv[r];
        const bool first = (c == 0 && r == 0);
        if (mean) {  // k_reduce_strided's `step`, same operations in the same order
          T dd = as_count(x1, skipna == 4 ? 2 : 3);
          dd = dd * w1;
          x1 = x1 * w1;
          if (skipna == 4) x1 = nan0(x1);
          dd = nan0(dd);
          den = first ? dd : den + dd;
        } else {
          if (skipna >= 2) x1 = as_count(x1, skipna);
          x1 = x1 * w1;
          if (skipna) x1 = nan0(x1);
        }
        acc = first ? x1 : acc + x1;
      }
    }
    if (c + 1 < ch.nchunk) {
      u32x4* dst = ring + (((size_t)o32 * 2 + (c & 1)) * ch.srow + lx) * np;
      dst[0] = chain_pack<T>(acc, c + 1);
      if (mean) dst[1] = chain_pack<T>(den, c + 1);
    } else {
      if (ch.nchunk > 1) {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        for (u32 par = 0; par < 2; ++par)
          for (u32 p = 0; p < np; ++p) ring[(((size_t)o32 * 2 + par) * ch.srow + lx) * np + p] = zero;
      }
      if (pair) {
        *reinterpret_cast<T*>(out + (int64_t)o32 * inner + xo) = acc;
        *reinterpret_cast<T*>(out + (g.outer + (int64_t)o32) * inner + xo) = den;
      } else {
        *reinterpret_cast<T*>(out + (int64_t)o32 * inner + xo) = mean ? acc / den : acc;
      }
    }
  }
}

// K4b: weighted sum along the CONTIGUOUS axis: one wave per row, lane-strided partial sums then
// a shuffle tree (tolerance parity; numpy itself is pairwise here).
// SK >= 0: the mode is a compile-time constant (the plain sums, skipna False / True: `integrate`, `sum`) -- decided per element
// through select chains it cost 17 instructions per cell where one addition is needed (DESIGN rule 15); SK < 0: any mode.
template <bool HAS_W, bool VEC, int SK, bool WFAST = false>
__global__ __launch_bounds__(BLOCK) void k_reduce_contig(const real* __restrict__ in,
                                                         real* __restrict__ out, Geo g, int skipna_rt,
                                                         const real* __restrict__ wgt, MIdx mw, int ntl, ZBand zb) {
  int skipna = (SK >= 0) ? SK : skipna_rt;
  u64 row = wave_id();
  if (zb.on) {  // weights broadcast along the slow outer dim: all levels of a band of rows before the next band,
                // so that the band's weight rows are served by the XCD's L2 (integrate along X with dx(Y,X): 48 %
                // of 8 TB/s in level-major order, the weights re-read from the fabric once per level)
    u32 z, y;
    if (row >= (u64)zb.per_band.d * ((zb.Y + zb.B - 1) / zb.B)) return;
    if (!zband_map(zb, (u32)row, z, y)) return;
    row = (u64)z * zb.Y + y;
  }
  if ((int64_t)row >= g.outer) return;
  const int lane = threadIdx.x & 63;
  const int64_t n = g.n_in;
  const real* prow = in + row * n;
  int64_t mb = 0;
  if (HAS_W) mb = outer_off(g, mw, row);
  // WFAST (host): the weights run along the rows with unit stride and every weight row starts on a 16-B boundary like
  // the field's -- the weight vector of a lane is then ONE load at the field vector's own index from a wave-uniform row
  // pointer.  The general form (`ldm`: a 64-bit multiply by the runtime stride and an alignment test per load) costs 5x
  // the scalar and 2.2x the vector instructions of the unweighted kernel: float32 rows 0.76 -> 0.60 of 8 TB/s.  A
  // compile-time switch: decided per load at run time it broke up the batch of independent loads (float32 0.60 -> 0.47).
  constexpr bool wfast = HAS_W && WFAST;
  const real* wrow = HAS_W ? wgt + mb : nullptr;
  const bool pair = skipna >= 6;  // 6 / 7: numerator and denominator sums side by side (see k_reduce_strided)
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;  // weighted mean in one pass (numerator and denominator together)
  const int nmode = mean ? (skipna == 4 ? 1 : 0) : skipna, dmode = skipna == 4 ? 2 : 3;
  real acc = real(0), den = real(0);
  // one input value and its weight -> numerator term (and, for the mean modes, the denominator term)
  auto terms = [&](real v, real wv, real& num, real& dn) {
    if (mean) {
      dn = as_count(v, dmode);
      if (HAS_W) dn = dn * wv;
      dn = nan0(dn);
    }
    if (nmode >= 2) v = as_count(v, nmode);
    if (HAS_W) v = v * wv;
    if (nmode) v = nan0(v);
    num = v;
  };
  int64_t k0 = 0, lead = 0;
  if (VEC) {  // array 16-B aligned (host): 16-B loads between the row's first and last 16-B boundary, NV partial
              // sums per lane, two loads in flight; the <= NV-1 cells before / after go through the scalar loops
    lead = (NV - (int64_t)((row * (u64)n) % NV)) % NV;
    dv a = splat<dv>(real(0)), ad = splat<dv>(real(0));
    const int64_t nvec = (n - lead) / NV;
    auto ldv = [&](int64_t k) -> dv {
      return (ntl & 1) ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(prow + k)) : *reinterpret_cast<const dv*>(prow + k);
    };
    auto add = [&](const dv& v, const dv& wv) {
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        real num, dn = real(0);
        terms(v[c], wv[c], num, dn);
        a[c] += num;
        ad[c] += dn;
      }
    };
    int64_t t = lane;
    // RU independent loads (and weight loads) before the first addition; the additions keep their order
    auto batches = [&](auto ru) {
      constexpr int RU = decltype(ru)::value;
      for (; t + (RU - 1) * WAVE < nvec; t += RU * WAVE) {
        dv v[RU], wv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int64_t k = lead + (t + u * WAVE) * NV;
          v[u] = ldv(k);
          if constexpr (!HAS_W) wv[u] = splat<dv>(real(1));
          else if constexpr (wfast) wv[u] = *reinterpret_cast<const dv*>(wrow + k);
          else wv[u] = ldm<dv>(wgt, mb + k * mw.axis, mw.axis);
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) add(v[u], wv[u]);
      }
    };
    if (ntl & 4) batches(std::integral_constant<int, 8>{});
    if (ntl & 6) batches(std::integral_constant<int, 4>{});
    for (; t < nvec; t += WAVE) {
      const int64_t k = lead + t * NV;
      const dv v = ldv(k);
      dv wv = splat<dv>(real(1));
      if constexpr (HAS_W) {
        if constexpr (wfast) wv = *reinterpret_cast<const dv*>(wrow + k);
        else wv = ldm<dv>(wgt, mb + k * mw.axis, mw.axis);
      }
      add(v, wv);
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) { acc += a[c]; den += ad[c]; }
    k0 = lead + nvec * NV;
    if (lane < lead) {  // the row's first cells, before its first 16-B boundary
      real num, dn = real(0);
      terms(prow[lane], HAS_W ? wgt[mb + lane * mw.axis] : real(1), num, dn);
      acc += num;
      den += dn;
    }
  }
  for (int64_t k = k0 + lane; k < n; k += WAVE) {
    real num, dn = real(0);
    terms(prow[k], HAS_W ? wgt[mb + k * mw.axis] : real(1), num, dn);
    acc += num;
    den += dn;
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    acc += __shfl_down(acc, d, WAVE);
    den += __shfl_down(den, d, WAVE);
  }
  if (lane == 0) {
    if (pair) { out[row] = acc; out[g.outer + row] = den; }
    else out[row] = mean ? acc / den : acc;
  }
}

// K4w: the same sums with one WORKGROUP per row: the 4 waves of a workgroup read ADJACENT 1-KB pieces of the row, so a
// workgroup streams one contiguous 4-KB window per step (a 28.8-KB row of the bench grid is ONE batch of 8 loads per lane)
// and the launch keeps a quarter as many separate row streams open at any moment as K4b does.  Per-wave shuffle trees, the
// 4 partial sums through LDS in wave order (re-associated like K4b: tolerance parity).  Aligned vector rows only (VEC),
// weights in the WFAST form or none.
template <bool HAS_W, int SK, int RUMAX = 8>
__global__ __launch_bounds__(BLOCK) void k_reduce_contig_wg(const real* __restrict__ in, real* __restrict__ out, Geo g,
                                                            int skipna_rt, const real* __restrict__ wgt, MIdx mw, int ntl, ZBand zb) {
  __shared__ real part[2][WPB];
  int skipna = (SK >= 0) ? SK : skipna_rt;
  // workgroup b runs on XCD b mod 8: the linear row order is cut into 8 contiguous bands, one per XCD (rule 3), so that
  // the levels of a band that share weight rows meet in ONE L2
  u64 row = (u64)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (zb.on) {
    u32 z, y;
    if (row >= (u64)zb.per_band.d * ((zb.Y + zb.B - 1) / zb.B)) return;
    if (!zband_map(zb, (u32)row, z, y)) return;
    row = (u64)z * zb.Y + y;
  }
  if ((int64_t)row >= g.outer) return;
  const int lane = threadIdx.x & 63;
  const int wv_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n = g.n_in;
  const real* prow = in + row * n;
  int64_t mb = 0;
  if (HAS_W) mb = outer_off(g, mw, row);
  const real* wrow = HAS_W ? wgt + mb : nullptr;
  const bool pair = skipna >= 6;
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;
  const int nmode = mean ? (skipna == 4 ? 1 : 0) : skipna, dmode = skipna == 4 ? 2 : 3;
  real acc = real(0), den = real(0);
  auto terms = [&](real v, real wv, real& num, real& dn) {
    if (mean) {
      dn = as_count(v, dmode);
      if (HAS_W) dn = dn * wv;
      dn = nan0(dn);
    }
    if (nmode >= 2) v = as_count(v, nmode);
    if (HAS_W) v = v * wv;
    if (nmode) v = nan0(v);
    num = v;
  };
  const int64_t lead = (NV - (int64_t)((row * (u64)n) % NV)) % NV;
  dv a = splat<dv>(real(0)), ad = splat<dv>(real(0));
  const int64_t nvec = (n - lead) / NV;
  auto ldv = [&](int64_t k) -> dv {
    return (ntl & 1) ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(prow + k)) : *reinterpret_cast<const dv*>(prow + k);
  };
  auto add = [&](const dv& v, const dv& wv) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      real num, dn = real(0);
      terms(v[c], wv[c], num, dn);
      a[c] += num;
      ad[c] += dn;
    }
  };
  int64_t t = threadIdx.x;  // vector index within the row: thread id == memory order across the whole workgroup
  auto batches = [&](auto ru) {
    constexpr int RU = decltype(ru)::value;
    for (; t + (RU - 1) * BLOCK < nvec; t += RU * BLOCK) {
      dv v[RU], wv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int64_t k = lead + (t + u * BLOCK) * NV;
        v[u] = ldv(k);
        if constexpr (!HAS_W) wv[u] = splat<dv>(real(1));
        else wv[u] = *reinterpret_cast<const dv*>(wrow + k);
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) add(v[u], wv[u]);
    }
  };
  if constexpr (RUMAX >= 8) batches(std::integral_constant<int, 8>{});
  batches(std::integral_constant<int, 4>{});
  batches(std::integral_constant<int, 2>{});
  for (; t < nvec; t += BLOCK) {
    const int64_t k = lead + t * NV;
    const dv v = ldv(k);
    dv wv = splat<dv>(real(1));
    if constexpr (HAS_W) wv = *reinterpret_cast<const dv*>(wrow + k);
    add(v, wv);
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) { acc += a[c]; den += ad[c]; }
  if (wv_id == 0) {  // the cells before the row's first / after its last 16-B boundary (<= NV - 1 each)
    const int64_t k0 = lead + nvec * NV;
    if (lane < lead) {
      real num, dn = real(0);
      terms(prow[lane], HAS_W ? wrow[lane] : real(1), num, dn);
      acc += num;
      den += dn;
    }
    if (k0 + lane < n) {
      real num, dn = real(0);
      terms(prow[k0 + lane], HAS_W ? wrow[k0 + lane] : real(1), num, dn);
      acc += num;
      den += dn;
    }
  }
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) {
    acc += __shfl_down(acc, d, WAVE);
    den += __shfl_down(den, d, WAVE);
  }
  if (lane == 0) { part[0][wv_id] = acc; part[1][wv_id] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    real s = part[0][0], sd = part[1][0];
#pragma unroll
    for (int w = 1; w < WPB; ++w) { s += part[0][w]; sd += part[1][w]; }
    if (pair) { out[row] = s; out[g.outer + row] = sd; }
    else out[row] = mean ? s / sd : s;
  }
}

// K4wz: weighted sums along the contiguous axis whose weights are shared by the levels (dx(Y, X) under (Z, Y, X)): one
// workgroup takes the SAME row of ZL consecutive levels and loads the row's weight vectors ONCE for all of them (rule 8: an
// L2 hit is not free) -- a row that fits one batch (<= RU * BLOCK vectors) keeps them in registers.  WFAST form only.
template <int SK, int ZL>
__global__ __launch_bounds__(BLOCK) void k_reduce_contig_wgz(const real* __restrict__ in, real* __restrict__ out, Geo g,
                                                             int skipna_rt, const real* __restrict__ wgt, MIdx mw, int ntl, ZBand zb,
                                                             u32 Z) {
  constexpr int RU = 8;
  __shared__ real part[2][ZL][WPB];
  int skipna = (SK >= 0) ? SK : skipna_rt;
  const u32 r = (u32)((u64)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  u32 zg, y;
  if (r >= (u64)zb.per_band.d * ((zb.Y + zb.B - 1) / zb.B)) return;
  if (!zband_map(zb, r, zg, y)) return;
  const u32 z0 = zg * ZL;
  const int lane = threadIdx.x & 63;
  const int wv_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n = g.n_in;
  const int64_t nvec = n / NV;  // WFAST: rows start on 16-B boundaries and hold whole vectors
  const u64 row0 = (u64)z0 * zb.Y + y;
  const real* wrow = wgt + outer_off(g, mw, row0);
  const bool pair = skipna >= 6;
  if (pair) skipna -= 2;
  const bool mean = skipna >= 4;
  const int nmode = mean ? (skipna == 4 ? 1 : 0) : skipna, dmode = skipna == 4 ? 2 : 3;
  auto terms = [&](real v, real wv, real& num, real& dn) {
    if (mean) {
      dn = as_count(v, dmode);
      dn = dn * wv;
      dn = nan0(dn);
    }
    if (nmode >= 2) v = as_count(v, nmode);
    v = v * wv;
    if (nmode) v = nan0(v);
    num = v;
  };
  auto ldv = [&](const real* p) -> dv {
    return (ntl & 1) ? __builtin_nontemporal_load(reinterpret_cast<const dv*>(p)) : *reinterpret_cast<const dv*>(p);
  };
  dv a[ZL], ad[ZL];
#pragma unroll
  for (int l = 0; l < ZL; ++l) { a[l] = splat<dv>(real(0)); ad[l] = splat<dv>(real(0)); }
  for (int64_t t0 = threadIdx.x; t0 < nvec; t0 += (int64_t)RU * BLOCK) {
    dv wv[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int64_t t = t0 + (int64_t)u * BLOCK;
      wv[u] = t < nvec ? *reinterpret_cast<const dv*>(wrow + t * NV) : splat<dv>(real(0));
    }
#pragma unroll
    for (int l = 0; l < ZL; ++l) {
      if (z0 + l >= Z) break;
      const real* prow = in + (row0 + (u64)l * zb.Y) * n;
      dv v[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int64_t t = t0 + (int64_t)u * BLOCK;
        if (t < nvec) v[u] = ldv(prow + t * NV);
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int64_t t = t0 + (int64_t)u * BLOCK;
        if (t < nvec) {
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            real num, dn = real(0);
            terms(v[u][c], wv[u][c], num, dn);
            a[l][c] += num;
            ad[l][c] += dn;
          }
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < ZL; ++l) {
    real acc = real(0), den = real(0);
#pragma unroll
    for (int c = 0; c < NV; ++c) { acc += a[l][c]; den += ad[l][c]; }
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) {
      acc += __shfl_down(acc, d, WAVE);
      den += __shfl_down(den, d, WAVE);
    }
    if (lane == 0) { part[0][l][wv_id] = acc; part[1][l][wv_id] = den; }
  }
  __syncthreads();
  if (threadIdx.x < ZL && z0 + threadIdx.x < Z) {
    const int l = threadIdx.x;
    real s = part[0][l][0], sd = part[1][l][0];
#pragma unroll
    for (int w = 1; w < WPB; ++w) { s += part[0][l][w]; sd += part[1][l][w]; }
    const u64 row = row0 + (u64)l * zb.Y;
    if (pair) { out[row] = s; out[g.outer + row] = sd; }
    else out[row] = mean ? s / sd : s;
  }
}

}  // namespace

// geometry of the marching twin (8-byte lanes, XCD-banded wave order) that follows every chained launch as its rescue
inline int rescue_grid(const Geo& g, u32* ntile, u64* nblocks) {
  *ntile = ceil_div_u32(g.inner, (int64_t)WAVE * HV);
  const u64 ntask = (u64)*ntile * (u64)g.outer;
  *nblocks = (((ntask + WPB - 1) / WPB + 7) / 8) * 8;
  return check_grid(*nblocks);
}

// Launch plan of the chained kernels (K5c / K4c): false when the march is not long-and-narrow enough, the sizes do
// not fit the 32-bit task arithmetic, the device's workgroup -> XCD mapping was not confirmed, or the workspace cannot
// be had (the caller then takes the marching kernel; an allocation failure leaves its message in xg_last_error).
// R = rows per chunk (16 halves the registers but doubles the links of every chain: 63 % against 69 %).
bool chain_plan(const Geo& g, int R, int sums_per_lane, bool shared_metric, void* stream, ChainArgs* ch, u32* ctile_out, u64* nblk_out,
                int zl = 1, bool stores = false) {
  // zl: outer indices ("levels") one task carries for its x-tile; a column of the plan is then (level group, x-tile)
  const u64 lanes = (u64)g.inner / HV, ctile = (lanes + WAVE - 1) / WAVE, ncol = ctile * (((u64)g.outer + zl - 1) / zl);
  const u64 nchunk = ((u64)g.n_in + R - 1) / R;
  // a long march that STORES (the scans) is better off chained however many columns there are, as long as its rows are
  // short (few x-tiles: Y of (Z, Y, X); measured +6 ... +18 points on six shapes from 40 to 4000 outer indices, while
  // whole-plane rows -- the Z axis -- lose 4); the weighted reductions only while the columns are few
  // (sum(T * dy) along Y of (2000, 300, 1024): 72 % marching, 54 % chained)
  const bool few = ncol < (u64)tune().deep_waves;
  // (columns of 64 rows and more: +10-12 points at n = 64 ... 250; shorter ones gain 3-6 without metrics but lose with
  // them -- a ragged single chunk -- so they stay with the march)
  const bool wanted = tune().scan_chain >= 3 ? true : tune().scan_chain >= 2 ? nchunk >= 2
                      : (stores && ctile <= (u64)tune().seg_max_tiles) ? g.n_in >= 64 : (g.n_in >= 256 && few);
  const u64 slot_bytes = (u64)g.outer * 2 * lanes * 16 * (u64)sums_per_lane;
  if (!wanted || nchunk >= (1u << 20) || ncol >= 0x7fffffffull || slot_bytes > (1ull << 30) || !xg_internal_chain_ok()) return false;
  ch->nchunk = (u32)nchunk;
  ch->ncol = (u32)ncol;
  ch->cpx = (u32)((ncol + 7) / 8);
  const int lv = tune().scan_chain_w < 1 ? 1 : tune().scan_chain_w;
  // whole levels side by side in a sub-band: at least ~56 chains must advance together or the hand-off latency of the
  // 75 links of a column, not the bandwidth, sets the pace (f32 rows have half the tiles of f64 rows: two levels)
  const u64 min_lv = tune().scan_chain_w == 1 ? (56 + ctile - 1) / ctile : 1;
  ch->W = (u32)(ctile * (u64)(lv >= 100 ? (lv - 100 < 1 ? 1 : lv - 100) : (lv > (int)min_lv ? (u64)lv : min_lv)));
  // (whole-plane rows, the Z axis, in column chunks of 256 ... 4096 tiles: 65-66 % chained, 66.5 % marching: march kept)
  if (shared_metric && lv < 100) ch->W = ch->cpx;  // (>= 100: experiment, sub-bands of lv - 100 levels whatever the metric)
  ch->tmaj = 0;
  if (shared_metric && tune().scan_chain_tmaj) {
    // A metric shared by the outer indices ("levels"): number the columns x-tile-major and let a sub-band be ALL levels of
    // a few x-tiles.  The chunk's metric rows are then fetched once for the whole chip (level-major: once per XCD, whose
    // band holds a few levels of every tile), and the chunks of a column follow each other after W tasks instead of after
    // a whole band: the 16-byte hand-off slots stay in the L2 (band-wide they were written back to HBM and read again:
    // +0.6 GB of writes for `integrate` along Y, PMC traffic 1.215x).  The price: an XCD works on all levels at once.
    // MEASURED (profiles/history/r03o_*): traffic as predicted -- cumint Y 1.084x -> 1.008x, integrate Y 1.215x -> 1.018x -- and the
    // kernels 10 % / 27 % SLOWER (1.77 -> 1.94 ms, 1.13 -> 1.44 ms): the level-major sweep is worth more than the bytes,
    // which the Infinity Cache absorbs.  Off by default; kept as the evidence that the extra traffic is a choice.
    // (Round 4: the band-wide order with the output rows stored `sc1 nt` -- dropped from the L2 as written, so that they
    // would not push the hand-off slots out -- reads 5.87 -> 6.08 GB and is no faster; profiles/history/r04b_ab_chain_pmc.log.)
    const u64 nout = ncol / ctile;  // outer indices (level groups) per x-tile
    if (nout >= 2 && nout < 0x7fffffffull) {
      ch->tmaj = (u32)nout;
      // (scan_chain_tmaj = k > 1: all levels of k x-tiles side by side -- k times the chains in flight, k times the
      // distance between the chunks of a column.  Round 4, paired over 5 placements, cumint Y: k = 1 / 2 / 3 / 4 / 6 -> time
      // +7.3 / +3.0 / +2.9 / +2.5 / +2.7 %, traffic 1.008 / 1.010 / 1.013 / 1.017 / 1.026x against 1.106x: the stall of the
      // single-tile order goes with the second tile, a 2.5 % cost of sweeping all levels at once stays.  Still off: the
      // default is the fastest order; k = 4 is the setting for whoever shares the HBM.  profiles/history/r04q_ab_tmaj_*.log)
      const u64 tiles = (56 + nout - 1) / nout > (u64)tune().scan_chain_tmaj ? (56 + nout - 1) / nout : (u64)tune().scan_chain_tmaj;
      ch->W = (u32)(nout * tiles);
    }
  }
  if (ch->W > ch->cpx) ch->W = ch->cpx;
  ch->srow = (u32)lanes;
  const u64 nblk = ((u64)ch->cpx * nchunk + WPB - 1) / WPB;
  if (nblk * 8 > 0x7fffffffull || (u64)ch->cpx * nchunk >= 0xffffffffull) return false;
  ch->nblk = (u32)nblk;
  ChainWs ws;
  if (xg_internal_chain_ws(stream, slot_bytes, &ws)) return false;
  ch->ticket = ws.ticket;
  ch->gave_up = ws.gave_up;
  ch->poison = ws.poison;
  ch->slots = ws.slots;
  ch->nslot = ws.slot_bytes / 16;
  ch->spin = tune().scan_chain_spin < 1 ? 1u : (u32)tune().scan_chain_spin;
  *ctile_out = (u32)ctile;
  *nblk_out = nblk;
  return true;
}

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int XG_FN(xg_cumsum1d)(const real* in, real* out, const int64_t* shape, int ndim, int axis, int reverse,
                    int skipna, int trim_lo, int trim_hi, int pad_lo, int pad_hi, int bc, real fill,
                    const real* m_in, const int64_t* m_in_strides, const real* m_out,
                    const int64_t* m_out_strides, void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if ((trim_lo | trim_hi | pad_lo | pad_hi) & ~1) return fail(XG_ERR_INVALID, "trim/pad widths must be 0 or 1");
  if (bc < XG_BC_NONE || bc > XG_BC_EXTEND) return fail(XG_ERR_INVALID, "unknown boundary mode %d", bc);
  if ((pad_lo || pad_hi) && bc == XG_BC_NONE) return fail(XG_ERR_INVALID, "halo cells requested but no boundary mode given");
  if ((m_in && !m_in_strides) || (m_out && !m_out_strides)) return fail(XG_ERR_INVALID, "metric without strides");
#ifdef XG_I64
  if (m_in || m_out) return fail(XG_ERR_UNSUPPORTED, "integer scans take no metrics: convert to float64 first (numpy promotes int * float)");
#endif
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
    if (tune().scan_vec && n_out >= 3 * NV && aligned16(out) && nblocks < 0x7ffffff0ull &&
        g.n_in < 0x7fff0000ll && n_out < 0x7fff0000ll) {
      const u32 nrows = (u32)nblocks;
      ZBand zb = make_zband(false, 0, 0, 1);
      u32 nwork = nrows;
      if (met && tune().zband && g.n_outer == 2 && g.outer_shape[0] >= 2 && (!m_in || mi.outer[0] == 0) && (!m_out || mo.outer[0] == 0)) {
        const u64 Z = (u64)g.outer_shape[0], Y = (u64)g.outer_shape[1];
        const u32 B = (u32)(tune().zb_rows > 0 ? tune().zb_rows : 16) * ((m_in && m_out) ? 1u : 2u);
        const u64 padded = ((Y + B - 1) / B) * B * Z;
        if (padded < 0x7ffffff0ull) {
          zb = make_zband(true, Z, Y, B);
          if (zb.on) nwork = (u32)padded;
        }
      }
      const u32 grid = ((nwork + 7) / 8) * 8;
      const bool nts = tune().nt_store;
      // rows of float32 are half as long in bytes: 128 threads per row measured -4.6 % on the plain scan there (0.710 -> 0.744;
      // with a metric +-0 when round 4 measured it, float64 +4 %: profiles/history/r04aa_ab_f32_rowshapes.log).  Since the
      // shifted weighted scan (the Grid's cumint X: `scan_sh1`) 128 wins there too: 1.131 -> 0.994 ms, 0.577 -> 0.656 of
      // 8 TB/s, the unshifted one unchanged (profiles/r06_kernels/r06ak_ab_scan_block_f32.log); 256 for 8-byte elements
      const int bs = tune().scan_block ? tune().scan_block : (sizeof(real) == 4 ? 128 : 256);
#define XG_L(M, NTS_, BS_) hipLaunchKernelGGL((k_cumsum_contig_vec<M, NTS_, BS_>), dim3(grid), dim3(BS_), 0, st, in, out, g, nrows, a, m_in, mi, m_out, mo, zb, nwork, (tune().scan_dpp ? 1 : 0) | (tune().scan_sh1 ? 2 : 0))
#define XG_B(M, NTS_) do { if (bs == 512) XG_L(M, NTS_, 512); else if (bs == 1024) XG_L(M, NTS_, 1024); else if (bs == 128) XG_L(M, NTS_, 128); else if (bs == 64) XG_L(M, NTS_, 64); else XG_L(M, NTS_, 256); } while (0)
#define XG_M(M) do { if (nts) XG_B(M, true); else XG_B(M, false); } while (0)
      switch (met) { case 0: XG_M(0); break; case 1: XG_M(1); break; case 2: XG_M(2); break; default: XG_M(3); }
#undef XG_B
#undef XG_L
#undef XG_M
    } else {
      if (g.n_in >= 0x7fff0000ll || n_out >= 0x7fff0000ll) return fail(XG_ERR_UNSUPPORTED, "rows of 2^31 cells or more along the contiguous scan axis");
#define XG_M(M) hipLaunchKernelGGL((k_cumsum_contig<M>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, a, m_in, mi, m_out, mo)
      switch (met) { case 0: XG_M(0); break; case 1: XG_M(1); break; case 2: XG_M(2); break; default: XG_M(3); }
#undef XG_M
    }
  } else {
    int V = (aligned16(in) && aligned16(out) && (g.inner % NV == 0) && vec_metric_ok(g, met != 0)) ? NV : 1;
    // few, long columns (cumsum along Y of (Z,Y,X): ~2k wave-tasks for 1024 SIMDs): one element per
    // lane doubles (f64) / quadruples (f32) the number of independent marches
    const bool long_march = g.n_in >= 256;
    if (V > 1 && long_march && (u64)ceil_div_u32(g.inner, (int64_t)WAVE * V) * (u64)g.outer < (u64)tune().scan_narrow_below) V = HV;  // 8-byte lanes
    const u32 ntile = ceil_div_u32(g.inner, (int64_t)WAVE * V);
    const u64 ntask = (u64)ntile * (u64)g.outer;
    const u64 nblocks = tune().march_band ? (((ntask + WPB - 1) / WPB + 7) / 8) * 8 : (ntask + WPB - 1) / WPB;
    if ((rc = check_grid(nblocks))) return rc;
    const bool nts = tune().nt_store;
    // few columns, long march (cumsum along Y: ~2k waves for the whole chip): occupancy cannot hide
    // the latency, so keep 16 loads in flight per lane instead of 4 (measured with the narrow lanes
    // above: cumsum along Y f32 48 -> 55 %, sum along Y 59 -> 67 % f32 / 68 -> 71 % f64)
    const bool deep = long_march && ntask < (u64)tune().deep_waves;
    // rolling-window variants exist for the default non-temporal loads + stores only; `pipe`: window length
    // window length: `scan_u` rows; with an input metric the window holds the metric values too (twice the registers:
    // 32 rows -> 256 VGPRs, ONE wave per SIMD, cumint along Y 40 % of 8 TB/s) => 8 rows
    // K5c: the long march as a chained flat launch (8-byte lanes, default cache policies)
    if (tune().scan_chain && nts && tune().nt_load && g.inner % HV == 0 && (reinterpret_cast<uintptr_t>(in) & 7u) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (HV == 1 || vec_metric_ok(g, met != 0))) {
      // a metric that does not depend on the outer index (dy(Y, X) under a (Z, Y, X) field): all columns of the
      // band advance side by side, so the metric rows of a chunk are read once per XCD and found in its L2 by the
      // other levels -- level-major order would stream the whole metric from HBM once per level (+50 % traffic)
      bool shared_metric = met != 0;
      for (int d = 0; d < g.n_outer; ++d) {
        if ((met & 2) && mi.outer[d] != 0) shared_metric = false;
        if ((met & 1) && mo.outer[d] != 0) shared_metric = false;
      }
      ChainArgs ch;
      u32 ctile = 0;
      u64 nblk = 0;
      if (chain_plan(g, 32, 1, shared_metric, stream, &ch, &ctile, &nblk, 1, true)) {
#define XG_C(M, R_) hipLaunchKernelGGL((k_cumsum_chain<M, R_>), dim3((u32)(nblk * 8)), dim3(BLOCK), 0, st, in, out, g, ctile, a, m_in, mi, m_out, mo, ch)
        u32 rtile = 0;
        u64 rblocks = 0;
        if ((rc = rescue_grid(g, &rtile, &rblocks))) return rc;
        const Rescue rs = {ch.poison, ch.gave_up, reinterpret_cast<u32x4*>(ch.slots), ch.nslot};
        switch (met) { case 0: XG_C(0, 32); break; case 1: XG_C(1, 32); break; case 2: XG_C(2, 32); break; default: XG_C(3, 32); }
#undef XG_C
        XG_LAUNCH_CHECK();
        // the marching twin, run-if poisoned (k_cumsum_rescue): leaves at once unless a chunk of the launch above gave up
#define XG_R(M) hipLaunchKernelGGL((k_cumsum_rescue<M>), dim3((u32)rblocks), dim3(BLOCK), 0, st, in, out, g, rtile, a, m_in, mi, m_out, mo, 1, rs)
        switch (met) { case 0: XG_R(0); break; case 1: XG_R(1); break; case 2: XG_R(2); break; default: XG_R(3); }
#undef XG_R
        XG_LAUNCH_CHECK();
        return XG_OK;
      }
    }
    const int su = (met & 2) ? (tune().scan_u < 8 ? tune().scan_u : 8) : tune().scan_u;
    const int pipe = (tune().scan_pipe && nts && tune().nt_load) ? (deep ? (su >= 32 ? 32 : su >= 24 ? 24 : su >= 16 ? 16 : 8) : (tune().scan_pipe >= 2 ? 8 : 0)) : 0;
#define XG_PL(V_, M, U_) hipLaunchKernelGGL((k_cumsum_strided<V_, M, true, true, U_, true>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, a, m_in, mi, m_out, mo, (tune().march_band ? 1 : 0) | (tune().scan_pace ? 2 : 0))
#define XG_GL(V_, M, NTL_, NTS) do { if (deep) hipLaunchKernelGGL((k_cumsum_strided<V_, M, NTL_, NTS, 16>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, a, m_in, mi, m_out, mo, (tune().march_band ? 1 : 0) | (tune().scan_pace ? 2 : 0)); \
                               else hipLaunchKernelGGL((k_cumsum_strided<V_, M, NTL_, NTS, 4>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, a, m_in, mi, m_out, mo, (tune().march_band ? 1 : 0) | (tune().scan_pace ? 2 : 0)); } while (0)
#define XG_GO(V_, M, NTS) do { if (tune().nt_load) XG_GL(V_, M, true, NTS); else XG_GL(V_, M, false, NTS); } while (0)
#define XG_M(V_, M) do { if (pipe == 32) XG_PL(V_, M, 32); else if (pipe == 24) XG_PL(V_, M, 24); else if (pipe == 16) XG_PL(V_, M, 16); else if (pipe == 8) XG_PL(V_, M, 8); \
                         else if (nts) XG_GO(V_, M, true); else XG_GO(V_, M, false); } while (0)
#define XG_V(V_) switch (met) { case 0: XG_M(V_, 0); break; case 1: XG_M(V_, 1); break; case 2: XG_M(V_, 2); break; default: XG_M(V_, 3); }
#ifdef XG_F32
    if (V == HV) { XG_V(HV) } else
#endif
    if (V > 1) { XG_V(NV) } else { XG_V(1) }
#undef XG_V
#undef XG_M
#undef XG_GO
#undef XG_GL
#undef XG_PL
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

int XG_FN(xg_reduce1d)(const real* in, real* out, const int64_t* shape, int ndim, int axis, int skipna,
                    const real* w, const int64_t* w_strides, void* stream) {
  if (!in || !out || !shape) return fail(XG_ERR_INVALID, "NULL array argument");
  if (w && !w_strides) return fail(XG_ERR_INVALID, "weight without strides");
  if (skipna < 0 || skipna > 7) return fail(XG_ERR_INVALID, "skipna / count / mean mode %d not in [0,7]", skipna);
#ifdef XG_I64
  if (w || skipna > 1) return fail(XG_ERR_UNSUPPORTED, "integer reductions are plain sums: weights and means are float (convert first)");
#endif
  Geo g; MIdx mw;
  int rc = build_geo(shape, ndim, axis, 1, w ? w_strides : nullptr, nullptr, &g, &mw, nullptr);
  if (rc) return rc;
  if (g.outer == 0 || g.inner == 0) return XG_OK;
  hipStream_t st = (hipStream_t)stream;
  if (g.n_in == 0) { XG_HIP(hipMemsetAsync(out, 0, sizeof(real) * g.outer * g.inner, st)); return XG_OK; }
  if (g.inner == 1) {
    ZBand zb = make_zband(false, 0, 0, 1);
    u64 nrows = (u64)g.outer;
    if (w && tune().zband && g.n_outer == 2 && mw.outer[0] == 0 && g.outer_shape[0] >= 2) {
      const u64 Z = (u64)g.outer_shape[0], Y = (u64)g.outer_shape[1];
      const u32 B = (u32)(tune().zb_rows > 0 ? tune().zb_rows : 16) * 2u;  // one metric: double-height bands
      const u64 padded = ((Y + B - 1) / B) * B * Z;
      if (padded < 0x7fffffffull) {
        zb = make_zband(true, Z, Y, B);
        if (zb.on) nrows = padded;
      }
    }
    const u64 nblocks = (nrows + WPB - 1) / WPB;
    if ((rc = check_grid(nblocks))) return rc;
    const bool vec = aligned16(in) && g.n_in >= 4 * NV;  // rows of any length: lead / tail cells go through scalar loads
    const int ntf = (tune().nt_load ? 1 : 0) | (tune().reduce_ru == 1 ? 2 : tune().reduce_ru >= 2 ? 4 : 0);
    bool wfast = w && vec && tune().reduce_wfast && mw.axis == 1 && g.n_in % NV == 0 && aligned16(w);
    for (int d = 0; d < g.n_outer && wfast; ++d)
      if (g.outer_shape[d] > 1 && mw.outer[d] % NV) wfast = false;
    const int sk = (tune().reduce_sk && (skipna == 0 || skipna == 1)) ? skipna : -1;
#define XG_RC(W_, V_, S_) hipLaunchKernelGGL((k_reduce_contig<W_, V_, S_>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw, ntf, zb)
#define XG_RS(W_, V_) do { if (sk == 0) XG_RC(W_, V_, 0); else if (sk == 1) XG_RC(W_, V_, 1); else XG_RC(W_, V_, -1); } while (0)
#define XG_RF(S_) hipLaunchKernelGGL((k_reduce_contig<true, true, S_, true>), dim3((u32)nblocks), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw, ntf, zb)
    // K4w: a workgroup per row (weights as in WFAST: unit stride, rows 16-B aligned like the field's)
    // (reduce_wg: bit 0 the unweighted sums, bit 1 the weighted ones, bit 2 the weighted ones with batches of 4 loads)
    const int wgm = tune().reduce_wg;
    const bool wg_ok = (w ? (wgm & 6) : (wgm & 1)) && vec && (!w || wfast) && nrows <= 0x7ffffff0ull && g.n_in >= (int64_t)BLOCK * NV;
#define XG_RW(W_, S_, R_) hipLaunchKernelGGL((k_reduce_contig_wg<W_, S_, R_>), dim3((u32)(((nrows + 7) / 8) * 8)), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw, ntf, zb)
    // K4wz: level-shared weights, ZL levels of one row per workgroup (reduce_wg bits 3 / 4: ZL = 2 / 4)
    // (8-byte elements only: with float32's 4-cell vectors the weighted form gains nothing at 2 levels -- 0.707 -> 0.702 -- and
    // spills at 4 -- 2.2 ms; profiles/history/r05y_ab_reduce_f32.log)
    if (sizeof(real) == 8 && w && wfast && zb.on && (wgm & 24) && vec && g.n_in >= (int64_t)BLOCK * NV) {
      const u32 Z = (u32)g.outer_shape[0];
      const int ZL = (wgm & 16) ? 4 : 2;
      const u64 Zg = ((u64)Z + ZL - 1) / ZL;
      ZBand zg = make_zband(true, Zg, (u64)g.outer_shape[1], zb.B);
      const u64 nwork = ((((u64)g.outer_shape[1] + zb.B - 1) / zb.B) * zb.B) * Zg;
      if (zg.on && nwork < 0x7ffffff0ull) {
        const u32 grid = (u32)(((nwork + 7) / 8) * 8);
#define XG_RZ(S_, L_) hipLaunchKernelGGL((k_reduce_contig_wgz<S_, L_>), dim3(grid), dim3(BLOCK), 0, st, in, out, g, skipna, w, mw, ntf, zg, Z)
        if (ZL == 4) { if (sk == 0) XG_RZ(0, 4); else if (sk == 1) XG_RZ(1, 4); else XG_RZ(-1, 4); }
        else { if (sk == 0) XG_RZ(0, 2); else if (sk == 1) XG_RZ(1, 2); else XG_RZ(-1, 2); }
#undef XG_RZ
        XG_LAUNCH_CHECK();
        return XG_OK;
      }
    }
    if (wg_ok) {
      if (w && (wgm & 4)) { if (sk == 0) XG_RW(true, 0, 4); else if (sk == 1) XG_RW(true, 1, 4); else XG_RW(true, -1, 4); }
      else if (w) { if (sk == 0) XG_RW(true, 0, 8); else if (sk == 1) XG_RW(true, 1, 8); else XG_RW(true, -1, 8); }
      else { if (sk == 0) XG_RW(false, 0, 8); else if (sk == 1) XG_RW(false, 1, 8); else XG_RW(false, -1, 8); }
    } else
    if (wfast) { if (sk == 0) XG_RF(0); else if (sk == 1) XG_RF(1); else XG_RF(-1); }
    else
    if (vec) { if (w) XG_RS(true, true); else XG_RS(false, true); }
    else { if (w) XG_RS(true, false); else XG_RS(false, false); }
#undef XG_RW
#undef XG_RF
#undef XG_RS
#undef XG_RC
  } else {
    int V = (aligned16(in) && aligned16(out) && (g.inner % NV == 0) && vec_metric_ok(g, w != nullptr)) ? NV : 1;
    const bool long_march = g.n_in >= 256;
    if (V > 1 && long_march && (u64)ceil_div_u32(g.inner, (int64_t)WAVE * V) * (u64)g.outer < (u64)tune().scan_narrow_below) V = HV;  // 8-byte lanes
    // (float32 with 4-byte lanes -- as many marches as float64 -- measured: sum along Y 0.652 -> 0.593; 8-byte lanes stay;
    // profiles/history/r03ao_ab_f32_narrow4.jsonl)
    const u32 ntile = ceil_div_u32(g.inner, (int64_t)WAVE * V);
    const u64 ntask = (u64)ntile * (u64)g.outer;
    const bool deep = long_march && ntask < (u64)tune().deep_waves;
    int rband = tune().march_band ? 1 : 0;
    // one weight per row (no inner stride: drF(Z) under (Z, Y, X), dy(Y) under (Z, Y, X)): scalar loads, nothing rides
    // in the window -- the march then streams like the unweighted one (sum along Y 55 -> 78 %; the chain below: 62 %)
    bool wu = w && tune().met_scalar && g.outer < 0xffffffffll;
    for (int d = 0; d < g.n_inner; ++d)
      if (w && mw.inner[d] != 0) wu = false;
    // K4c: the long weighted march as a chained flat launch (the unweighted one already streams at 80 %)
    // (the UNWEIGHTED long march as a chain: measured 59 % against 76 % (f64), 39 % against 63 % (f32) -- no stores, no
    // weight window: nothing for the chain to fix, only its hand-offs to pay)
    if (w && !wu && tune().march_ofast) {
      bool shared = true;
      for (int d = 0; d < g.n_outer; ++d)
        if (mw.outer[d] != 0) shared = false;
      if (shared && g.outer >= 2) rband |= 2;
    }
    // any banded order walks ceil(grid / 8) workgroups per XCD band: the grid must be a multiple of 8 whenever `rband` is
    // non-zero, also when only the outer-fastest order (bit 1) asked for it (march_band = 0, march_ofast = 1)
    const u64 nblocks = rband ? (((ntask + WPB - 1) / WPB + 7) / 8) * 8 : (ntask + WPB - 1) / WPB;
    if ((rc = check_grid(nblocks))) return rc;
    if (w && !wu && tune().scan_chain && tune().nt_load && g.inner % HV == 0 && (reinterpret_cast<uintptr_t>(in) & 7u) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (HV == 1 || vec_metric_ok(g, true))) {
      bool shared_w = true;
      for (int d = 0; d < g.n_outer; ++d)
        if (mw.outer[d] != 0) shared_w = false;
      // weights with an outer dim of their own (hFacC(Z, Y, X)) stream from HBM like the field: the march's window of
      // 8 rows + 8 weight rows already runs at 76 % there, the chain at 74 %
      ChainArgs ch;
      u32 ctile = 0;
      u64 nblk = 0;
      // K4Z: ZL levels per wave share the weight row in registers (round 6)
      // (float32 with 1000 + value -- the default: float32 rows have half the x-tiles and K4Z few waves, yet it beats K4L
      // since the weight-load form left the row loop: sum 0.56 -> 0.70 of 8 TB/s, mean 0.54 -> 0.59, profiles/r06_kernels/r06ai_*.
      // A/B on 75 x 2400 x 3600 f64, paired over 6 placements: integrate Y 0.953 -> 0.872 ms (0.689 -> 0.753 of 8 TB/s) with
      // 3 levels x 12 rows in flight, 0.891 with 3 x 8, 0.949 with 2 x 8, 0.988 with 4 x 8; average Y 0.962 -> 0.918 (3 x 8;
      // 3 x 12: 0.932, 3 x 16: 1.34))
      // (`reduce_ldsw` = 0 switches BOTH level-sharing marches off: the chained kernels / the plain march below)
      if (shared_w && tune().reduce_zmarch && tune().reduce_ldsw && (sizeof(real) == 8 || tune().reduce_zmarch >= 1000) && g.outer >= 3 && g.n_in >= 64 &&
          g.outer < 0x7fffffffll) {
        const int zm = tune().reduce_zmarch % 1000;  // 100 * ZL + U
        const int zl = zm / 100 < 2 ? 3 : (zm / 100 > 4 ? 4 : zm / 100);
        const u32 ztile = ceil_div_u32(g.inner, (int64_t)WAVE * HV);
        const u64 zgrp = ((u64)g.outer + zl - 1) / zl, ztask = zgrp * ztile;
        if (ztask < 0x7ffffff0ull) {
          const u32 zgrid = (u32)(((ztask + 7) / 8) * 8);
#define XG_ZM(U_, ZL_, M_) hipLaunchKernelGGL((k_reduce_zmarch<U_, ZL_, M_>), dim3(zgrid), dim3(WAVE), 0, st, in, out, g, ztile, (u32)zgrp, (u32)ztask, skipna, w, mw)
#define XG_Z(U_, ZL_) do { if (skipna) XG_ZM(U_, ZL_, 1); else XG_ZM(U_, ZL_, 0); } while (0)
          // the count / mean / pair modes (two running sums per level in the means): the 8-row window
#define XG_ZC(ZL_) do { if (skipna == 2) XG_ZM(8, ZL_, 2); else if (skipna == 3) XG_ZM(8, ZL_, 3); else if (skipna == 4 || skipna == 6) XG_ZM(8, ZL_, 4); else XG_ZM(8, ZL_, 5); } while (0)
          const int zu = zm % 100;
          if (skipna >= 2) { if (zl == 2) XG_ZC(2); else if (zl == 4) XG_ZC(4); else XG_ZC(3); }
          else if (zl == 2) { if (zu >= 24) XG_Z(24, 2); else if (zu >= 16) XG_Z(16, 2); else if (zu >= 12) XG_Z(12, 2); else XG_Z(8, 2); }
          else if (zl == 4) XG_Z(8, 4);
          else { if (zu >= 16) XG_Z(16, 3); else if (zu >= 12) XG_Z(12, 3); else XG_Z(8, 3); }
#undef XG_ZC
#undef XG_Z
#undef XG_ZM
          XG_LAUNCH_CHECK();
          return XG_OK;
        }
      }
      // K4L: the weights of a block of rows once per workgroup through LDS, the waves of a workgroup = consecutive levels
      if (shared_w && tune().reduce_ldsw && g.outer >= 2 && g.n_in >= 64 && g.outer < 0x7fffffffll) {
        // (A/B of the shapes, profiles/history/r03u_ab_ldsw_shapes*.jsonl: blocks of 8 rows x 4 levels 0.69 / 0.68 for sum / mean;
        // 16 x 4: 0.67 / 0.66; 4 x 4: 0.68 / 0.66; 8 x 2: 0.66 / 0.65; 24 x 4: 0.60; 8 levels per workgroup: 0.53-0.54;
        // the chained K4cz on the same box: 0.58 / 0.51)
        constexpr int LU = 8, LW = 4;
        const u32 ltile = ceil_div_u32(g.inner, (int64_t)WAVE * HV);
        const u64 groups = ((u64)g.outer + LW - 1) / LW, lblk = groups * ltile;
        if (lblk < 0x7ffffff0ull) {
          const u32 lgrid = (u32)(((lblk + 7) / 8) * 8);
          // rows per block: 8 in float64 (16: 0.716 -> 0.697); float32 has half the workgroups for the same bytes and wants the
          // longer window (0.518 -> 0.541); profiles/history/r03ap_*
          const int lu = tune().reduce_ldsw_u ? tune().reduce_ldsw_u : (sizeof(real) == 4 ? 16 : 8);
          if (lu >= 16) hipLaunchKernelGGL((k_reduce_ldsw<16, LW>), dim3(lgrid), dim3(LW * WAVE), 0, st, in, out, g, ltile, (u32)lblk, skipna, w, mw, tune().reduce_ldsw >= 2 ? (u32)groups : 0u);
          else
          hipLaunchKernelGGL((k_reduce_ldsw<LU, LW>), dim3(lgrid), dim3(LW * WAVE), 0, st, in, out, g, ltile, (u32)lblk, skipna, w, mw, tune().reduce_ldsw >= 2 ? (u32)groups : 0u);
          XG_LAUNCH_CHECK();
          return XG_OK;
        }
      }
      bool chained = false;
      u32 rtile = 0;
      u64 rblocks = 0;
      if ((rc = rescue_grid(g, &rtile, &rblocks))) return rc;  // before any chained launch: a chain never runs without its twin
      const int zl = tune().reduce_zl;  // levels per task sharing the weight rows (K4cz); 1: K4c
      // (measured: 2 levels x 16 rows 61-63 %, 3 x 16 59 %, 4 x 16 55 %, 4 x 8 59 %, 2 x 32 50 %, K4c 56-57 %, the march 51 %)
      if (shared_w && zl >= 2 && g.outer >= 2 && chain_plan(g, 16, skipna >= 4 ? 2 : 1, shared_w, stream, &ch, &ctile, &nblk, zl >= 4 ? 4 : 2)) {
        if (zl >= 4) hipLaunchKernelGGL((k_reduce_chain_z<16, 4>), dim3((u32)(nblk * 8)), dim3(BLOCK), 0, st, in, out, g, ctile, skipna, w, mw, ch);
        else hipLaunchKernelGGL((k_reduce_chain_z<16, 2>), dim3((u32)(nblk * 8)), dim3(BLOCK), 0, st, in, out, g, ctile, skipna, w, mw, ch);
        XG_LAUNCH_CHECK();
        chained = true;
      } else if (shared_w && chain_plan(g, 32, skipna >= 4 ? 2 : 1, shared_w, stream, &ch, &ctile, &nblk)) {
        hipLaunchKernelGGL((k_reduce_chain<true, 32>), dim3((u32)(nblk * 8)), dim3(BLOCK), 0, st, in, out, g, ctile, skipna, w, mw, ch);
        XG_LAUNCH_CHECK();
        chained = true;
      }
      if (chained) {  // the marching twin, run-if poisoned (k_reduce_rescue)
        const Rescue rs = {ch.poison, ch.gave_up, reinterpret_cast<u32x4*>(ch.slots), ch.nslot};
        hipLaunchKernelGGL(k_reduce_rescue, dim3((u32)rblocks), dim3(BLOCK), 0, st, in, out, g, rtile, skipna, w, mw, 1, rs);
        XG_LAUNCH_CHECK();
        return XG_OK;
      }
    }
#define XG_GL(V_, W_, NTL_) do { if (deep) hipLaunchKernelGGL((k_reduce_strided<V_, W_, NTL_, 16>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, skipna, w, mw, rband); \
                           else hipLaunchKernelGGL((k_reduce_strided<V_, W_, NTL_, 4>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, skipna, w, mw, rband); } while (0)
    const int su = (w && !wu) ? (tune().scan_u < 8 ? tune().scan_u : 8) : tune().scan_u;  // weights ride in the window: 8 rows (registers)
    const int pipe = (tune().scan_pipe && tune().nt_load) ? (deep ? (su >= 32 ? 32 : su >= 24 ? 24 : su >= 16 ? 16 : 8) : (tune().scan_pipe >= 2 ? 8 : 0)) : 0;
#define XG_PL(V_, W_, U_) hipLaunchKernelGGL((k_reduce_strided<V_, W_, true, U_, true>), dim3((u32)nblocks), dim3(BLOCK), march_lds(), st, in, out, g, ntile, skipna, w, mw, rband)
#define XG_GO(V_, W_) do { if (pipe == 32) XG_PL(V_, W_, 32); else if (pipe == 24) XG_PL(V_, W_, 24); else if (pipe == 16) XG_PL(V_, W_, 16); else if (pipe == 8) XG_PL(V_, W_, 8); \
                           else if (tune().nt_load) XG_GL(V_, W_, true); else XG_GL(V_, W_, false); } while (0)
#ifdef XG_F32
    if (V == HV) { if (wu) XG_GO(HV, 2); else if (w) XG_GO(HV, 1); else XG_GO(HV, 0); } else
#endif
    if (V > 1) { if (wu) XG_GO(NV, 2); else if (w) XG_GO(NV, 1); else XG_GO(NV, 0); }
    else { if (wu) XG_GO(1, 2); else if (w) XG_GO(1, 1); else XG_GO(1, 0); }
#undef XG_GO
#undef XG_PL
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

}  // extern "C"
