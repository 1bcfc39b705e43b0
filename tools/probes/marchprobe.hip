// marchprobe.hip -- why does a running sum along Y of (75, 2400, 3600) f64 (4 288 long marches, each reading a row and
// writing a row per step) stay at 60-68 % of 8 TB/s when the read-only march reaches 80 % and the flat copy 81 %?
// Hypothesis: a wave's loads and stores share one in-order vmcnt queue, so every load issued behind a store waits for
// that store's write acknowledgement; flat kernels hide it by wave turnover, a long-lived marching wave cannot.
// Test: split the roles -- a LOADER wave (loads only) hands rows through LDS to a STORER wave (stores only).
// Tuning aid for xgcm_amd/csrc/xg_scan.hip (not part of the product).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/marchprobe.hip -o gpurun_out/marchprobe && gpurun_out/marchprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_rand(double* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31; z *= 0x94D049BB133111EBull; z ^= z >> 29;
    out[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

// A: one wave per (level, x-tile of 64 lanes x 8 B): rolling window of U rows in registers (the product kernel's form)
template <int U>
__global__ __launch_bounds__(256) void k_single(const double* __restrict__ in, double* __restrict__ out, unsigned outer, unsigned n, unsigned inner, unsigned ntile, unsigned ntask) {
  const unsigned nb = gridDim.x, pb = nb >> 3;
  const unsigned b = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const unsigned task = b * 4 + (threadIdx.x >> 6);
  if (task >= ntask) return;
  const unsigned o = task / ntile, tile = task - o * ntile;
  const unsigned x = tile * 64 + (threadIdx.x & 63);
  if (x >= inner) return;
  const double* p = in + (size_t)o * n * inner + x;
  double* q = out + (size_t)o * n * inner + x;
  double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * inner);
  double acc = 0.0;
  unsigned t = 0;
  for (; t + 2 * U <= n; t += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc += v[u];
      v[u] = __builtin_nontemporal_load(p + (size_t)(t + U + u) * inner);
      __builtin_nontemporal_store(acc, q + (size_t)(t + u) * inner);
    }
  }
  for (; t < n; t += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double xv = v[u];
      if (t + U + u < n) v[u] = __builtin_nontemporal_load(p + (size_t)(t + U + u) * inner);
      if (t + u < n) { acc += xv; __builtin_nontemporal_store(acc, q + (size_t)(t + u) * inner); }
    }
  }
}

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// B: workgroup = P pairs of (loader wave, storer wave); a pair owns one (level, x-tile of 64 lanes x 16 B).  The loader
// keeps D groups of G rows in flight in registers and drops each landed group into one of two LDS stages; the storer
// picks it up after the barrier, adds the running sum and stores.  One barrier per group, n % G == 0 assumed.
template <int G, int D, int P>
__global__ __launch_bounds__(128 * P) void k_split(const double* __restrict__ in, double* __restrict__ out, unsigned outer, unsigned n, unsigned inner, unsigned ntile, unsigned ntask) {
  __shared__ d2 stage[P][2][G][64];
  const unsigned nb = gridDim.x, pb = nb >> 3;
  const unsigned b = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned pair = wv >> 1;
  const bool loader = (wv & 1) == 0;
  unsigned task = b * P + pair;
  const bool live = task < ntask;
  if (!live) task = ntask - 1;  // keeps the barrier count of the workgroup uniform
  const unsigned o = task / ntile, tile = task - o * ntile;
  const unsigned x = tile * 128 + lane * 2;
  const bool on = live && x < inner;  // inner % 2 == 0
  const size_t base = (size_t)o * n * inner + (on ? x : 0);
  const unsigned ng = n / G;
  if (loader) {
    // every load of the main loop is issued unconditionally (lanes past the row end read a clamped address): a load
    // under a branch would force the compiler to wait for vmcnt(0) instead of counting the younger loads
    const unsigned xc = (live && x < inner) ? x : inner - 2;
    const d2* p = reinterpret_cast<const d2*>(in + (size_t)o * n * inner + xc);
    const size_t rs = inner / 2;  // row stride in d2
    d2 v[D][G];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < G; ++r) v[d][r] = __builtin_nontemporal_load(p + (size_t)(d * G + r) * rs);
    unsigned g = 0;
    for (; g + 2 * D <= ng; g += D) {  // ng % D == 0
#pragma unroll
      for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int r = 0; r < G; ++r) stage[pair][(g + d) & 1][r][lane] = v[d][r];
#pragma unroll
        for (int r = 0; r < G; ++r) v[d][r] = __builtin_nontemporal_load(p + (size_t)((g + d + D) * G + r) * rs);
        lds_barrier();
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {  // the last D groups: nothing left to load
#pragma unroll
      for (int r = 0; r < G; ++r) stage[pair][(g + d) & 1][r][lane] = v[d][r];
      lds_barrier();
    }
  } else {
    d2* q = reinterpret_cast<d2*>(out + base);
    const size_t rs = inner / 2;
    d2 acc = {0.0, 0.0};
    for (unsigned g = 0; g < ng; ++g) {
      lds_barrier();
      d2 w[G];
#pragma unroll
      for (int r = 0; r < G; ++r) w[r] = stage[pair][g & 1][r][lane];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        acc += w[r];
        if (on) __builtin_nontemporal_store(acc, q + (size_t)(g * G + r) * rs);
      }
    }
  }
}

// C: as B, but the loader's loads go straight into LDS (global_load_lds_dwordx4: no registers, the ring of D + 1 groups
// bounds the bytes in flight)
template <int G, int D, int P>
__global__ __launch_bounds__(128 * P) void k_split_lds(const double* __restrict__ in, double* __restrict__ out, unsigned outer, unsigned n, unsigned inner, unsigned ntile, unsigned ntask) {
  __shared__ d2 ring[P][D + 1][G][64];
  const unsigned nb = gridDim.x, pb = nb >> 3;
  const unsigned b = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned pair = wv >> 1;
  const bool loader = (wv & 1) == 0;
  unsigned task = b * P + pair;
  const bool live = task < ntask;
  if (!live) task = ntask - 1;
  const unsigned o = task / ntile, tile = task - o * ntile;
  const unsigned x = tile * 128 + lane * 2;
  const bool on = live && x < inner;
  const size_t base = (size_t)o * n * inner + (on ? x : 0);
  const unsigned ng = n / G;
  const size_t rs = inner / 2;
  if (loader) {
    const d2* p = reinterpret_cast<const d2*>(in + base);
    auto issue = [&](unsigned grp) {
      if (!on) return;
      const unsigned slot = grp % (D + 1);
#pragma unroll
      for (int r = 0; r < G; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)(grp * G + r) * rs),
                                         (__attribute__((address_space(3))) void*)&ring[pair][slot][r][0], 16, 0, 0);
    };
    for (unsigned d = 0; d < (unsigned)D && d < ng; ++d) issue(d);
    for (unsigned g = 0; g < ng; ++g) {
      // group g has landed when at most (D - 1) * G younger loads are outstanding
      const unsigned younger = (ng - 1 - g < (unsigned)(D - 1)) ? ng - 1 - g : (unsigned)(D - 1);
      if (younger == D - 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (((D - 1) * G) & 0xF) | ((((D - 1) * G) >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);  // the tail: everything
      __builtin_amdgcn_s_barrier();
      if (g + D < ng) issue(g + D);
    }
  } else {
    d2* q = reinterpret_cast<d2*>(out + base);
    d2 acc = {0.0, 0.0};
    for (unsigned g = 0; g < ng; ++g) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      d2 w[G];
      const unsigned slot = g % (D + 1);
#pragma unroll
      for (int r = 0; r < G; ++r) w[r] = ring[pair][slot][r][lane];
#pragma unroll
      for (int r = 0; r < G; ++r) {
        acc += w[r];
        if (on) __builtin_nontemporal_store(acc, q + (size_t)(g * G + r) * rs);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // reads done before the next barrier frees the slot
    }
  }
}


// D: access-pattern test without long-lived waves: a flat launch whose wave-task is (row chunk c of R rows, level, x-tile);
// each wave loads its R rows, forms the prefix inside the chunk (NO carry between chunks: wrong sums, right traffic) and
// stores.  ORDER 0: chunk-major (c, level, tile) = the instantaneous footprint of the march (every level at the same
// rows); ORDER 1: level-major (level, c, tile) = each level swept contiguously.
template <int R, int ORDER>
__global__ __launch_bounds__(256) void k_chunks(const double* __restrict__ in, double* __restrict__ out, unsigned outer, unsigned n, unsigned inner, unsigned ntile, unsigned ntask) {
  const unsigned nb = gridDim.x, pb = nb >> 3;
  const unsigned b = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const unsigned task = b * 4 + (threadIdx.x >> 6);
  if (task >= ntask) return;
  const unsigned nc = n / R;
  unsigned c, o, tile;
  if (ORDER == 0) { c = task / (outer * ntile); const unsigned r = task - c * outer * ntile; o = r / ntile; tile = r - o * ntile; }
  else { o = task / (nc * ntile); const unsigned r = task - o * nc * ntile; c = r / ntile; tile = r - c * ntile; }
  const unsigned x = tile * 128 + (threadIdx.x & 63) * 2;
  if (x >= inner) return;
  const size_t base = ((size_t)o * n + (size_t)c * R) * inner + x;
  const d2* p = reinterpret_cast<const d2*>(in + base);
  d2* q = reinterpret_cast<d2*>(out + base);
  const size_t rs = inner / 2;
  d2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(p + r * rs);
  d2 acc = {0.0, 0.0};
#pragma unroll
  for (int r = 0; r < R; ++r) { acc += v[r]; __builtin_nontemporal_store(acc, q + r * rs); }
}


// E: the real thing -- a CHAINED flat launch.  Wave-task = (column = (level, x-tile), row chunk c of R rows).  An XCD owns a
// band of consecutive columns, cut into sub-bands of W columns; inside a sub-band the tasks run chunk-major (all W columns
// at chunk 0, then chunk 1, ...), so W chains advance side by side and the footprint is W / ntile levels per XCD.  A task
// loads its R rows at once, waits for the running sum that chunk c - 1 of its column published, adds its rows in sequence
// (same order as the march: bit-identical) and publishes its own last sum BEFORE issuing its R stores.  Tasks are handed
// out through a ticket per XCD, so every task a wave can wait for is already running.
// SYNC 0: value slots + epoch flag, agent-scope (sc1) stores: acknowledged by MEMORY, behind the XCD's write stream;
// SYNC 1: plain stores (acknowledged by the L2 the chain lives in), sc1 loads (bypass the L1, hit the L2);
// SYNC 2: self-validating 16-B slots {value, epoch} per lane value: one store, one polling load per link.
template <int R, int V, int SYNC>
__global__ __launch_bounds__(256) void k_chain(const double* __restrict__ in, double* __restrict__ out, unsigned outer, unsigned n, unsigned inner,
                                               unsigned ntile, unsigned nchunk, double* carry, unsigned* flag, unsigned* ticket, unsigned cpx, unsigned ncol, unsigned W,
                                               unsigned long long gen) {
  typedef double T __attribute__((ext_vector_type(V)));
  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  __shared__ unsigned s_ticket;
  const unsigned xcd = blockIdx.x & 7;
  if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[xcd], 1u);
  __syncthreads();
  const unsigned q = s_ticket * 4 + (threadIdx.x >> 6);
  const unsigned col_lo = xcd * cpx, col_hi = min(col_lo + cpx, ncol);
  if (col_lo >= col_hi) return;
  const unsigned ncols = col_hi - col_lo;
  if (q >= ncols * nchunk) return;
  unsigned j = q / (nchunk * W);
  const unsigned nsub = (ncols + W - 1) / W;
  if (j >= nsub) j = nsub - 1;
  const unsigned sub_lo = col_lo + j * W, w = min(W, col_hi - sub_lo);
  const unsigned ql = q - j * nchunk * W, c = ql / w, col = sub_lo + (ql - c * w);
  const unsigned o = col / ntile, tile = col - o * ntile;
  const unsigned lane = threadIdx.x & 63;
  const unsigned x = (tile * 64 + lane) * V;
  if (x >= inner) return;
  const size_t base = ((size_t)o * n + (size_t)c * R) * inner + x;
  const T* p = reinterpret_cast<const T*>(in + base);
  T* qo = reinterpret_cast<T*>(out + base);
  const size_t rs = inner / V;
  T v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(p + r * rs);
  T acc = {};
  unsigned* fl = flag + col;
  if (SYNC == 3 || SYNC == 4) {
    // slot of a value = {low half, epoch, high half, epoch}: valid at 8-B granularity, so a 16-B access that the memory
    // system splits in two cannot pair a new epoch with an old half
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4* slots = reinterpret_cast<u32x4*>(carry);
    if (c > 0) {
      const unsigned want = (unsigned)gen + c;
      const u32x4* cs = slots + ((size_t)o * 2 + ((c - 1) & 1)) * inner + x;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        u32x4 got;
        unsigned tries = 0;
        do {
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(got) : "v"(cs + k) : "memory");
        } while ((got[1] != want || got[3] != want) && ++tries < (1u << 18));
        acc[k] = __builtin_bit_cast(double, (u64)got[0] | ((u64)got[2] << 32));
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { acc += v[r]; v[r] = acc; }
    if (c + 1 < nchunk) {
      u32x4* cd = slots + ((size_t)o * 2 + (c & 1)) * inner + x;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const double ak = acc[k];  // (a scalar temporary: __builtin_bit_cast on a vector element reads element 0)
        const u64 bits = __builtin_bit_cast(u64, ak);
        const unsigned ep = (unsigned)gen + c + 1;
        u32x4 put = {(unsigned)bits, ep, (unsigned)(bits >> 32), ep};
        if (SYNC == 3) cd[k] = put;
        else {  // agent-scope stores: visible to every XCD, not only to the L2 this wave sits behind
          u64* c8 = reinterpret_cast<u64*>(cd + k);
          __hip_atomic_store(c8, (u64)put[0] | ((u64)ep << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(c8 + 1, (u64)put[2] | ((u64)ep << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
  } else if (SYNC == 2) {
    u64x2* slots = reinterpret_cast<u64x2*>(carry);  // [outer][2][inner] of {value bits, epoch}
    if (c > 0) {
      const u64 want = gen + c;
      const u64x2* cs = slots + ((size_t)o * 2 + ((c - 1) & 1)) * inner + x;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        u64x2 got;
        unsigned tries = 0;
        do {
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(got) : "v"(cs + k) : "memory");
        } while (got[1] != want && ++tries < (1u << 18));  // a bounded spin: a broken chain shows as a MISMATCH, not a hang
        acc[k] = __builtin_bit_cast(double, got[0]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { acc += v[r]; v[r] = acc; }
    if (c + 1 < nchunk) {
      u64x2* cd = slots + ((size_t)o * 2 + (c & 1)) * inner + x;
#pragma unroll
      for (int k = 0; k < V; ++k) {
        u64x2 put = {__builtin_bit_cast(u64, acc[k]), gen + c + 1};
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(cd + k), "v"(put) : "memory");
      }
    }
  } else {
  if (c > 0) {
    while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const double* cs = carry + ((size_t)o * 2 + ((c - 1) & 1)) * inner + x;
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = __hip_atomic_load(cs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) { acc += v[r]; v[r] = acc; }
  if (c + 1 < nchunk) {
    double* cd = carry + ((size_t)o * 2 + (c & 1)) * inner + x;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (SYNC == 0) __hip_atomic_store(cd + k, acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: through to memory
      else __hip_atomic_store(cd + k, acc[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);          // plain: acknowledged by the L2
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the carry has reached the L2
    if (lane == 0) {
      if (SYNC == 0) __hip_atomic_store(fl, c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(fl, c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) __builtin_nontemporal_store(v[r], qo + r * rs);
}

template <class F>
static float timeit(F&& launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const unsigned outer = 75, n = 2400, inner = 3600;
  const size_t cells = (size_t)outer * n * inner;
  double *in, *out, *ref;
  CK(hipMalloc(&in, cells * 8)); CK(hipMalloc(&out, cells * 8)); CK(hipMalloc(&ref, cells * 8));
  hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, in, cells);
  double* ws_carry; unsigned* ws_flag;
  CK(hipMalloc(&ws_carry, (size_t)outer * 2 * inner * 16)); CK(hipMemset(ws_carry, 0, (size_t)outer * 2 * inner * 16)); unsigned long long gen = 0; CK(hipMalloc(&ws_flag, (size_t)outer * 64 * 4 + 64));
  CK(hipDeviceSynchronize());
  auto rep = [&](const char* name, float ms, const char* chk) { printf("%-64s %8.4f ms  %.3f of 8 TB/s  %s\n", name, ms, 16.0 * cells / ms / 1e6 / 8000, chk); fflush(stdout); };
  std::vector<double> hr(3600 * 4), ho(3600 * 4);
  auto check = [&]() -> const char* {  // bitwise against variant A on the last rows of the first / last level
    const size_t offs[2] = {(size_t)(n - 2) * inner, cells - 2 * (size_t)inner};
    for (size_t off : offs) {
      CK(hipMemcpy(hr.data(), ref + off, 2 * inner * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ho.data(), out + off, 2 * inner * 8, hipMemcpyDeviceToHost));
      if (memcmp(hr.data(), ho.data(), 2 * inner * 8)) return "MISMATCH";
    }
    return "bits ok";
  };
  printf("# E sync: 0 = slots + flag, agent-scope stores; 1 = slots + flag, plain stores + sc1 loads; 3 = self-validating 16-B slots, plain store;\n#         4 = as 3 with agent-scope stores (SYNC 2, a 16-B {value, epoch} slot written from inline asm, was dropped)\n");
  for (int rnd = 0; rnd < 2; ++rnd) {
    {
      const unsigned ntile = (inner + 63) / 64, ntask = outer * ntile, grid = (((ntask + 3) / 4 + 7) / 8) * 8;
#define SINGLE(U) { float ms = timeit([&] { hipLaunchKernelGGL((k_single<U>), dim3(grid), dim3(256), 0, 0, in, rnd == 0 && U == 32 ? ref : out, outer, n, inner, ntile, ntask); }, 7); rep("A single wave, window " #U, ms, ""); }
      SINGLE(32) SINGLE(16)
    }
    {
      const unsigned ntile = (inner + 127) / 128, ntask = outer * ntile;
#define SPLIT(G, D, P) { const unsigned grid = (((ntask + P - 1) / P + 7) / 8) * 8; CK(hipMemsetAsync(out, 0, cells * 8)); \
      float ms = timeit([&] { hipLaunchKernelGGL((k_split<G, D, P>), dim3(grid), dim3(128 * P), 0, 0, in, out, outer, n, inner, ntile, ntask); }, 7); rep("B loader/storer via LDS, G=" #G " D=" #D " pairs/wg=" #P, ms, check()); }
      SPLIT(8, 2, 1) SPLIT(8, 3, 1)
#define SPLITL(G, D, P) { const unsigned grid = (((ntask + P - 1) / P + 7) / 8) * 8; CK(hipMemsetAsync(out, 0, cells * 8)); \
      float ms = timeit([&] { hipLaunchKernelGGL((k_split_lds<G, D, P>), dim3(grid), dim3(128 * P), 0, 0, in, out, outer, n, inner, ntile, ntask); }, 7); rep("C loader -> LDS direct, G=" #G " D=" #D " pairs/wg=" #P, ms, check()); }
      SPLITL(4, 3, 1) SPLITL(2, 7, 1)
#define CHUNKS(R, ORDER) { const unsigned nt = outer * (n / R) * ntile, grid = (((nt + 3) / 4 + 7) / 8) * 8; \
      float ms = timeit([&] { hipLaunchKernelGGL((k_chunks<R, ORDER>), dim3(grid), dim3(256), 0, 0, in, out, outer, n, inner, ntile, nt); }, 7); rep("D flat chunks (no carry), R=" #R " order=" #ORDER, ms, ""); }
      CHUNKS(16, 0) CHUNKS(16, 1)
    }
    {
#define CHAIN(R, V, SYNC, LV) { const unsigned ntile = (inner + 64 * V - 1) / (64 * V), ncol = outer * ntile, cpx = (ncol + 7) / 8, nchunk = n / R, W = LV * ntile; \
      const unsigned grid = 8 * ((cpx * nchunk + 3) / 4); CK(hipMemsetAsync(out, 0, cells * 8)); \
      float ms = timeit([&] { CK(hipMemsetAsync(ws_flag, 0, (size_t)outer * 64 * 4 + 64)); gen += 1ull << 20; \
        hipLaunchKernelGGL((k_chain<R, V, SYNC>), dim3(grid), dim3(256), 0, 0, in, out, outer, n, inner, ntile, nchunk, ws_carry, ws_flag + 16, ws_flag, cpx, ncol, W, gen); }, 7); \
      rep("E chained flat chunks, R=" #R " doubles/lane=" #V " sync=" #SYNC " levels side by side=" #LV, ms, check()); }
      CHAIN(32, 1, 0, 1) CHAIN(32, 1, 1, 1) CHAIN(32, 1, 3, 1) CHAIN(32, 2, 3, 1) CHAIN(16, 1, 3, 2) CHAIN(40, 1, 3, 1) CHAIN(32, 1, 3, 2) CHAIN(32, 1, 4, 1)
    }
  }
  return 0;
}
