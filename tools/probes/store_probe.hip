// store_probe.hip -- the KERNEL-side variables of the Z scan's store pattern (VERDICT r05 "next round" 2).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_probe.hip -o tools/probes/store_probe && tools/probes/store_probe [P ny nx]
// The scan along Z of (Z, Y, X) writes P = 75 planes 69 MB apart AT ONCE; round 5 varied where the output lives (plain /
// scattered physical memory) and found one time per box, 1.65 - 2.02 ms.  This probe varies what the KERNEL does instead:
//   NT   bytes one wave writes to one plane in one go: NT x (64 lanes x LB bytes), LB = 8 or 16  -> 512 B ... 8 KB
//        (a workgroup = 4 adjacent waves: 2 KB ... 32 KB contiguous per plane)
//   G    planes one task walks (the rest of the column belongs to later tasks: a scan needs a carry between them, the
//        chain of k_cumsum_chain; here only the MEMORY pattern is probed): 75 = the march, 1 = a flat sweep per plane
//   W    tasks of one plane group that run side by side before the next group of the same columns starts: the number of
//        planes an XCD has open at a time is ~G, each being written W x NT KB wide
//   band XCD-contiguous wave order (the library's) or plain dispatch order (every XCD interleaved over the same range)
// Modes: WO = stores only, RW = load + add + store (the scan's traffic), RO = loads only (a sum into one value per lane).
// Buffers: plain hipMalloc and "scattered" (separately created 64 MiB physical allocations behind one virtual range,
// the library's xg_scatter_alloc).  Output: one JSON line per (variant, mode, buffers) with median / min ms of 7 launches
// and the fraction of 8 TB/s on the bytes moved; first the flat fill / copy ceilings and the library-shaped march.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef double dv __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Plan {
  int P, G;            // planes, planes per task
  u32 ngroup;          // ceil(P / G)
  u32 nsuper;          // super-tiles (NT tiles of 64 lanes) per plane
  u32 W;               // super-tiles of one sub-band
  u32 band;            // 1: XCD-contiguous order
  int64_t plane;       // elements (of the lane type) per plane
};

__device__ __forceinline__ u64 task_id(u32 band) {
  const u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (!band) return (u64)blockIdx.x * 4 + w;
  const u32 pb = (gridDim.x + 7) >> 3;
  return (u64)((blockIdx.x & 7) * pb + (blockIdx.x >> 3)) * 4 + w;
}

// task order: XCD band (by the launch) -> sub-band of W super-tiles -> plane group -> super-tile of the sub-band
__device__ __forceinline__ bool decode(const Plan& p, u64 t, u32& gi, u32& st) {
  const u64 per_sub = (u64)p.W * p.ngroup;
  const u32 sub = (u32)(t / per_sub);
  const u32 sub_lo = sub * p.W;
  if (sub_lo >= p.nsuper) return false;
  const u32 w = (p.nsuper - sub_lo < p.W) ? p.nsuper - sub_lo : p.W;
  const u64 r = t - (u64)sub * per_sub;
  gi = (u32)(r / w);
  if (gi >= p.ngroup) return false;
  st = sub_lo + (u32)(r - (u64)gi * w);
  return true;
}

// T = double (8-B lanes) or dv (16-B lanes); MODE 0 = WO, 1 = RW, 2 = RO
template <typename T, int NT, int MODE, int ZUX = 0>
__global__ __launch_bounds__(256) void k_pattern(const T* __restrict__ in, T* __restrict__ out, Plan p) {
  u32 gi, st;
  if (!decode(p, task_id(p.band), gi, st)) return;
  const int lane = threadIdx.x & 63;
  const int64_t x0 = ((int64_t)st * NT) * 64 + lane;
  constexpr int ZU = ZUX ? ZUX : (NT >= 4 ? 1 : 4 / NT);  // >= 4 loads in flight per lane, as in the library's march (ZUX: explicit)
  T acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = T(1.0 + lane);
  const int z0 = gi * p.G, z1 = (z0 + p.G < p.P) ? z0 + p.G : p.P;
  bool ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ok[t] = x0 + t * 64 < p.plane;
  int z = z0;
  for (; z + ZU <= z1; z += ZU) {
    T v[ZU][NT];
    if (MODE != 0) {
#pragma unroll
      for (int u = 0; u < ZU; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (ok[t]) v[u][t] = __builtin_nontemporal_load(in + (int64_t)(z + u) * p.plane + x0 + t * 64);
    }
#pragma unroll
    for (int u = 0; u < ZU; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (MODE != 0) acc[t] += v[u][t]; else acc[t] += T(1.0);
        if (MODE != 2 && ok[t]) __builtin_nontemporal_store(acc[t], out + (int64_t)(z + u) * p.plane + x0 + t * 64);
      }
  }
  for (; z < z1; ++z)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (MODE != 0) { if (ok[t]) acc[t] += __builtin_nontemporal_load(in + (int64_t)z * p.plane + x0 + t * 64); } else acc[t] += T(1.0);
      if (MODE != 2 && ok[t]) __builtin_nontemporal_store(acc[t], out + (int64_t)z * p.plane + x0 + t * 64);
    }
  if (MODE == 2) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (ok[t]) __builtin_nontemporal_store(acc[t], out + (int64_t)z0 * p.plane + x0 + t * 64);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_flat(const dv* __restrict__ in, dv* __restrict__ out, int64_t nvec) {
  const u32 pb = (gridDim.x + 7) >> 3;
  const int64_t i = (int64_t)((blockIdx.x & 7) * pb + (blockIdx.x >> 3)) * 256 + threadIdx.x;
  if (i >= nvec) return;
  dv v = {1.0, 2.0};
  if (MODE != 0) v = __builtin_nontemporal_load(in + i);
  __builtin_nontemporal_store(v, out + i);
}

// ---- buffers ----
// (defined before the level-chain helpers below use them)
struct Buf { void* p; size_t bytes; bool scattered; std::vector<hipMemGenericAllocationHandle_t> h; };
static Buf make(size_t bytes, bool scattered) {
  Buf b; b.bytes = bytes; b.scattered = scattered; b.p = nullptr;
  if (!scattered) { CK(hipMalloc(&b.p, bytes)); return b; }
  hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  const size_t chunk = 64ull << 20;
  const size_t n = (bytes + chunk - 1) / chunk;
  CK(hipMemAddressReserve(&b.p, n * chunk, 0, nullptr, 0));
  hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    CK(hipMemMap((char*)b.p + i * chunk, chunk, 0, h, 0));
    b.h.push_back(h);
  }
  CK(hipMemSetAccess(b.p, n * chunk, &acc, 1));
  return b;
}

template <typename F>
static void timeit(F launch, double* med, double* mn) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch(); CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  *med = ts[ts.size() / 2]; *mn = ts[0];
}


// ---------------------------------------------------------------------------------------------------------------------
// The level-major scan the pattern rows above ask for, as a REAL scan (checked bit for bit against the march):
// task = (sub-band of W x-tiles, chunk c of R levels, tile), chunk-major inside the sub-band, handed out by per-XCD
// tickets (a task's predecessor always holds an earlier ticket: it is running or done).  The carry of chunk c is the
// LAST OUTPUT ROW of chunk c - 1 itself: stored with a plain store (stays in the XCD's L2), then `s_waitcnt vmcnt(0)`
// (the L2 has it), then a 4-byte flag per tile; the successor polls the flag (sc1: past the L1) and reads the row back
// from the L2.  No 16-byte slots per lane, no extra HBM bytes beyond the flags.  Bounded spin: a starved task writes NaN.
// ---------------------------------------------------------------------------------------------------------------------
template <int R, int BS>
__global__ __launch_bounds__(BS) void k_levelchain(const dv* __restrict__ in, dv* out, int P, int64_t plane, u32 ntile, u32 W, u32 nblk_xcd,
                                                   u32* ticket, u32* flag, u32 base, u32 spin, u32* gave_up) {
  __shared__ u32 s_t;
  constexpr u32 WPB = BS / 64;
  const u32 xcd = blockIdx.x & 7;
  if (threadIdx.x == 0) {
    const u32 t = atomicAdd(&ticket[xcd * 32], 1u);
    if (t == nblk_xcd - 1) __hip_atomic_store(&ticket[xcd * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_t = t;
  }
  __syncthreads();
  const u32 q = __builtin_amdgcn_readfirstlane(s_t * WPB + (threadIdx.x >> 6));
  const u32 cpx = (ntile + 7) / 8, col_lo = xcd * cpx;
  if (col_lo >= ntile) return;
  const u32 col_hi = (ntile - col_lo < cpx) ? ntile : col_lo + cpx, ncols = col_hi - col_lo;
  const u32 nchunk = (P + R - 1) / R;
  if (q >= ncols * nchunk) return;
  u32 j = q / (nchunk * W);
  const u32 nsub = (ncols + W - 1) / W;
  if (j >= nsub) j = nsub - 1;
  const u32 sub_lo = col_lo + j * W;
  const u32 w = (col_hi - sub_lo < W) ? col_hi - sub_lo : W;
  const u32 ql = q - j * nchunk * W;
  const u32 c = __builtin_amdgcn_readfirstlane(ql / w);
  const u32 tile = sub_lo + (ql - c * w);
  const int lane = threadIdx.x & 63;
  const int64_t x = (int64_t)tile * 64 + lane;
  if (x >= plane) return;  // (whole tiles in the probe's shapes: no flag is left unpublished)
  const int z0 = c * R, rows = (P - z0 < R) ? P - z0 : R;
  const dv* pin = in + (int64_t)z0 * plane + x;
  dv* pout = out + (int64_t)z0 * plane + x;
  dv v[R];
  if (rows == R) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = __builtin_nontemporal_load(pin + (int64_t)r * plane);
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) { v[r] = dv{0.0, 0.0}; if (r < rows) v[r] = __builtin_nontemporal_load(pin + (int64_t)r * plane); }
  }
  dv acc = {0.0, 0.0};
  if (c > 0) {
    const u32 want = base + c;
    u32 got, tries = 0;
    const u32* fp = flag + tile;
    do {
      asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(got) : "v"(fp) : "memory");
    } while (got != want && ++tries < spin);
    if (got != want) {
      if (lane == 0) atomicAdd(gave_up, 1u);
      acc = dv{__builtin_nan(""), __builtin_nan("")};
    } else {
      const dv* cp = pout - plane;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=&v"(acc) : "v"(cp) : "memory");
    }
  }
  const bool more = c + 1 < nchunk;
  if (rows == R) {
#pragma unroll
    for (int r = 0; r < R; ++r) { acc = (c > 0 || r > 0) ? acc + v[r] : v[r]; v[r] = acc; }
#pragma unroll
    for (int r = 0; r < R - 1; ++r) __builtin_nontemporal_store(v[r], pout + (int64_t)r * plane);
    if (more) pout[(int64_t)(R - 1) * plane] = v[R - 1];  // the carry row: a plain store, kept by the L2
    else __builtin_nontemporal_store(v[R - 1], pout + (int64_t)(R - 1) * plane);
  } else {  // the ragged last chunk of a column
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (r < rows) { acc = (c > 0 || r > 0) ? acc + v[r] : v[r]; __builtin_nontemporal_store(acc, pout + (int64_t)r * plane); }
  }
  if (more) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(flag + tile, base + c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void k_refscan(const dv* __restrict__ in, dv* __restrict__ out, int P, int64_t plane) {
  const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (x >= plane) return;
  dv acc = in[x];
  out[x] = acc;
  for (int z = 1; z < P; ++z) { acc += in[(int64_t)z * plane + x]; out[(int64_t)z * plane + x] = acc; }
}
__global__ void k_fillin(double* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    u64 z = (u64)i * 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
    p[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}
__global__ void k_diffcount(const u64* a, const u64* b, int64_t n, u64* cnt) {
  u64 local = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) local += (a[i] != b[i]);
  if (local) atomicAdd(cnt, local);
}

template <int R, int BS>
static void run_levelchain(const Buf& in, const Buf& out, const Buf& ref, int P, int64_t plane_bytes, u32 W, u32* ticket, u32* flag, u32* gave_up, u64* cnt, u32& base) {
  const int64_t plane = plane_bytes / 16;
  const u32 ntile = (u32)((plane + 63) / 64), cpx = (ntile + 7) / 8, nchunk = (P + R - 1) / R;
  const u32 nblk_xcd = (cpx * nchunk + BS / 64 - 1) / (BS / 64);
  if (W == 0 || W > cpx) W = cpx;
  CK(hipMemset(out.p, 0xff, out.bytes));
  CK(hipMemset(gave_up, 0, 4));
  double med, mn;
  timeit([&] {
    hipLaunchKernelGGL((k_levelchain<R, BS>), dim3(nblk_xcd * 8), dim3(BS), 0, 0, (const dv*)in.p, (dv*)out.p, P, plane, ntile, W, nblk_xcd, ticket, flag, base, 50000u, gave_up);
    base += nchunk + 1;
  }, &med, &mn);
  CK(hipGetLastError());
  CK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(k_diffcount, dim3(4096), dim3(256), 0, 0, (const u64*)out.p, (const u64*)ref.p, (int64_t)(out.bytes / 8), cnt);
  u64 bad; u32 gu;
  CK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
  printf("{\"variant\": \"levelchain\", \"R\": %d, \"BS\": %d, \"W\": %u, \"buffers\": \"%s\", \"mode\": \"RW\", \"ms\": %.4f, \"min_ms\": %.4f, \"frac\": %.4f, \"mismatches\": %llu, \"gave_up\": %u}\n",
         R, BS, W, out.scattered ? "scattered" : "plain", med, mn, 2.0 * out.bytes / (med * 1e-3) / 8e12, bad, gu);
  fflush(stdout);
  if (gu) { printf("{\"abort\": \"tasks gave up: the hand-off does not work as assumed\"}\n"); exit(2); }
}

static const char* MODES[3] = {"WO", "RW", "RO"};
static void report(const char* name, int lb, int nt, int G, u32 W, int band, int mode, bool scat, double med, double mn, double bytes) {
  printf("{\"variant\": \"%s\", \"lane_bytes\": %d, \"NT\": %d, \"wave_burst_B\": %d, \"G\": %d, \"W\": %u, \"band\": %d, \"mode\": \"%s\", \"buffers\": \"%s\", "
         "\"ms\": %.4f, \"min_ms\": %.4f, \"frac\": %.4f}\n", name, lb, nt, nt * 64 * lb, G, W, band, MODES[mode], scat ? "scattered" : "plain", med, mn,
         bytes / (med * 1e-3) / 8e12);
  fflush(stdout);
}

template <typename T, int NT>
static void run_pattern(const char* name, const Buf& in, const Buf& out, int P, int64_t plane_bytes, int G, u32 Wreq, int band, int modes_mask) {
  Plan p; p.P = P; p.G = G; p.ngroup = (P + G - 1) / G; p.plane = plane_bytes / (int64_t)sizeof(T); p.band = band;
  p.nsuper = (u32)((p.plane + (int64_t)NT * 64 - 1) / ((int64_t)NT * 64));
  // (a sub-band never crosses an XCD band: W = 0 means the whole band)
  const u32 per_xcd = (p.nsuper + 7) / 8;
  p.W = (Wreq == 0 || Wreq > per_xcd) ? per_xcd : Wreq;
  const u64 nsub = (p.nsuper + p.W - 1) / p.W;
  const u64 ntask = nsub * p.W * p.ngroup;
  const u32 nblk = (u32)((((ntask + 3) / 4) + 7) / 8 * 8);
  for (int mode = 0; mode < 3; ++mode) {
    if (!(modes_mask & (1 << mode))) continue;
    double med, mn;
    timeit([&] {
      if (mode == 0) hipLaunchKernelGGL((k_pattern<T, NT, 0>), dim3(nblk), dim3(256), 0, 0, (const T*)in.p, (T*)out.p, p);
      if (mode == 1) hipLaunchKernelGGL((k_pattern<T, NT, 1>), dim3(nblk), dim3(256), 0, 0, (const T*)in.p, (T*)out.p, p);
      if (mode == 2) hipLaunchKernelGGL((k_pattern<T, NT, 2>), dim3(nblk), dim3(256), 0, 0, (const T*)in.p, (T*)out.p, p);
    }, &med, &mn);
    CK(hipGetLastError());
    const double bytes = (double)plane_bytes * P * (mode == 1 ? 2.0 : 1.0);
    report(name, (int)sizeof(T), NT, G, p.W, band, mode, out.scattered, med, mn, bytes);
  }
}

// the march with ZU loads in flight per lane before the ZU stores (latency probe: the counters say the march is not back-pressured)
template <int ZU>
static void run_depth(const Buf& in, const Buf& out, int P, int64_t plane_bytes) {
  Plan p; p.P = P; p.G = P; p.ngroup = 1; p.plane = plane_bytes / 16; p.band = 1;
  p.nsuper = (u32)((p.plane + 63) / 64);
  p.W = (p.nsuper + 7) / 8;
  const u64 ntask = (u64)((p.nsuper + p.W - 1) / p.W) * p.W;
  const u32 nblk = (u32)((((ntask + 3) / 4) + 7) / 8 * 8);
  double med, mn;
  timeit([&] { hipLaunchKernelGGL((k_pattern<dv, 1, 1, ZU>), dim3(nblk), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, p); }, &med, &mn);
  printf("{\"variant\": \"march depth\", \"loads_in_flight\": %d, \"buffers\": \"%s\", \"mode\": \"RW\", \"ms\": %.4f, \"min_ms\": %.4f, \"frac\": %.4f}\n",
         ZU, out.scattered ? "scattered" : "plain", med, mn, 2.0 * out.bytes / (med * 1e-3) / 8e12);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 75;
  const int64_t ny = argc > 2 ? atoll(argv[2]) : 2400, nx = argc > 3 ? atoll(argv[3]) : 3600;
  const int64_t plane_bytes = ny * nx * 8;
  const size_t bytes = (size_t)plane_bytes * P;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"P\": %d, \"plane_bytes\": %lld, \"total_GB\": %.3f}\n", prop.gcnArchName, P, (long long)plane_bytes, bytes / 1e9);
  const char* what = argc > 4 ? argv[4] : "all";  // all | pattern | chain | pmc
  const bool pmc = strcmp(what, "pmc") == 0;
  const bool depth = strcmp(what, "depth") == 0;
  const bool do_pattern = strcmp(what, "chain") != 0 && !pmc && !depth, do_chain = strcmp(what, "pattern") != 0 && !pmc && !depth;
  Buf in = make(bytes, false);
  hipLaunchKernelGGL(k_fillin, dim3(8192), dim3(256), 0, 0, (double*)in.p, (int64_t)(bytes / 8));
  Buf ref = make(bytes, false);
  hipLaunchKernelGGL(k_refscan, dim3((u32)((plane_bytes / 16 + 255) / 256)), dim3(256), 0, 0, (const dv*)in.p, (dv*)ref.p, P, plane_bytes / 16);
  CK(hipDeviceSynchronize());
  u32 *ticket, *flag, *gave_up; u64* cnt; u32 base = 1;
  CK(hipMalloc(&ticket, 8 * 32 * 4)); CK(hipMemset(ticket, 0, 8 * 32 * 4));
  CK(hipMalloc(&flag, (size_t)(plane_bytes / 16 / 64 + 64) * 4)); CK(hipMemset(flag, 0, (size_t)(plane_bytes / 16 / 64 + 64) * 4));
  CK(hipMalloc(&gave_up, 4)); CK(hipMalloc(&cnt, 8));
  if (strcmp(what, "anatomy") == 0) {
    // What makes a scattered buffer bad -- the SET of physical chunks or their ORDER?  NB buffers graded; the worst one's
    // chunks are then mapped again in other orders (and mixed with the best one's) into fresh address ranges and graded
    const int NB = argc > 5 ? atoi(argv[5]) : 8;
    const size_t chunk = 64ull << 20;
    const size_t nchunk = (bytes + chunk - 1) / chunk;
    std::vector<Buf> bufs;
    for (int b = 0; b < NB; ++b) bufs.push_back(make(bytes, true));
    auto planes_ms = [&](void* base, int p0, int p1) {
      Plan p; p.P = p1 - p0; p.G = p.P; p.ngroup = 1; p.plane = plane_bytes / 16; p.band = 1;
      p.nsuper = (u32)((p.plane + 63) / 64); p.W = (p.nsuper + 7) / 8;
      const u64 ntask = (u64)((p.nsuper + p.W - 1) / p.W) * p.W;
      const u32 nblk = (u32)((((ntask + 3) / 4) + 7) / 8 * 8);
      double med, mn;
      dv* o = (dv*)((char*)base + (size_t)p0 * plane_bytes);
      timeit([&] { hipLaunchKernelGGL((k_pattern<dv, 1, 0>), dim3(nblk), dim3(256), 0, 0, (const dv*)in.p, o, p); }, &med, &mn);
      return med;
    };
    std::vector<double> g(NB);
    int worst = 0, best = 0;
    for (int b = 0; b < NB; ++b) {
      g[b] = planes_ms(bufs[b].p, 0, P);
      if (g[b] > g[worst]) worst = b;
      if (g[b] < g[best]) best = b;
      printf("{\"variant\": \"anatomy\", \"buffer\": %d, \"planes_fill_ms\": %.4f}\n", b, g[b]);
    }
    for (int b : {worst, best}) {
      printf("{\"variant\": \"anatomy\", \"buffer\": %d, \"which\": \"%s\", \"first_half_ms\": %.4f, \"second_half_ms\": %.4f, \"q1\": %.4f, \"q2\": %.4f, \"q3\": %.4f, \"q4\": %.4f}\n",
             b, b == worst ? "worst" : "best", planes_ms(bufs[b].p, 0, P / 2), planes_ms(bufs[b].p, P / 2, P), planes_ms(bufs[b].p, 0, P / 4),
             planes_ms(bufs[b].p, P / 4, P / 2), planes_ms(bufs[b].p, P / 2, 3 * P / 4), planes_ms(bufs[b].p, 3 * P / 4, P));
    }
    fflush(stdout);
    // remap: the same physical chunks behind a fresh address range in another order
    auto remap_grade = [&](const char* name, const std::vector<hipMemGenericAllocationHandle_t>& hs) {
      void* va = nullptr;
      CK(hipMemAddressReserve(&va, hs.size() * chunk, 0, nullptr, 0));
      for (size_t i = 0; i < hs.size(); ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, hs[i], 0));
      hipMemAccessDesc acc; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, hs.size() * chunk, &acc, 1));
      const double ms = planes_ms(va, 0, P);
      printf("{\"variant\": \"anatomy\", \"remap\": \"%s\", \"planes_fill_ms\": %.4f}\n", name, ms);
      fflush(stdout);
      CK(hipDeviceSynchronize());
      for (size_t i = 0; i < hs.size(); ++i) CK(hipMemUnmap((char*)va + i * chunk, chunk));
      // (the range is not given back: a reused range served stale pages in the library's pool)
    };
    const auto& hw = bufs[worst].h; const auto& hb = bufs[best].h;
    remap_grade("worst, same order", hw);
    remap_grade("best, same order", hb);
    { auto v = hw; std::reverse(v.begin(), v.end()); remap_grade("worst, reversed", v); }
    { std::vector<hipMemGenericAllocationHandle_t> v; for (size_t i = 0; i < nchunk; i += 2) v.push_back(hw[i]); for (size_t i = 1; i < nchunk; i += 2) v.push_back(hw[i]); remap_grade("worst, evens then odds", v); }
    for (int seed = 1; seed <= 3; ++seed) {
      auto v = hw; u64 st = 88172645463325252ull * seed;
      for (size_t i = v.size() - 1; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; std::swap(v[i], v[st % (i + 1)]); }
      remap_grade("worst, shuffled", v);
    }
    { std::vector<hipMemGenericAllocationHandle_t> v; for (size_t i = 0; i < nchunk; ++i) v.push_back((i & 1) ? hw[i] : hb[i]); remap_grade("even chunks of the best, odd of the worst", v); }
    { std::vector<hipMemGenericAllocationHandle_t> v; for (size_t i = 0; i < nchunk; ++i) v.push_back(i < nchunk / 2 ? hb[i] : hw[i]); remap_grade("first half best, second half worst", v); }
    { auto v = hb; std::reverse(v.begin(), v.end()); remap_grade("best, reversed", v); }
    return 0;
  }
  if (strcmp(what, "grade") == 0) {
    // Is a slow placement a property of the PROCESS or of the BUFFER?  NB scattered buffers alive at once; for each: the flat fill,
    // the many-plane fill (the candidate GRADING kernel: write-only, no input needed) and the scan-shaped march it should predict
    const int NB = argc > 5 ? atoi(argv[5]) : 10;
    std::vector<Buf> bufs;
    for (int b = 0; b < NB; ++b) bufs.push_back(make(bytes, true));
    const int64_t nvec = (int64_t)(bytes / 16);
    const u32 nb = (u32)(((nvec + 255) / 256 + 7) / 8 * 8);
    for (int pass = 0; pass < 2; ++pass)
      for (int b = 0; b < NB; ++b) {
        Buf& out = bufs[b];
        double f_med, f_mn, w_med, w_mn, r_med, r_mn;
        timeit([&] { hipLaunchKernelGGL((k_flat<0>), dim3(nb), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, nvec); }, &f_med, &f_mn);
        Plan p; p.P = P; p.G = P; p.ngroup = 1; p.plane = plane_bytes / 16; p.band = 1;
        p.nsuper = (u32)((p.plane + 63) / 64); p.W = (p.nsuper + 7) / 8;
        const u64 ntask = (u64)((p.nsuper + p.W - 1) / p.W) * p.W;
        const u32 nblk = (u32)((((ntask + 3) / 4) + 7) / 8 * 8);
        timeit([&] { hipLaunchKernelGGL((k_pattern<dv, 1, 0>), dim3(nblk), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, p); }, &w_med, &w_mn);
        timeit([&] { hipLaunchKernelGGL((k_pattern<dv, 1, 1>), dim3(nblk), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, p); }, &r_med, &r_mn);
        printf("{\"variant\": \"grade\", \"pass\": %d, \"buffer\": %d, \"flat_fill_ms\": %.4f, \"planes_fill_ms\": %.4f, \"planes_over_flat\": %.3f, \"march_rw_ms\": %.4f}\n",
               pass, b, f_med, w_med, w_med / f_med, r_med);
        fflush(stdout);
      }
    return 0;
  }
  for (int scat = 0; scat < 2; ++scat) {
    Buf out = make(bytes, scat != 0);
    if (pmc) {
      // a fixed sequence for tools/probes/store_probe_pmc.py: every line below = 9 dispatches (2 warm-ups + 7 timed) of ONE kernel,
      // in this order, on this buffer kind; the counters of the last 3 dispatches of each group are averaged
      const int64_t nvec = (int64_t)(bytes / 16);
      const u32 nb = (u32)(((nvec + 255) / 256 + 7) / 8 * 8);
      double med, mn;
      timeit([&] { hipLaunchKernelGGL((k_flat<1>), dim3(nb), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, nvec); }, &med, &mn);
      report("flat copy", 16, 1, 1, 0, 1, 1, scat, med, mn, 2.0 * bytes);
      run_pattern<dv, 1>("march", in, out, P, plane_bytes, P, 0, 1, 2);
      run_pattern<dv, 1>("grouped", in, out, P, plane_bytes, 8, 1024, 1, 2);
      run_pattern<dv, 1>("grouped", in, out, P, plane_bytes, 4, 1024, 1, 2);
      run_pattern<dv, 1>("grouped", in, out, P, plane_bytes, 1, 1024, 1, 2);
      run_levelchain<5, 1024>(in, out, ref, P, plane_bytes, 0, ticket, flag, gave_up, cnt, base);
      continue;
    }
    if (strcmp(what, "depth") == 0) {
      run_depth<1>(in, out, P, plane_bytes); run_depth<2>(in, out, P, plane_bytes); run_depth<4>(in, out, P, plane_bytes);
      run_depth<8>(in, out, P, plane_bytes); run_depth<15>(in, out, P, plane_bytes); run_depth<25>(in, out, P, plane_bytes);
      continue;
    }
    if (do_chain) {
      run_pattern<dv, 1>("march", in, out, P, plane_bytes, P, 0, 1, 2);
      const u32 Wc[] = {1024, 4096, 0};
      for (u32 W : Wc) {
#define LC(R_, BS_) run_levelchain<R_, BS_>(in, out, ref, P, plane_bytes, W, ticket, flag, gave_up, cnt, base)
        LC(2, 256); LC(2, 1024); LC(3, 256); LC(3, 1024); LC(4, 256); LC(4, 1024); LC(5, 256); LC(5, 1024); LC(8, 256); LC(8, 1024); LC(15, 256); LC(15, 1024);
#undef LC
      }
    }
    if (!do_pattern) continue;
    const int64_t nvec = (int64_t)(bytes / 16);
    const u32 nb = (u32)(((nvec + 255) / 256 + 7) / 8 * 8);
    double med, mn;
    timeit([&] { hipLaunchKernelGGL((k_flat<0>), dim3(nb), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, nvec); }, &med, &mn);
    report("flat fill", 16, 1, 1, 0, 1, 0, scat, med, mn, (double)bytes);
    timeit([&] { hipLaunchKernelGGL((k_flat<1>), dim3(nb), dim3(256), 0, 0, (const dv*)in.p, (dv*)out.p, nvec); }, &med, &mn);
    report("flat copy", 16, 1, 1, 0, 1, 1, scat, med, mn, 2.0 * bytes);
    // (a) the march (all planes open), burst per wave and plane 512 B ... 8 KB
    run_pattern<double, 1>("march", in, out, P, plane_bytes, P, 0, 1, 7);
    run_pattern<dv, 1>("march", in, out, P, plane_bytes, P, 0, 1, 7);
    run_pattern<dv, 2>("march", in, out, P, plane_bytes, P, 0, 1, 3);
    run_pattern<dv, 4>("march", in, out, P, plane_bytes, P, 0, 1, 3);
    run_pattern<dv, 8>("march", in, out, P, plane_bytes, P, 0, 1, 3);
    run_pattern<dv, 1>("march unbanded", in, out, P, plane_bytes, P, 0, 0, 3);
    // (b) planes open at a time: groups of G planes, sub-bands of W super-tiles
    const int Gs[] = {38, 25, 15, 8, 4, 1};
    const u32 Ws[] = {64, 1024, 0};
    for (int G : Gs)
      for (u32 W : Ws) {
        run_pattern<dv, 1>("grouped", in, out, P, plane_bytes, G, W, 1, 3);
        if (G == 25 || G == 8) run_pattern<dv, 4>("grouped", in, out, P, plane_bytes, G, W / 4, 1, 3);
      }
    run_pattern<dv, 1>("grouped unbanded", in, out, P, plane_bytes, 25, 0, 0, 3);
    run_pattern<dv, 1>("grouped unbanded", in, out, P, plane_bytes, 8, 0, 0, 3);
    run_pattern<dv, 1>("grouped unbanded", in, out, P, plane_bytes, 1, 0, 0, 3);
  }
  return 0;
}
