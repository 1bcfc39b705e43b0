// levels_probe.hip -- the level-major Z scan (K5L, round 3) as a stand-alone probe: does ANY shape of it beat the march?
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probes/levels_probe.hip -o tools/levels_probe && tools/levels_probe
// One generation of waves; a wave keeps the running sums of NT consecutive x-tiles (64 lanes x 16 B) in registers and
// sweeps the levels once; rows through buffer descriptors; straight-line levels; loads run D - 1 groups of G tiles ahead.
// Variants <NT, G, D>: registers = 4 NT (sums) + 4 G D (buffers) + ~25.  Every variant is checked bit for bit against the
// march.  Results go to stdout (profiles/history/r03p_levels_probe_*.txt).
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
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int64_t uni(int64_t v) {
  const u32 lo = __builtin_amdgcn_readfirstlane((u32)(u64)v), hi = __builtin_amdgcn_readfirstlane((u32)((u64)v >> 32));
  return (int64_t)((u64)lo | ((u64)hi << 32));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const double* base, int64_t off, u32 bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base + off), 0, (int)bytes, 0x00020000);
}

// the march: one x-tile per wave, U loads in flight (the product's k_cumsum_strided<2, 0, true, true, 4>)
template <int U>
__global__ __launch_bounds__(256) void k_march(const double* __restrict__ in, double* __restrict__ out, int n, int64_t inner) {
  const u32 pb = (gridDim.x + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const int64_t x = ((int64_t)lb * 256 + threadIdx.x) * 2;
  if (x >= inner) return;
  dv acc = {-0.0, -0.0};
  int k = 0;
  for (; k + U <= n; k += U) {
    dv v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const dv*>(in + (int64_t)(k + u) * inner + x));
#pragma unroll
    for (int u = 0; u < U; ++u) { acc += v[u]; __builtin_nontemporal_store(acc, reinterpret_cast<dv*>(out + (int64_t)(k + u) * inner + x)); }
  }
  for (; k < n; ++k) {
    acc += __builtin_nontemporal_load(reinterpret_cast<const dv*>(in + (int64_t)k * inner + x));
    __builtin_nontemporal_store(acc, reinterpret_cast<dv*>(out + (int64_t)k * inner + x));
  }
}

__global__ __launch_bounds__(256) void k_copy(const double* __restrict__ in, double* __restrict__ out, int64_t nvec) {
  const u32 pb = (gridDim.x + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const int64_t i = (int64_t)lb * 256 + threadIdx.x;
  if (i < nvec) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const dv*>(in) + i), reinterpret_cast<dv*>(out) + i);
}

// MODE 0: the scan; 1: loads and adds only (one store per tile at the end); 2: adds and stores only (no loads)
template <int NT, int G, int D, int MODE = 0>
__global__ __launch_bounds__(256) void k_levels(const double* __restrict__ in, double* __restrict__ out, int n, int64_t inner, u32 lanes_row) {
  constexpr int NG = NT / G;
  static_assert(NT % G == 0 && NG % D == 0 && D >= 2 && D <= NG, "static buffer indices");
  const u32 pb = (gridDim.x + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  const u32 w = __builtin_amdgcn_readfirstlane(lb * 4 + (threadIdx.x >> 6));
  const u32 t0 = w * NT;
  const u32 tiles_row = (lanes_row + 63) / 64;
  if (t0 >= tiles_row) return;
  const u32 own = tiles_row - t0;
  const u32 cnt = __builtin_amdgcn_readfirstlane(own < (u32)NT ? own : (u32)NT);
  const u32 row_bytes = lanes_row * 16u;
  const u32 xb = (threadIdx.x & 63) * 16u;
  u32 tk = t0;
  auto voff = [&](int j) -> int { return (int)((((u32)j < cnt) ? (tk + (u32)j) * 1024u : 0xfffffff0u) + xb); };
  dv acc[NT], buf[D][G];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = dv{-0.0, -0.0};
  auto load_group = [&](dv (&dst)[G], __amdgpu_buffer_rsrc_t rs, int jg) {
#pragma unroll
    for (int i = 0; i < G; ++i) dst[i] = __builtin_bit_cast(dv, __builtin_amdgcn_raw_buffer_load_b128(rs, voff(jg * G + i), 0, 2));
  };
  if (MODE != 2) {  // prologue: the first D - 1 groups of level 0
    const __amdgpu_buffer_rsrc_t r0 = rsrc(in, 0, row_bytes);
#pragma unroll
    for (int p = 0; p < D - 1; ++p) load_group(buf[p], r0, p);
  }
  for (int k = 0; k < n; ++k) {
    asm volatile("" : "+s"(tk));
    const __amdgpu_buffer_rsrc_t rcur = rsrc(in, uni((int64_t)k * inner), row_bytes);
    const __amdgpu_buffer_rsrc_t rnext = rsrc(in, uni((int64_t)(k + 1 < n ? k + 1 : k) * inner), k + 1 < n ? row_bytes : 0u);
    const __amdgpu_buffer_rsrc_t ro = rsrc(out, uni((int64_t)k * inner), row_bytes);
#pragma unroll
    for (int jg = 0; jg < NG; ++jg) {
      constexpr int AHEAD = D - 1;
      const int nx = jg + AHEAD;  // group to fetch now (wraps into the next level)
      if (MODE != 2) {
        if (nx < NG) load_group(buf[nx % D], rcur, nx);
        else load_group(buf[nx % D], rnext, nx - NG);  // NG % D == 0: (nx - NG) % D == nx % D
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const int j = jg * G + i;
        acc[j] = acc[j] + (MODE == 2 ? dv{1.0, 1.0} : buf[jg % D][i]);
        if (MODE != 1 || k == n - 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j]), ro, voff(j), 0, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

static double* d_in; static double* d_out; static double* d_ref;
static const int NZ = 75; static const int64_t INNER = 2400ll * 3600;

template <typename F> float timeit(F launch, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); launch(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

__global__ void k_cmp(const u64* a, const u64* b, int64_t n, u64* bad) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  u64 c = 0;
  for (; i < n; i += stride) c += (a[i] != b[i]);
  if (c) atomicAdd(bad, c);
}
static u64 mismatches() {
  u64* d; CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
  hipLaunchKernelGGL(k_cmp, dim3(8192), dim3(256), 0, 0, (const u64*)d_out, (const u64*)d_ref, (int64_t)NZ * INNER, d);
  u64 h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CK(hipFree(d)); return h;
}
__global__ void k_fill(double* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    u64 z = (u64)i * 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = (double)(z >> 11) * 0x1.0p-53 - 0.5;
  }
}

template <int NT, int G, int D, int MODE = 0> void run_levels(const char* tag) {
  int per_cu = 0, cus = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_levels<NT, G, D, MODE>, 256, 0));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const u32 lanes_row = (u32)(INNER / 2), tiles = (lanes_row + 63) / 64, waves = (tiles + NT - 1) / NT;
  const u32 blocks = (((waves + 3) / 4 + 7) / 8) * 8;
  CK(hipMemset(d_out, 0xff, sizeof(double) * NZ * INNER));
  auto launch = [&]() { hipLaunchKernelGGL((k_levels<NT, G, D, MODE>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, NZ, INNER, lanes_row); };
  const float ms = timeit(launch, 9);
  const u64 bad = MODE ? 0 : mismatches();
  const double bytes = (MODE ? 8.0 : 16.0) * NZ * INNER;
  printf("levels NT=%2d G=%d D=%d  %-10s waves %5u of %5d resident (%d wg/CU)  %.4f ms  %.3f of 8 TB/s  %s\n", NT, G, D, tag, waves, per_cu * cus * 4, per_cu, ms,
         bytes / (ms * 1e-3) / 8e12, MODE ? "(one direction: 8 B per cell)" : bad ? "BITS DIFFER" : "bits ok");
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2;
  CK(hipMalloc(&d_in, sizeof(double) * NZ * INNER)); CK(hipMalloc(&d_out, sizeof(double) * NZ * INNER)); CK(hipMalloc(&d_ref, sizeof(double) * NZ * INNER));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, d_in, (int64_t)NZ * INNER);
  const u32 mblocks = ((u32)((INNER / 2 + 255) / 256) + 7) / 8 * 8;
  hipLaunchKernelGGL(k_march<4>, dim3(mblocks), dim3(256), 0, 0, d_in, d_ref, NZ, INNER);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r) {
    printf("-- round %d\n", r);
    { auto l = [&]() { hipLaunchKernelGGL(k_copy, dim3((u32)((NZ * INNER / 2 + 255) / 256 + 7) / 8 * 8), dim3(256), 0, 0, d_in, d_out, NZ * INNER / 2); };
      float ms = timeit(l, 9); printf("copy                                                              %.4f ms  %.3f of 8 TB/s\n", ms, 16.0 * NZ * INNER / (ms * 1e-3) / 8e12); }
    { auto l = [&]() { hipLaunchKernelGGL(k_march<4>, dim3(mblocks), dim3(256), 0, 0, d_in, d_out, NZ, INNER); };
      float ms = timeit(l, 9); printf("march U=4                                                         %.4f ms  %.3f of 8 TB/s  %s\n", ms, 16.0 * NZ * INNER / (ms * 1e-3) / 8e12, mismatches() ? "BITS DIFFER" : "bits ok"); }
    { auto l = [&]() { hipLaunchKernelGGL(k_march<8>, dim3(mblocks), dim3(256), 0, 0, d_in, d_out, NZ, INNER); };
      float ms = timeit(l, 9); printf("march U=8                                                         %.4f ms  %.3f of 8 TB/s  %s\n", ms, 16.0 * NZ * INNER / (ms * 1e-3) / 8e12, mismatches() ? "BITS DIFFER" : "bits ok"); }
    run_levels<24, 4, 2>("");
    run_levels<24, 4, 3>("");
    run_levels<24, 4, 2, 1>("loads only");
    run_levels<24, 4, 2, 2>("stores only");
    run_levels<8, 4, 2, 1>("loads only");
    run_levels<8, 4, 2, 2>("stores only");
    run_levels<24, 2, 4>("");
    run_levels<32, 4, 2>("");
    run_levels<16, 4, 2>("2 gens");
    run_levels<16, 4, 4>("2 gens");
    run_levels<8, 4, 2>("2-3 gens");
    run_levels<8, 2, 4>("2-3 gens");
    run_levels<12, 2, 3>("2 gens");
  }
  return 0;
}
