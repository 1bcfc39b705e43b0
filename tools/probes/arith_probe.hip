// arith_probe.hip -- ceilings for the kernels that do real arithmetic per byte: what rate does a
// STRIPPED kernel reach that moves the same bytes and executes the same count of IEEE divisions / products per cell,
// with nothing else (no halo logic, no neighbour exchange, no searches)?  Tuning aid, not part of the product.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probes/arith_probe.hip -o tools/probes/arith_probe && tools/probes/arith_probe
//
// A. flat kernels on a (75, 2400, 3600) f64 field, one 16-B vector per thread, XCD-banded, band-major row order (all
//    levels of a band of 16 rows before the next band: the metric planes stay in the XCD's L2), non-temporal loads / stores:
//      copy                                   out = a                          (the streaming ceiling of this shape)
//      1 div         derivative               out = a / m1
//      1 mul 1 div   metric_weighted (1 axis) out = (a * m1) / m2
//      2 mul 2 div   metric_weighted (2 axes) out = (((a * m1) / m2) * m2) / m3  -- k_stencil2d_ys' arithmetic per cell
//      3 mul 2 div   the same + the product of the neighbouring cell, recomputed instead of exchanged
//    The metric planes are (2400, 3600), shared by the 75 levels (8 / 75 B per cell and plane).
// B. column marches (one lane = one column pair, 75 levels, planes 69 MB apart), the shape of the vertical transforms:
//      march copy    75 rows of phi + 76 rows of theta in, 50 rows out, no arithmetic
//      march + D     the same with D IEEE divisions per input cell feeding the outputs (conservative remap: one division
//                    per (cell, bin) overlap = (75 + 50) / 75 = 1.67 per cell on average)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NZ = 75, NY = 2400, NX = 3600, BAND = 16;
constexpr int VPR = NX / 2;  // 16-B vectors per row

__global__ void k_rand(double* p, size_t n, double lo, double hi, unsigned long long seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i + seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = lo + (hi - lo) * ((z >> 11) * 0x1.0p-53);
  }
}

// MODE: 0 copy, 1 one division, 2 product + division, 3 two products + two divisions, 4 three products + two divisions
template <int MODE>
__global__ __launch_bounds__(256) void k_flat(const double* __restrict__ a, double* __restrict__ out, const double* __restrict__ m1,
                                              const double* __restrict__ m2, const double* __restrict__ m3, u32 nblk) {
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);  // XCD banding
  if (lb >= nblk) return;
  // work item (wave-uniform row): band-major -- (band, level, row in band), one wave per 64 vectors of a row
  const u32 wave = lb * 4 + (threadIdx.x >> 6);
  const u32 tiles = (VPR + 63) / 64;              // 29 waves per row
  const u32 row_w = wave / tiles, tile = wave - row_w * tiles;
  const u32 per_band = NZ * BAND;
  const u32 band = row_w / per_band, rem = row_w - band * per_band;
  const u32 z = rem / BAND, y = band * BAND + (rem - z * BAND);
  if (y >= NY) return;
  const u32 v = tile * 64 + (threadIdx.x & 63);
  if (v >= VPR) return;
  const size_t cell = ((size_t)z * NY + y) * NX + 2 * (size_t)v;
  const size_t mcell = (size_t)y * NX + 2 * (size_t)v;
  d2 x = __builtin_nontemporal_load((const d2*)(a + cell));
  if (MODE >= 1) {
    const d2 p1 = *(const d2*)(m1 + mcell);
    if (MODE == 1) x = x / p1;
    else {
      const d2 p2 = *(const d2*)(m2 + mcell);
      x = (x * p1) / p2;
      if (MODE >= 3) {
        const d2 p3 = *(const d2*)(m3 + mcell);
        if (MODE == 4) {  // the neighbouring cell's product, recomputed (its field value is in the lane's own vector / next lane)
          d2 nb; nb.x = x.y; nb.y = x.x;
          x = x + nb * p1;
        }
        x = (x * p2) / p3;
      }
    }
  }
  __builtin_nontemporal_store(x, (d2*)(out + cell));
}

// the same arithmetic with ZK consecutive LEVELS per thread sharing the metric vectors (what the product's metric kernels do:
// the metric loads, L2 hits, compete with the field loads for the CU's outstanding-request capacity)
template <int MODE, int ZK>
__global__ __launch_bounds__(256) void k_flat_zk(const double* __restrict__ a, double* __restrict__ out, const double* __restrict__ m1,
                                                 const double* __restrict__ m2, const double* __restrict__ m3, u32 nblk) {
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 wave = lb * 4 + (threadIdx.x >> 6);
  constexpr u32 ZG = (NZ + ZK - 1) / ZK;
  const u32 tiles = (VPR + 63) / 64;
  const u32 row_w = wave / tiles, tile = wave - row_w * tiles;
  const u32 per_band = ZG * BAND;
  const u32 band = row_w / per_band, rem = row_w - band * per_band;
  const u32 zg = rem / BAND, y = band * BAND + (rem - zg * BAND);
  if (y >= NY) return;
  const u32 v = tile * 64 + (threadIdx.x & 63);
  if (v >= VPR) return;
  const size_t plane = (size_t)NY * NX;
  const size_t mcell = (size_t)y * NX + 2 * (size_t)v;
  const u32 z0 = zg * ZK;
  d2 x[ZK];
#pragma unroll
  for (int k = 0; k < ZK; ++k) x[k] = __builtin_nontemporal_load((const d2*)(a + (size_t)((z0 + k < NZ) ? z0 + k : NZ - 1) * plane + mcell));
  const d2 p1 = *(const d2*)(m1 + mcell);
  d2 p2 = p1, p3 = p1;
  if (MODE >= 2) p2 = *(const d2*)(m2 + mcell);
  if (MODE >= 3) p3 = *(const d2*)(m3 + mcell);
#pragma unroll
  for (int k = 0; k < ZK; ++k) {
    d2 t = x[k];
    if (MODE == 1) t = t / p1;
    if (MODE >= 2) t = (t * p1) / p2;
    if (MODE >= 3) t = (t * p2) / p3;
    if (z0 + k < NZ) __builtin_nontemporal_store(t, (d2*)(out + (size_t)(z0 + k) * plane + mcell));
  }
}

// The LADDER from the flat 4-levels-per-thread kernel towards the product's two-axis metric_weighted interp (K8 / K8y / K8m):
//   SEGR = 0: as k_flat_zk<3>: no stencil at all (reference point);
//   SEGR = 1, XS only (YS false): + the X stencil -- the neighbour product from the lane below by DPP, interp -- still one
//             intermediate row per output row;
//   SEGR = 1, YS: + the Y stencil with the row below RECOMPUTED: two X stages (loads, products, division round trip) per
//             output row -- K8 with one row per task;
//   SEGR = 2, YS: three X stages for two output rows -- K8's shape (SEG = 2);
//   SEGR = 4, YS: five for four.
// No boundary logic (indices clamped), no LDS, no barrier: what the operator's data flow costs by itself.
__device__ __forceinline__ double lane_below(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x138, 0xf, 0xf, false);
  const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x138, 0xf, 0xf, false);
  return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}
template <int ZK, int SEGR, bool YS>
__global__ __launch_bounds__(256) void k_ladder(const double* __restrict__ a, double* __restrict__ out, const double* __restrict__ m1,
                                                const double* __restrict__ m2, const double* __restrict__ m3, u32 nblk) {
  const u32 pb = (nblk + 7) >> 3;
  const u32 lb = (blockIdx.x & 7) * pb + (blockIdx.x >> 3);
  if (lb >= nblk) return;
  const u32 wave = lb * 4 + (threadIdx.x >> 6);
  constexpr u32 ZG = (NZ + ZK - 1) / ZK;
  constexpr u32 NSEG = NY / SEGR;                 // (2400 is a multiple of 1, 2, 4)
  constexpr u32 BSEG = BAND / SEGR;
  const u32 tiles = (VPR + 63) / 64;
  const u32 row_w = wave / tiles, tile = wave - row_w * tiles;
  const u32 per_band = ZG * BSEG;
  const u32 band = row_w / per_band, rem = row_w - band * per_band;
  const u32 zg = rem / BSEG, sg = band * BSEG + (rem - zg * BSEG);
  if (sg >= NSEG) return;
  const u32 v = tile * 64 + (threadIdx.x & 63);
  if (v >= VPR) return;
  const size_t plane = (size_t)NY * NX;
  const u32 z0 = zg * ZK;
  const u32 j0 = sg * SEGR;
  constexpr int NR = YS ? SEGR + 1 : SEGR;  // intermediate rows this thread computes
  d2 t[ZK][NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const u32 y = YS ? ((j0 + u == 0) ? 0 : j0 + u - 1) : j0 + u;
    const size_t mcell = (size_t)y * NX + 2 * (size_t)v;
    const d2 p1 = *(const d2*)(m1 + mcell), p2 = *(const d2*)(m2 + mcell);
    d2 x[ZK];
#pragma unroll
    for (int k = 0; k < ZK; ++k) x[k] = __builtin_nontemporal_load((const d2*)(a + (size_t)((z0 + k < NZ) ? z0 + k : NZ - 1) * plane + mcell));
#pragma unroll
    for (int k = 0; k < ZK; ++k) {
      const d2 pr = x[k] * p1;
      double nb = lane_below(pr.y);
      if ((threadIdx.x & 63) == 0) nb = pr.x;  // (the product's scalar load for this lane is not modelled)
      d2 i;
      i.x = (nb + pr.x) * 0.5;
      i.y = (pr.x + pr.y) * 0.5;
      t[k][u] = (i / p2) * p2;
    }
  }
#pragma unroll
  for (int u = 0; u < SEGR; ++u) {
    const size_t mcell = (size_t)(j0 + u) * NX + 2 * (size_t)v;
    const d2 p3 = *(const d2*)(m3 + mcell);
#pragma unroll
    for (int k = 0; k < ZK; ++k) {
      d2 r = YS ? (t[k][u] + t[k][u + 1]) * 0.5 : t[k][u];
      r = r / p3;
      if (z0 + k < NZ) __builtin_nontemporal_store(r, (d2*)(out + (size_t)(z0 + k) * plane + mcell));
    }
  }
}

// one lane = one column pair; D divisions per input cell; 50 output rows per column
template <int D>
__global__ __launch_bounds__(256) void k_march(const double* __restrict__ phi, const double* __restrict__ theta, double* __restrict__ out,
                                               u32 ncol2, int m) {
  const u32 c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncol2) return;
  const size_t plane = (size_t)NY * NX;
  const double* pp = phi + 2 * (size_t)c;
  const double* pt = theta + 2 * (size_t)c;
  double* po = out + 2 * (size_t)c;
  d2 acc = {0.0, 0.0};
  d2 t0 = __builtin_nontemporal_load((const d2*)pt);
  int jo = 0;
  for (int k = 0; k < NZ; k += 5) {
    d2 p[5], t[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      p[u] = __builtin_nontemporal_load((const d2*)(pp + (size_t)(k + u) * plane));
      t[u] = __builtin_nontemporal_load((const d2*)(pt + (size_t)(k + u + 1) * plane));
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      d2 dz = t[u] - t0;
      d2 v = p[u];
      if (D == 99) {  // DIVERGENT: a cell overlaps 0 .. 4 bins depending on its own thickness (random-walk columns): every
                      // lane its own trip count (mean 1.67), the wave runs the longest of its 64
        const int trips = (int)((__builtin_bit_cast(unsigned long long, dz.x) >> 40) % 6u);  // 0..5 from mantissa bits
        const int nt = trips < 2 ? 1 : (trips < 4 ? 2 : (trips == 4 ? 0 : 4));                // 1,1,2,2,0,4 -> mean 1.67
        for (int d = 0; d < nt; ++d) v = v / dz;
      } else {
#pragma unroll
      for (int d = 0; d < D; ++d) v = v / dz;  // dependent IEEE divisions (the overlap fraction feeds the accumulation)
      }
      if (D == 0) v = v + dz;                  // (keeps the theta loads alive without a division)
      acc = acc + v;
      t0 = t[u];
      // 50 rows out per 75 in: two rows every three cells
      if ((k + u) % 3 != 2 && jo < m) { __builtin_nontemporal_store(acc, (d2*)(po + (size_t)jo * plane)); ++jo; }
    }
  }
}

// the same march on FLOAT32 data with V columns per lane (4-, 8-, 16-byte lanes): what a float32 transform column costs
// when a lane owns one column (a 256-B wave-row) against two or four
template <int V, int D>
__global__ __launch_bounds__(256) void k_march_f32(const float* __restrict__ phi, const float* __restrict__ theta, float* __restrict__ out,
                                                   u32 ngroups, int m) {
  typedef float fv __attribute__((ext_vector_type(V)));
  const u32 c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ngroups) return;
  const size_t plane = (size_t)NY * NX;
  const float* pp = phi + (size_t)V * c;
  const float* pt = theta + (size_t)V * c;
  float* po = out + (size_t)V * c;
  fv acc = 0.0f;
  fv t0 = __builtin_nontemporal_load((const fv*)pt);
  int jo = 0;
  for (int k = 0; k < NZ; k += 5) {
    fv p[5], t[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      p[u] = __builtin_nontemporal_load((const fv*)(pp + (size_t)(k + u) * plane));
      t[u] = __builtin_nontemporal_load((const fv*)(pt + (size_t)(k + u + 1) * plane));
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      fv dz = t[u] - t0;
      fv v = p[u];
#pragma unroll
      for (int d = 0; d < D; ++d) v = v / dz;
      if (D == 0) v = v + dz;
      acc = acc + v;
      t0 = t[u];
      if ((k + u) % 3 != 2 && jo < m) { __builtin_nontemporal_store(acc, (fv*)(po + (size_t)jo * plane)); ++jo; }
    }
  }
}

template <typename F>
float timeit(F f, int reps = 9) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const size_t plane = (size_t)NY * NX, n = plane * NZ;
  const int m = 50;
  double *a, *o, *m1, *m2, *m3, *th;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&o, n * 8)); CK(hipMalloc(&th, (n + plane) * 8));
  CK(hipMalloc(&m1, plane * 8)); CK(hipMalloc(&m2, plane * 8)); CK(hipMalloc(&m3, plane * 8));
  hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, a, n, -0.5, 0.5, 2ull);
  hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, th, n + plane, 1.0, 2.0, 3ull);
  hipLaunchKernelGGL(k_rand, dim3(1024), dim3(256), 0, 0, m1, plane, 1000.0, 2000.0, 31ull);
  hipLaunchKernelGGL(k_rand, dim3(1024), dim3(256), 0, 0, m2, plane, 1000.0, 2000.0, 32ull);
  hipLaunchKernelGGL(k_rand, dim3(1024), dim3(256), 0, 0, m3, plane, 1000.0, 2000.0, 33ull);
  CK(hipDeviceSynchronize());
  const u32 tiles = (VPR + 63) / 64;
  const size_t rows_w = (size_t)((NY + BAND - 1) / BAND) * BAND * NZ;
  const u32 nblk = (u32)((rows_w * tiles + 3) / 4), grid = ((nblk + 7) / 8) * 8;
  auto rep = [&](const char* name, float ms, double bytes) {
    printf("%-58s %8.4f ms  %8.1f GB/s  %.3f of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000); fflush(stdout); };
  for (int rnd = 0; rnd < 2; ++rnd) {
#define FLAT(MODE, name, planes) { float ms = timeit([&] { hipLaunchKernelGGL((k_flat<MODE>), dim3(grid), dim3(256), 0, 0, a, o, m1, m2, m3, nblk); }); \
    rep(name, ms, 16.0 * n + planes * 8.0 * plane); }
    FLAT(0, "A flat copy (band-major rows)", 0)
    FLAT(1, "A 1 div            (derivative)", 1)
    FLAT(2, "A 1 mul 1 div      (metric_weighted, one axis)", 2)
    FLAT(3, "A 2 mul 2 div      (metric_weighted, two axes)", 3)
    FLAT(4, "A 3 mul 2 div      (+ neighbour product recomputed)", 3)
#define FLATZK(MODE, ZK, name, planes) { const u32 zg = (NZ + ZK - 1) / ZK; const size_t rw = (size_t)((NY + BAND - 1) / BAND) * BAND * zg; \
    const u32 nb = (u32)((rw * tiles + 3) / 4), gr = ((nb + 7) / 8) * 8; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_flat_zk<MODE, ZK>), dim3(gr), dim3(256), 0, 0, a, o, m1, m2, m3, nb); }); \
    rep(name, ms, 16.0 * n + planes * 8.0 * plane); }
    FLATZK(0, 4, "A copy, 4 levels per thread", 0)
    FLATZK(1, 2, "A 1 div, 2 levels per thread sharing the metric", 1)
    FLATZK(2, 4, "A 1 mul 1 div, 4 levels per thread", 2)
    FLATZK(3, 4, "A 2 mul 2 div, 4 levels per thread", 3)
    FLATZK(3, 8, "A 2 mul 2 div, 8 levels per thread", 3)
#define LADDER(ZK, SEGR, YS, name) { const u32 zg = (NZ + ZK - 1) / ZK; const size_t rw = (size_t)((NY / SEGR + BAND / SEGR - 1) / (BAND / SEGR)) * (BAND / SEGR) * zg; \
    const u32 nb = (u32)((rw * tiles + 3) / 4), gr = ((nb + 7) / 8) * 8; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_ladder<ZK, SEGR, YS>), dim3(gr), dim3(256), 0, 0, a, o, m1, m2, m3, nb); }); \
    rep(name, ms, 16.0 * n + 3 * 8.0 * plane); }
    LADDER(4, 1, false, "L 2 mul 2 div + X stencil (DPP neighbour), 4 levels")
    LADDER(4, 1, true, "L + Y stencil, row below recomputed: 2 X stages per row")
    LADDER(4, 2, true, "L + Y stencil, 3 X stages per 2 rows (K8's shape)")
    LADDER(4, 4, true, "L + Y stencil, 5 X stages per 4 rows")
    LADDER(2, 4, true, "L + Y stencil, 5 X stages per 4 rows, 2 levels")
    const u32 ncol2 = (u32)(plane / 2);
#define MARCH(D, name) { float ms = timeit([&] { hipLaunchKernelGGL((k_march<D>), dim3((ncol2 + 255) / 256), dim3(256), 0, 0, a, th, o, ncol2, m); }); \
    rep(name, ms, 8.0 * (2.0 * n + plane + (double)m * plane)); }
    MARCH(0, "B march: 75 + 76 rows in, 50 rows out, no arithmetic")
    MARCH(1, "B march + 1 division per cell")
    MARCH(2, "B march + 2 divisions per cell (conservative: 1.67 on average)")
    MARCH(99, "B march + 0..4 divisions per cell, DIVERGENT per lane (mean 1.67)")
    // float32: the same buffers read as floats (twice the columns per plane are not needed: the first half of every plane)
#define MARCHF(V, D, name) { const u32 ng = (u32)(plane / V); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_march_f32<V, D>), dim3((ng + 255) / 256), dim3(256), 0, 0, (const float*)a, (const float*)th, (float*)o, ng, m); }); \
    rep(name, ms, 4.0 * (2.0 * n + plane + (double)m * plane)); }
    MARCHF(1, 0, "F float32 march, 1 column per lane (4-byte lanes), no arithmetic")
    MARCHF(2, 0, "F float32 march, 2 columns per lane (8-byte lanes)")
    MARCHF(4, 0, "F float32 march, 4 columns per lane (16-byte lanes)")
    MARCHF(1, 1, "F float32 march, 1 column per lane + 1 division per cell")
    MARCHF(2, 1, "F float32 march, 2 columns per lane + 1 division per cell")
    MARCHF(4, 1, "F float32 march, 4 columns per lane + 1 division per cell")
  }
  return 0;
}
