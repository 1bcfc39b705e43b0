// streambench.hip -- what HBM streaming rate can a read-one/write-one f64 kernel reach on this
// MI355X, and with which launch shape?  Tuning aid for xgcm_amd/csrc/xg_*.hip (not part of the product).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/streambench.hip -o gpurun_out/streambench && ./streambench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// tile = BLOCK*R vec elements; thread handles t + r*BLOCK within its tile. REMAP: XCD-contiguous chunks.
template <int R, bool NTL, bool NTS, bool REMAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy(const d2* __restrict__ in, d2* __restrict__ out, size_t nvec, unsigned nblk) {
  unsigned b = blockIdx.x;
  if (REMAP) { unsigned per = nblk / 8; if (b < per * 8) b = (b % 8) * per + b / 8; }
  size_t base = (size_t)b * BLOCK * R + threadIdx.x;
  d2 v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { size_t i = base + (size_t)r * BLOCK; if (i < nvec) v[r] = ld<NTL>(in + i); }
#pragma unroll
  for (int r = 0; r < R; ++r) { size_t i = base + (size_t)r * BLOCK; if (i < nvec) st<NTS>(out + i, v[r]); }
}

// persistent grid-stride copy
template <int R, bool NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_copy_gs(const d2* __restrict__ in, d2* __restrict__ out, size_t nvec) {
  size_t stride = (size_t)gridDim.x * BLOCK * R;
  for (size_t base = (size_t)blockIdx.x * BLOCK * R + threadIdx.x; base < nvec; base += stride) {
    d2 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { size_t i = base + (size_t)r * BLOCK; if (i < nvec) v[r] = in[i]; }
#pragma unroll
    for (int r = 0; r < R; ++r) { size_t i = base + (size_t)r * BLOCK; if (i < nvec) st<NTS>(out + i, v[r]); }
  }
}

// diff along contiguous rows of length L (periodic), R rows per thread (same shape as k_stencil_contig)
template <int R, bool NTS>
__global__ __launch_bounds__(256) void k_diffx(const double* __restrict__ in, double* __restrict__ out, size_t rows, int L, unsigned ntile) {
  unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned tile = w % ntile; size_t r0 = (size_t)(w / ntile) * R;
  if (r0 >= rows) return;
  int i0 = (tile * 64 + (threadIdx.x & 63)) * 2;
  if (i0 >= L) return;
  int nidx = i0 == 0 ? L - 1 : i0 - 1;
  d2 pr[R]; double nb[R];
#pragma unroll
  for (int u = 0; u < R; ++u) if (r0 + u < rows) { const double* p = in + (r0 + u) * L; pr[u] = *(const d2*)(p + i0); nb[u] = p[nidx]; }
#pragma unroll
  for (int u = 0; u < R; ++u) if (r0 + u < rows) { d2 o; o.x = pr[u].x - nb[u]; o.y = pr[u].y - pr[u].x; st<NTS>((d2*)(out + (r0 + u) * L + i0), o); }
}


// E1: flat mapping: thread gid -> (row, i0) by division; perfectly linear, no partial waves
template <bool NTS>
__global__ __launch_bounds__(256) void k_diffx_flat(const double* __restrict__ in, double* __restrict__ out, size_t nvec, unsigned vpr) {
  size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nvec) return;
  size_t row = gid / vpr; unsigned xv = (unsigned)(gid - row * vpr);
  unsigned L = vpr * 2, i0 = xv * 2;
  const double* p = in + row * L;
  unsigned nidx = i0 == 0 ? L - 1 : i0 - 1;
  d2 a = *(const d2*)(p + i0); double nb = p[nidx];
  d2 o; o.x = a.x - nb; o.y = a.y - a.x;
  st<NTS>((d2*)(out + row * L + i0), o);
}

// E2/E3: stencil along a strided axis in LINEAR output order: out[j,x] = in[j,x] - in[j-1,x], both
// loaded (the j-1 row is an L2 / MALL hit).  ntile_pad = tiles per row rounded for XCD alignment.
template <bool NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_diffy_two(const double* __restrict__ in, double* __restrict__ out, size_t nrows, size_t nper,
                                                      size_t inner, unsigned ntile_pad) {
  constexpr int WPB = BLOCK / 64;
  size_t w = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
  unsigned tile = (unsigned)(w % ntile_pad); size_t row = w / ntile_pad;
  if (row >= nrows) return;
  size_t x = ((size_t)tile * 64 + (threadIdx.x & 63)) * 2;
  if (x >= inner) return;
  size_t j = row % nper;
  size_t prev = j == 0 ? row : row - 1;
  d2 a = *(const d2*)(in + row * inner + x);
  d2 b = *(const d2*)(in + prev * inner + x);
  d2 o = a - b;
  st<NTS>((d2*)(out + row * inner + x), o);
}

// marching reference (same as k_stencil_strided): SEG rows per wave
template <int SEG, bool NTS>
__global__ __launch_bounds__(256) void k_diffy_march(const double* __restrict__ in, double* __restrict__ out, size_t outer, size_t n, size_t inner, unsigned ntile) {
  size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned nseg = (unsigned)((n + SEG - 1) / SEG);
  unsigned tile = (unsigned)(w % ntile); size_t r = w / ntile; unsigned sg = (unsigned)(r % nseg); size_t o = r / nseg;
  if (o >= outer) return;
  size_t x = ((size_t)tile * 64 + (threadIdx.x & 63)) * 2;
  if (x >= inner) return;
  size_t j0 = (size_t)sg * SEG, j1 = j0 + SEG < n ? j0 + SEG : n;
  const double* p = in + o * n * inner + x; double* q = out + o * n * inner + x;
  d2 prev = *(const d2*)(p + (j0 == 0 ? 0 : j0 - 1) * inner);
  size_t j = j0;
  for (; j + 8 <= j1; j += 8) { d2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const d2*)(p + (j + u) * inner);
#pragma unroll
    for (int u = 0; u < 8; ++u) { st<NTS>((d2*)(q + (j + u) * inner), v[u] - prev); prev = v[u]; } }
  for (; j < j1; ++j) { d2 v = *(const d2*)(p + j * inner); st<NTS>((d2*)(q + j * inner), v - prev); prev = v; }
}


// XCD-banded variants: block b runs on XCD b%8 (observed); give each XCD a contiguous 1/8 of the
// linear wave sequence so a row re-read one row later hits the SAME XCD's L2.
__device__ __forceinline__ bool banded(unsigned nb, size_t& w) {
  unsigned b = blockIdx.x, pb = (nb + 7) / 8;
  unsigned lb = (b % 8) * pb + b / 8;
  if (lb >= nb) return false;
  w = (size_t)lb * 4 + (threadIdx.x >> 6);
  return true;
}
template <bool BAND>
__global__ __launch_bounds__(256) void k_diffy_two_b(const double* __restrict__ in, double* __restrict__ out, size_t nrows, unsigned nper,
                                                      unsigned inner, unsigned ntile, unsigned nb) {
  size_t w;
  if (BAND) { if (!banded(nb, w)) return; } else w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned tile = (unsigned)(w % ntile); size_t row = w / ntile;
  if (row >= nrows) return;
  unsigned x = (tile * 64 + (threadIdx.x & 63)) * 2;
  if (x >= inner) return;
  unsigned j = (unsigned)(row % nper);
  size_t prev = j == 0 ? row : row - 1;
  d2 a = *(const d2*)(in + row * inner + x);
  d2 b = *(const d2*)(in + prev * inner + x);
  st<true>((d2*)(out + row * inner + x), a - b);
}
template <int SEG, bool BAND>
__global__ __launch_bounds__(256) void k_diffy_march_b(const double* __restrict__ in, double* __restrict__ out, size_t outer, unsigned n, unsigned inner, unsigned ntile, unsigned nb) {
  size_t w;
  if (BAND) { if (!banded(nb, w)) return; } else w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned nseg = (n + SEG - 1) / SEG;
  unsigned tile = (unsigned)(w % ntile); size_t r = w / ntile; unsigned sg = (unsigned)(r % nseg); size_t o = r / nseg;
  if (o >= outer) return;
  unsigned x = (tile * 64 + (threadIdx.x & 63)) * 2;
  if (x >= inner) return;
  unsigned j0 = sg * SEG, j1 = j0 + SEG < n ? j0 + SEG : n;
  const double* p = in + o * n * inner + x; double* q = out + o * n * inner + x;
  d2 v[SEG + 1];
  v[0] = *(const d2*)(p + (size_t)(j0 == 0 ? 0 : j0 - 1) * inner);
#pragma unroll
  for (int u = 0; u < SEG; ++u) if (j0 + u < j1) v[u + 1] = *(const d2*)(p + (size_t)(j0 + u) * inner);
#pragma unroll
  for (int u = 0; u < SEG; ++u) if (j0 + u < j1) st<true>((d2*)(q + (size_t)(j0 + u) * inner), v[u + 1] - v[u]);
}


// Z-march with big workgroups kept in lockstep by barriers (DRAM page locality experiment)
template <int BLK, int U, bool SYNC, bool BAND>
__global__ __launch_bounds__(BLK) void k_diffz_march(const double* __restrict__ in, double* __restrict__ out, unsigned n, size_t inner, unsigned nblk) {
  unsigned b = blockIdx.x;
  if (BAND) { unsigned pb = (nblk + 7) / 8; b = (b % 8) * pb + b / 8; if (b >= nblk) return; }
  size_t x = ((size_t)b * BLK + threadIdx.x) * 2;
  bool act = x < inner;
  const double* p = in + (act ? x : 0); double* q = out + (act ? x : 0);
  d2 prev = *(const d2*)p;
  for (unsigned j = 0; j < n; j += U) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (j + u < n) v[u] = *(const d2*)(p + (size_t)(j + u) * inner);
#pragma unroll
    for (int u = 0; u < U; ++u) if (j + u < n) { if (act) st<true>((d2*)(q + (size_t)(j + u) * inner), v[u] - prev); prev = v[u]; }
    if (SYNC) __syncthreads();
  }
}
template <bool BAND>
__global__ __launch_bounds__(256) void k_diffx_flat_b(const double* __restrict__ in, double* __restrict__ out, size_t nvec, unsigned vpr, unsigned nblk) {
  unsigned b = blockIdx.x;
  if (BAND) { unsigned pb = (nblk + 7) / 8; b = (b % 8) * pb + b / 8; if (b >= nblk) return; }
  size_t gid = (size_t)b * 256 + threadIdx.x;
  if (gid >= nvec) return;
  size_t row = gid / vpr; unsigned xv = (unsigned)(gid - row * vpr);
  unsigned L = vpr * 2, i0 = xv * 2;
  const double* p = in + row * L;
  unsigned nidx = i0 == 0 ? L - 1 : i0 - 1;
  d2 a = *(const d2*)(p + i0); double nb = p[nidx];
  d2 o; o.x = a.x - nb; o.y = a.y - a.x;
  st<true>((d2*)(out + row * L + i0), o);
}


// occupancy-limited Z-march (dynamic LDS just to cap resident workgroups per CU), runtime n/inner
template <int U, int BLK = 256>
__global__ __launch_bounds__(BLK) void k_diffz_occ(const double* __restrict__ in, double* __restrict__ out, unsigned n, size_t inner) {
  extern __shared__ double lds_dummy[];
  size_t x = ((size_t)blockIdx.x * BLK + threadIdx.x) * 2;
  if (x >= inner) return;
  if (n == 0xffffffffu) lds_dummy[threadIdx.x] = 1.0;  // keep the allocation alive
  const double* p = in + x; double* q = out + x;
  d2 prev = *(const d2*)p;
  for (unsigned j = 0; j < n; j += U) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (j + u < n) v[u] = *(const d2*)(p + (size_t)(j + u) * inner);
#pragma unroll
    for (int u = 0; u < U; ++u) if (j + u < n) { st<true>((d2*)(q + (size_t)(j + u) * inner), v[u] - prev); prev = v[u]; }
  }
}


// Z-march where each lane owns RT x-tiles that are ADJACENT (so a wave touches RT KiB contiguous per
// level) and keeps U levels in flight: RT*U loads per lane, grouped by DRAM locality.
template <int RT, int U>
__global__ __launch_bounds__(256) void k_diffz_rt(const double* __restrict__ in, double* __restrict__ out, unsigned n, size_t inner) {
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  size_t x0 = ((size_t)wave * RT * 64 + lane) * 2;  // tile r at x0 + r*128
  if (x0 >= inner) return;
  d2 prev[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) prev[r] = (x0 + (size_t)r * 128 < inner) ? *(const d2*)(in + x0 + (size_t)r * 128) : d2{0, 0};
  for (unsigned j = 0; j < n; j += U) {
    d2 v[U][RT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < RT; ++r) if (j + u < n && x0 + (size_t)r * 128 < inner) v[u][r] = *(const d2*)(in + (size_t)(j + u) * inner + x0 + (size_t)r * 128);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < RT; ++r) if (j + u < n && x0 + (size_t)r * 128 < inner) { st<true>((d2*)(out + (size_t)(j + u) * inner + x0 + (size_t)r * 128), v[u][r] - prev[r]); prev[r] = v[u][r]; }
  }
}

__global__ void k_rand(double* out, size_t n) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    unsigned long long z = i + 0x9E3779B97F4A7C15ull * 2;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    out[i] = (double)(z >> 11) * 0x1.0p-53 - 0.5;
  }
}

template <typename F> float timeit(F f, int reps = 8) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}


// read-only / write-only ceilings: one 16-B vector per thread, linear order
template <bool NTL>
__global__ __launch_bounds__(256) void k_readonly(const d2* __restrict__ in, double* __restrict__ sink, size_t nvec) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  d2 v = ld<NTL>(in + i);
  if (v.x == 1.2345e300 && v.y == -1.0) sink[0] = v.x;  // never true: keeps the load alive
}
template <bool NTS>
__global__ __launch_bounds__(256) void k_writeonly(d2* __restrict__ out, size_t nvec) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  d2 v; v.x = (double)i; v.y = 0.5;
  st<NTS>(out + i, v);
}

// three streams (two reads, one write), one 16-B vector of each per thread: the shape of `a * b`, of the fused
// vorticity (u, v -> zeta) and of every kernel that carries a full-size second operand
template <bool NTL, bool NTS, bool REMAP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_triad(const d2* __restrict__ a, const d2* __restrict__ b, d2* __restrict__ out, size_t nvec, unsigned nblk) {
  unsigned blk = blockIdx.x;
  if (REMAP) { unsigned per = (nblk + 7) / 8; blk = (blk % 8) * per + blk / 8; if (blk >= nblk) return; }
  size_t i = (size_t)blk * BLOCK + threadIdx.x;
  if (i >= nvec) return;
  d2 x = ld<NTL>(a + i), y = ld<NTL>(b + i);
  d2 o; o.x = x.x * y.x; o.y = x.y * y.y;
  st<NTS>(out + i, o);
}

int main() {
  if (getenv("SB_TRIAD")) {
    const size_t n = 75ull * 2400 * 3600, nvec = n / 2;
    double *a, *b, *o;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&o, n * 8));
    hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, a, n); hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, b, n); CK(hipDeviceSynchronize());
    auto rep3 = [&](const char* name, float ms) { printf("%-52s %8.4f ms  %8.1f GB/s  %.3f of 8TB/s (24 B per pair of cells)\n", name, ms, 3.0 * n * 8 / ms / 1e6, 3.0 * n * 8 / ms / 1e6 / 8000); fflush(stdout); };
#define TRIAD(NTL, NTS, REMAP, BLOCK) { unsigned nblk = (unsigned)((nvec + BLOCK - 1) / BLOCK); unsigned grid = REMAP ? ((nblk + 7) / 8) * 8 : nblk; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_triad<NTL, NTS, REMAP, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, (const d2*)a, (const d2*)b, (d2*)o, nvec, nblk); }, 12); \
    rep3("triad ntl=" #NTL " nts=" #NTS " banded=" #REMAP " blk=" #BLOCK, ms); }
    for (int rnd = 0; rnd < 2; ++rnd) {
      TRIAD(false, false, false, 256) TRIAD(false, true, false, 256) TRIAD(true, true, false, 256) TRIAD(true, true, true, 256)
      TRIAD(false, true, true, 256) TRIAD(true, true, false, 64) TRIAD(true, true, false, 128) TRIAD(true, true, false, 512) TRIAD(true, true, false, 1024)
    }
    return 0;
  }
  const size_t n = 75ull * 2400 * 3600;  // doubles
  const size_t nvec = n / 2;
  double *in, *out;
  CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
  CK(hipMemset(out, 0, n * 8));
  if (getenv("SB_CONST")) { CK(hipMemset(in, 1, n * 8)); printf("# input: constant bytes (outputs of diff are ZERO -> inflated rates)\n"); }
  else { hipLaunchKernelGGL(k_rand, dim3(8192), dim3(256), 0, 0, in, n); CK(hipDeviceSynchronize()); printf("# input: uniform random doubles\n"); }
  auto report = [&](const char* name, float ms) { printf("%-44s %8.4f ms  %8.1f GB/s  %.3f of 8TB/s\n", name, ms, 2.0 * n * 8 / ms / 1e6, 2.0 * n * 8 / ms / 1e6 / 8000); fflush(stdout); };
  { float ms = timeit([&] { CK(hipMemcpyAsync(out, in, n * 8, hipMemcpyDeviceToDevice, 0)); }); report("hipMemcpy D2D", ms); }
#define COPY(R, NTL, NTS, REMAP, BLOCK) { unsigned nblk = (unsigned)((nvec + (size_t)BLOCK * R - 1) / ((size_t)BLOCK * R)); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_copy<R, NTL, NTS, REMAP, BLOCK>), dim3(nblk), dim3(BLOCK), 0, 0, (const d2*)in, (d2*)out, nvec, nblk); }); \
    report("copy R=" #R " ntl=" #NTL " nts=" #NTS " remap=" #REMAP " blk=" #BLOCK, ms); }
  if (getenv("SB_EXTRA")) {  // block-size sweep of the one-vector-per-thread copy, read-only and write-only ceilings
    COPY(1, false, true, false, 64) COPY(1, false, true, false, 128) COPY(1, false, true, false, 512) COPY(1, false, true, false, 1024)
    COPY(1, false, true, true, 256) COPY(1, true, true, false, 256)
    auto report1 = [&](const char* name, float ms) { printf("%-44s %8.4f ms  %8.1f GB/s  %.3f of 8TB/s (one-way bytes)\n", name, ms, 1.0 * n * 8 / ms / 1e6, 1.0 * n * 8 / ms / 1e6 / 8000); fflush(stdout); };
    { float ms = timeit([&] { hipLaunchKernelGGL((k_readonly<false>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const d2*)in, out, nvec); }); report1("read-only 16 B/thread", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL((k_readonly<true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (const d2*)in, out, nvec); }); report1("read-only 16 B/thread, nt loads", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL((k_writeonly<false>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (d2*)out, nvec); }); report1("write-only 16 B/thread", ms); }
    { float ms = timeit([&] { hipLaunchKernelGGL((k_writeonly<true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (d2*)out, nvec); }); report1("write-only 16 B/thread, nt stores", ms); }
    return 0;
  }
  COPY(1, false, false, false, 256) COPY(1, false, true, false, 256) COPY(2, false, true, false, 256) COPY(4, false, true, false, 256)
  COPY(8, false, true, false, 256) COPY(16, false, true, false, 256) COPY(4, true, true, false, 256) COPY(4, false, true, true, 256)
  COPY(8, false, true, true, 256) COPY(4, false, true, false, 512) COPY(4, false, true, false, 1024) COPY(8, false, true, false, 64)
  COPY(8, false, false, false, 256)
#define GS(R, NTS, BLOCK, NB) { float ms = timeit([&] { hipLaunchKernelGGL((k_copy_gs<R, NTS, BLOCK>), dim3(NB), dim3(BLOCK), 0, 0, (const d2*)in, (d2*)out, nvec); }); \
    report("grid-stride R=" #R " nts=" #NTS " blk=" #BLOCK " blocks=" #NB, ms); }
  GS(4, true, 256, 2048) GS(4, true, 256, 4096) GS(8, true, 256, 2048) GS(4, true, 512, 1024) GS(4, true, 1024, 512) GS(4, true, 256, 8192)
#define DX(R, NTS) { const int L = 3600; size_t rows = n / L; unsigned ntile = (L / 2 + 63) / 64; size_t ntask = (size_t)ntile * ((rows + R - 1) / R); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffx<R, NTS>), dim3((unsigned)((ntask + 3) / 4)), dim3(256), 0, 0, in, out, rows, L, ntile); }); report("diffx R=" #R " nts=" #NTS, ms); }
  DX(2, true) DX(4, true) DX(8, true) DX(16, true) DX(8, false)

  DX(1, true)
  { unsigned vpr = 1800; float ms = timeit([&] { hipLaunchKernelGGL((k_diffx_flat<true>), dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, in, out, nvec, vpr); }); report("diffx flat nts", ms); }
#define DY2(BLOCK, NT, NPER, INNER, NAME) { size_t inner = INNER, nper = NPER, nrows = n / inner; unsigned ntile = (unsigned)((inner / 2 + 63) / 64); unsigned ntp = NT ? NT : ntile; \
    size_t nw = (size_t)ntp * nrows; constexpr int WPB_ = BLOCK / 64; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffy_two<true, BLOCK>), dim3((unsigned)((nw + WPB_ - 1) / WPB_)), dim3(BLOCK), 0, 0, in, out, nrows, nper, inner, ntp); }); report(NAME, ms); }
  DY2(256, 0, 2400, 3600, "diffY two-load linear ntile=29 blk256")
  DY2(256, 32, 2400, 3600, "diffY two-load linear ntile=32(xcd) blk256")
  DY2(64, 32, 2400, 3600, "diffY two-load linear ntile=32(xcd) blk64")
  DY2(64, 0, 2400, 3600, "diffY two-load linear ntile=29 blk64")
  DY2(256, 0, 75, 8640000, "diffZ two-load linear blk256")
#define DYM(SEG, OUTER, N, INNER, NAME) { size_t inner = INNER, nn = N, outer = OUTER; unsigned ntile = (unsigned)((inner / 2 + 63) / 64); unsigned nseg = (unsigned)((nn + SEG - 1) / SEG); \
    size_t nw = (size_t)ntile * nseg * outer; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffy_march<SEG, true>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, 0, in, out, outer, nn, inner, ntile); }); report(NAME, ms); }
  DYM(2, 75, 2400, 3600, "diffY march SEG=2") DYM(4, 75, 2400, 3600, "diffY march SEG=4") DYM(8, 75, 2400, 3600, "diffY march SEG=8") DYM(16, 75, 2400, 3600, "diffY march SEG=16") DYM(64, 75, 2400, 3600, "diffY march SEG=64")
  DYM(75, 1, 75, 8640000, "diffZ march SEG=75") DYM(25, 1, 75, 8640000, "diffZ march SEG=25") DYM(15, 1, 75, 8640000, "diffZ march SEG=15")

#define DY2B(BAND, NAME) { unsigned inner = 3600, nper = 2400; size_t nrows = n / inner; unsigned ntile = (inner / 2 + 63) / 64; size_t nw = (size_t)ntile * nrows; unsigned nb = (unsigned)((nw + 3) / 4); \
    unsigned grid = BAND ? ((nb + 7) / 8) * 8 : nb; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffy_two_b<BAND>), dim3(grid), dim3(256), 0, 0, in, out, nrows, nper, inner, ntile, nb); }); report(NAME, ms); }
  DY2B(false, "diffY two-load (u32 math) plain") DY2B(true, "diffY two-load XCD-banded")
#define DYMB(SEG, BAND, NAME) { unsigned inner = 3600, nn = 2400; size_t outer = 75; unsigned ntile = (inner / 2 + 63) / 64; unsigned nseg = (nn + SEG - 1) / SEG; \
    size_t nw = (size_t)ntile * nseg * outer; unsigned nb = (unsigned)((nw + 3) / 4); unsigned grid = BAND ? ((nb + 7) / 8) * 8 : nb; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffy_march_b<SEG, BAND>), dim3(grid), dim3(256), 0, 0, in, out, outer, nn, inner, ntile, nb); }); report(NAME, ms); }
  DYMB(1, false, "diffY regmarch SEG=1 plain") DYMB(2, false, "diffY regmarch SEG=2 plain") DYMB(3, false, "diffY regmarch SEG=3 plain") DYMB(4, false, "diffY regmarch SEG=4 plain")
  DYMB(6, false, "diffY regmarch SEG=6 plain")
  DYMB(1, true, "diffY regmarch SEG=1 banded") DYMB(2, true, "diffY regmarch SEG=2 banded") DYMB(3, true, "diffY regmarch SEG=3 banded") DYMB(4, true, "diffY regmarch SEG=4 banded")
  DYMB(8, true, "diffY regmarch SEG=8 banded")

#define DZ(BLK, U, SYNC, BAND, NAME) { size_t inner = 8640000; unsigned nn = 75; unsigned nblk = (unsigned)((inner / 2 + BLK - 1) / BLK); unsigned grid = BAND ? ((nblk + 7) / 8) * 8 : nblk; \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffz_march<BLK, U, SYNC, BAND>), dim3(grid), dim3(BLK), 0, 0, in, out, nn, inner, nblk); }); report(NAME, ms); }
  DZ(256, 8, false, false, "diffZ march blk256 U8") DZ(256, 5, false, false, "diffZ march blk256 U5") DZ(256, 3, false, false, "diffZ march blk256 U3")
  DZ(256, 8, false, true, "diffZ march blk256 U8 banded") DZ(1024, 8, false, false, "diffZ march blk1024 U8") DZ(1024, 8, true, false, "diffZ march blk1024 U8 sync")
  DZ(1024, 4, true, false, "diffZ march blk1024 U4 sync") DZ(512, 5, true, false, "diffZ march blk512 U5 sync") DZ(1024, 5, true, true, "diffZ march blk1024 U5 sync banded")
  DZ(64, 8, false, false, "diffZ march blk64 U8") DZ(256, 15, false, false, "diffZ march blk256 U15")
  { unsigned vpr = 1800; unsigned nblk = (unsigned)((nvec + 255) / 256); 
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffx_flat_b<false>), dim3(nblk), dim3(256), 0, 0, in, out, nvec, vpr, nblk); }); report("diffx flat plain", ms);
    ms = timeit([&] { hipLaunchKernelGGL((k_diffx_flat_b<true>), dim3(((nblk + 7) / 8) * 8), dim3(256), 0, 0, in, out, nvec, vpr, nblk); }); report("diffx flat banded", ms); }

#define DZO(U, LDSKB, NLEV, NAME) { unsigned nn = NLEV; size_t inner = n / nn; inner -= inner % 2; unsigned nblk = (unsigned)((inner / 2 + 255) / 256); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffz_occ<U>), dim3(nblk), dim3(256), LDSKB * 1024, 0, in, out, nn, inner); }); report(NAME, ms); }
  DZO(8, 0, 75, "diffZ occ: 75 lev, no LDS cap U8") DZO(8, 20, 75, "diffZ occ: 75 lev, 8 blk/CU U8") DZO(8, 40, 75, "diffZ occ: 75 lev, 4 blk/CU U8")
  DZO(8, 80, 75, "diffZ occ: 75 lev, 2 blk/CU U8") DZO(4, 80, 75, "diffZ occ: 75 lev, 2 blk/CU U4") DZO(8, 160, 75, "diffZ occ: 75 lev, 1 blk/CU U8")
  DZO(8, 0, 8, "diffZ occ: 8 lev x 81M") DZO(8, 0, 25, "diffZ occ: 25 lev") DZO(8, 0, 300, "diffZ occ: 300 lev x 2.16M") DZO(8, 0, 1200, "diffZ occ: 1200 lev x 0.54M")

  DZO(1, 0, 75, "diffZ occ: 75 lev, U1 full occ") DZO(2, 0, 75, "diffZ occ: 75 lev, U2 full occ") DZO(3, 0, 75, "diffZ occ: 75 lev, U3 full occ")
  DZO(1, 40, 75, "diffZ occ: 75 lev, U1 4blk/CU") DZO(2, 40, 75, "diffZ occ: 75 lev, U2 4blk/CU") DZO(2, 80, 75, "diffZ occ: 75 lev, U2 2blk/CU") DZO(4, 160, 75, "diffZ occ: 75 lev, U4 1blk/CU")
  DZO(15, 160, 75, "diffZ occ: 75 lev, U15 1blk/CU") DZO(25, 160, 75, "diffZ occ: 75 lev, U25 1blk/CU")

#define DZB(U, BLK, LDSKB, NAME) { unsigned nn = 75; size_t inner = n / nn; unsigned nblk = (unsigned)((inner / 2 + BLK - 1) / BLK); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffz_occ<U, BLK>), dim3(nblk), dim3(BLK), LDSKB * 1024, 0, in, out, nn, inner); }); report(NAME, ms); }
  DZB(2, 256, 160, "diffZ 1x256/CU U2") DZB(3, 256, 160, "diffZ 1x256/CU U3") DZB(4, 256, 160, "diffZ 1x256/CU U4") DZB(5, 256, 160, "diffZ 1x256/CU U5") DZB(6, 256, 160, "diffZ 1x256/CU U6")
  DZB(4, 512, 160, "diffZ 1x512/CU U4") DZB(2, 512, 160, "diffZ 1x512/CU U2") DZB(4, 128, 160, "diffZ 1x128/CU U4") DZB(8, 128, 160, "diffZ 1x128/CU U8") DZB(8, 64, 160, "diffZ 1x64/CU U8")
  DZB(3, 256, 80, "diffZ 2x256/CU U3") DZB(4, 256, 80, "diffZ 2x256/CU U4") DZB(5, 256, 80, "diffZ 2x256/CU U5") DZB(4, 1024, 160, "diffZ 1x1024/CU U4") DZB(2, 1024, 160, "diffZ 1x1024/CU U2")

#define DZR(RT, U, NAME) { unsigned nn = 75; size_t inner = n / nn; unsigned nw = (unsigned)((inner / 2 + 64 * RT - 1) / (64 * RT)); \
    float ms = timeit([&] { hipLaunchKernelGGL((k_diffz_rt<RT, U>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, out, nn, inner); }); report(NAME, ms); }
  DZR(1, 4, "diffZ rt: RT1 U4") DZR(2, 2, "diffZ rt: RT2 U2") DZR(2, 4, "diffZ rt: RT2 U4") DZR(4, 1, "diffZ rt: RT4 U1") DZR(4, 2, "diffZ rt: RT4 U2")
  DZR(8, 1, "diffZ rt: RT8 U1") DZR(4, 4, "diffZ rt: RT4 U4") DZR(2, 1, "diffZ rt: RT2 U1") DZR(1, 1, "diffZ rt: RT1 U1") DZR(1, 8, "diffZ rt: RT1 U8")
  return 0;
}
