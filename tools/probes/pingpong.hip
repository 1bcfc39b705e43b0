// pingpong.hip -- one-way latency of a flag hand-off between two workgroups (same XCD / neighbouring XCDs) by cache-policy
// flavour of the polling load and the publishing store.  Decides how the chained scan (xg_scan.hip, K5c) passes its
// running sums.  Every poll loop gives up after 2^20 tries (reported as GAVE UP): a flavour that reads a stale L1 line
// must not hang the box.  Tuning aid, not part of the product.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/pingpong.hip -o build/pingpong && build/pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int L> __device__ __forceinline__ unsigned poll(unsigned* p) {
  unsigned v, z = 0;
  if (L == 0) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("global_load_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) asm volatile("global_atomic_or %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
  if (L == 4) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 5) asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 6) asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int S> __device__ __forceinline__ void publish(unsigned* p, unsigned v) {
  if (S == 0) asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
  if (S == 1) asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
  if (S == 2) asm volatile("global_store_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
  if (S == 3) asm volatile("global_atomic_swap %0, %1, off\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
  if (S == 4) asm volatile("global_store_dword %0, %1, off nt\n s_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

template <int L, int S>
__global__ void k_pp(unsigned* flags, unsigned rounds, unsigned partner, unsigned* gaveup, unsigned* xcc) {
  if (threadIdx.x != 0) return;
  const bool a = blockIdx.x == 0, b = blockIdx.x == partner;
  if (!a && !b) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[a ? 0 : 1] = id & 0xf;
  unsigned* fa = flags;       // written by A, polled by B
  unsigned* fb = flags + 64;  // another cache line: written by B, polled by A
  for (unsigned i = 1; i <= rounds; ++i) {
    if (a) publish<S>(fa, i);
    unsigned tries = 0;
    unsigned* w = a ? fb : fa;
    while (poll<L>(w) < i) { if (++tries > (1u << 20)) { *gaveup = 1; return; } }
    if (b) publish<S>(fb, i);
  }
}

int main() {
  unsigned *flags, *gave, *xcc;
  CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&gave, 4)); CK(hipMalloc(&xcc, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned rounds = 2000;
#define RUN(L, S, PARTNER) { CK(hipMemset(flags, 0, 4096)); CK(hipMemset(gave, 0, 4)); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k_pp<L, S>), dim3(16), dim3(64), 0, 0, flags, rounds, PARTNER, gave, xcc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); unsigned g, x[2]; CK(hipMemcpy(&g, gave, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost)); \
    printf("load flavour %d, store flavour %d, blocks 0 and %2d (XCC %u / %u): %s one-way %.3f us\n", L, S, PARTNER, x[0], x[1], g ? "GAVE UP" : "ok", ms * 1000.0 / rounds / 2); fflush(stdout); }
  printf("# load flavours: 0 sc1, 1 sc0, 2 nt, 3 atomic_or(0) returning, 4 sc0 sc1, 5 plain, 6 buffer_inv sc0 + plain;  store: 0 sc1, 1 plain, 2 sc0, 3 atomic_swap, 4 nt\n");
  RUN(0, 0, 8) RUN(0, 1, 8) RUN(3, 1, 8) RUN(3, 3, 8) RUN(3, 0, 8) RUN(1, 1, 8) RUN(2, 1, 8) RUN(4, 0, 8) RUN(6, 1, 8) RUN(5, 1, 8) RUN(1, 2, 8) RUN(2, 4, 8)
  RUN(0, 0, 1) RUN(3, 3, 1) RUN(3, 1, 1) RUN(0, 1, 1)
  return 0;
}
