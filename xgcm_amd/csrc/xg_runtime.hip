// xg_runtime.hip -- type-independent runtime helpers of the C ABI (version, errors, device memory,
// streams, events) and the synthetic-field generator.  Part of libxgcm_hip.so, see xg_common.hpp.

#include "xg_common.hpp"

#ifdef XG_PRIMARY
static thread_local char g_errbuf[XG_ERRBUF_LEN] = {0};
extern "C" __attribute__((visibility("hidden"))) char* xg_internal_errbuf(void) { return g_errbuf; }
#endif

namespace {

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_fill_synthetic(real* __restrict__ out, int64_t n, u64 seed, u64 offset,
                                                          double scale, double shift) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    u64 z = (u64)i + offset + seed * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    double uu = (double)(z >> 11) * 0x1.0p-53;
    out[i] = (real)(uu * scale + shift);  // formed in f64, rounded once (oracle: synthetic(...).astype(dtype))
  }
}

}  // namespace

extern "C" {

#ifdef XG_PRIMARY
int xg_version(void) { return XG_ABI_VERSION; }

int xg_last_error(char* buf, int n) {
  const char* g_err = xg_internal_errbuf();
  int len = (int)strlen(g_err);
  if (buf && n > 0) {
    int c = len < n - 1 ? len : n - 1;
    memcpy(buf, g_err, c);
    buf[c] = 0;
  }
  return len;
}

int xg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int xg_set_device(int device) { XG_HIP(hipSetDevice(device)); return 0; }
int xg_malloc(void** ptr, uint64_t bytes) { XG_HIP(hipMalloc(ptr, bytes)); return 0; }
int xg_free(void* ptr) { XG_HIP(hipFree(ptr)); return 0; }
int xg_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream) {
  XG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
int xg_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream) {
  XG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
int xg_stream_sync(void* stream) { XG_HIP(hipStreamSynchronize((hipStream_t)stream)); return 0; }
int xg_event_create(void** ev) { XG_HIP(hipEventCreate((hipEvent_t*)ev)); return 0; }
int xg_event_record(void* ev, void* stream) { XG_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
int xg_event_elapsed_ms(void* start, void* stop, float* ms) {
  XG_HIP(hipEventSynchronize((hipEvent_t)stop));
  XG_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}
int xg_event_destroy(void* ev) { XG_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }
#endif  // XG_PRIMARY

int XG_FN(xg_fill_synthetic)(real* out, int64_t n, uint64_t seed, uint64_t offset, double scale, double shift, void* stream) {
  if (!out && n > 0) return fail(XG_ERR_INVALID, "NULL output");
  if (n <= 0) return XG_OK;
  u64 nblocks = ((u64)n + BLOCK - 1) / BLOCK;
  if (nblocks > 256ull * 32) nblocks = 256ull * 32;  // grid-stride above 8192 blocks
  hipLaunchKernelGGL(k_fill_synthetic, dim3((u32)nblocks), dim3(BLOCK), 0, (hipStream_t)stream, out, n, (u64)seed, (u64)offset, scale, shift);
  XG_LAUNCH_CHECK();
  return XG_OK;
}


}  // extern "C"
