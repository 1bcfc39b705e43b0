/* c_abi_demo.c -- the drop-in boundary used from plain C (no Python, no torch, no C++):
 *   gcc -O2 examples/c_abi_demo.c -Iinclude -Lxgcm_amd -lxgcm_hip -Wl,-rpath,$PWD/xgcm_amd -lm -o build/c_abi_demo
 * Generates a synthetic (Z,Y,X) field in HBM, runs diff along X (periodic, center->left) and
 * interp along Y (extend) through xg_stencil1d_f64, copies the results back and checks every
 * cell against the two-line scalar definition of the reference's raw bodies
 * (xgcm/gridops.py:23-24,76-77 on a numpy.pad-ed row).  Prints the kernel time from hipEvents.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "xgcm_hip.h"

#define CHECK(call)                                                    \
  do {                                                                 \
    int rc_ = (call);                                                  \
    if (rc_ != 0) {                                                    \
      char msg[512];                                                   \
      xg_last_error(msg, sizeof msg);                                  \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, msg);              \
      return 1;                                                        \
    }                                                                  \
  } while (0)

int main(void) {
  const int64_t nz = 4, ny = 96, nx = 256;
  const int64_t shape[3] = {nz, ny, nx};
  const int64_t n = nz * ny * nx;
  if (xg_version() != XG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  if (xg_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 2; }
  void *d_in = NULL, *d_dx = NULL, *d_iy = NULL, *e0 = NULL, *e1 = NULL;
  CHECK(xg_malloc(&d_in, n * 8));
  CHECK(xg_malloc(&d_dx, n * 8));
  CHECK(xg_malloc(&d_iy, n * 8));
  CHECK(xg_event_create(&e0));
  CHECK(xg_event_create(&e1));
  CHECK(xg_fill_synthetic_f64((double*)d_in, n, 2, 0, 1.0, -0.5, NULL));
  CHECK(xg_event_record(e0, NULL));
  CHECK(xg_stencil1d_f64(XG_OP_DIFF, (const double*)d_in, (double*)d_dx, shape, 3, 2, nx, 1, 0, XG_BC_PERIODIC, 0.0,
                         NULL, NULL, NULL, NULL, NULL));
  CHECK(xg_stencil1d_f64(XG_OP_INTERP, (const double*)d_in, (double*)d_iy, shape, 3, 1, ny, 1, 0, XG_BC_EXTEND, 0.0,
                         NULL, NULL, NULL, NULL, NULL));
  CHECK(xg_event_record(e1, NULL));
  double *in = malloc(n * 8), *dx = malloc(n * 8), *iy = malloc(n * 8);
  CHECK(xg_memcpy_d2h(in, d_in, n * 8, NULL));
  CHECK(xg_memcpy_d2h(dx, d_dx, n * 8, NULL));
  CHECK(xg_memcpy_d2h(iy, d_iy, n * 8, NULL));
  CHECK(xg_stream_sync(NULL));
  float ms = 0;
  CHECK(xg_event_elapsed_ms(e0, e1, &ms));
  long bad = 0;
  for (int64_t z = 0; z < nz; ++z)
    for (int64_t y = 0; y < ny; ++y)
      for (int64_t x = 0; x < nx; ++x) {
        const int64_t c = (z * ny + y) * nx + x;
        const double left = in[(z * ny + y) * nx + (x == 0 ? nx - 1 : x - 1)];  /* wrap */
        const double below = in[(z * ny + (y == 0 ? 0 : y - 1)) * nx + x];      /* edge */
        if (dx[c] != in[c] - left) ++bad;
        if (iy[c] != (below + in[c]) / 2.0) ++bad;
      }
  /* an invalid request is reported, not executed */
  int rc = xg_stencil1d_f64(XG_OP_DIFF, (const double*)d_in, (double*)d_dx, shape, 3, 2, nx, 1, 0, XG_BC_NONE, 0.0,
                            NULL, NULL, NULL, NULL, NULL);
  char msg[256];
  xg_last_error(msg, sizeof msg);
  printf("c_abi_demo: %ld mismatching cells of %ld, two launches %.3f ms, bad-call status %d (%s)\n", bad,
         (long)(2 * n), ms, rc, msg);
  xg_free(d_in); xg_free(d_dx); xg_free(d_iy); xg_event_destroy(e0); xg_event_destroy(e1);
  free(in); free(dx); free(iy);
  return (bad == 0 && rc == XG_ERR_INVALID) ? 0 : 3;
}
