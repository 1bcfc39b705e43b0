// xcc_probe: which XCD (XCC_ID hardware register) does workgroup b of a 1-D launch land on?
// The kernels' "banding" assumes b % 8 (round-robin dispatch, SPX mode); this prints the observed map.
//   hipcc --offload-arch=gfx950 tools/probes/xcc_probe.hip -o build/xcc_probe && build/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_probe(unsigned* xcc, unsigned* cu) {
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits [3:0]
    cu[blockIdx.x] = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (3 << 11));     // HW_REG_HW_ID, CU_ID bits [11:8]
  }
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 8192;
  unsigned *dx, *dc, *hx = (unsigned*)malloc(nb * 4), *hc = (unsigned*)malloc(nb * 4);
  hipMalloc(&dx, nb * 4);
  hipMalloc(&dc, nb * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, dx, dc);
    hipMemcpy(hx, dx, nb * 4, hipMemcpyDeviceToHost);
    long hist[8][16] = {{0}};
    int mismatch = 0;
    for (int b = 0; b < nb; ++b) {
      hist[b & 7][hx[b] & 15]++;
      if ((hx[b] & 15) != (unsigned)(b & 7)) ++mismatch;
    }
    printf("launch %d: %d blocks, %d with XCC_ID != blockIdx %% 8\n", rep, nb, mismatch);
    for (int r = 0; r < 8; ++r) {
      printf("  b%%8=%d ->", r);
      for (int x = 0; x < 16; ++x) if (hist[r][x]) printf(" xcc%d:%ld", x, hist[r][x]);
      printf("\n");
    }
  }
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device: %s, %d CUs, L2 %d KiB, memory bus %d bit, clock %d kHz\n", p.name, p.multiProcessorCount, p.l2CacheSize / 1024,
         p.memoryBusWidth, p.memoryClockRate);
  return 0;
}
