// per-CU output store throughput: segment size per row x store policy x number of active CUs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WT> __device__ __forceinline__ void st16(f32x4* p, f32x4 v) {
  if (WT == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if (WT == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
}
// each block (512 threads = 8 waves) writes a 256-row x ROWB-byte tile of a row-major matrix with leading dimension ld bytes;
// a wave-instruction covers (1024 / SEG) rows x SEG bytes; the tile is swept so that every byte is written once
template <int SEG, int WT>
__global__ void __launch_bounds__(512) k_store(char* out, long ld, int rowb, int rows, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  char* base = out + (long)tile_m * rows * ld + (long)tile_n * rowb;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  constexpr int LPR = SEG / 16;            // lanes per row segment
  constexpr int RPI = 64 / LPR;            // rows per instruction
  const int segs = rowb / SEG;             // segments per row
  // wave w takes rows [w * rows/8, (w+1) * rows/8)
  const int r0 = wave * (rows / 8);
  for (int r = 0; r < rows / 8; r += RPI)
    for (int s = 0; s < segs; ++s)
      st16<WT>((f32x4*)(base + (long)(r0 + r + lane / LPR) * ld + s * SEG + (lane % LPR) * 16), v);
}
template <int SEG, int WT> float run(char* buf, long ld, int rowb, int rows, int nblk, int tiles_n, hipStream_t st) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k_store<SEG, WT><<<nblk, 512, 0, st>>>(buf, ld, rowb, rows, tiles_n);
  hipEventRecord(a, st);
  for (int i = 0; i < 20; ++i) k_store<SEG, WT><<<nblk, 512, 0, st>>>(buf, ld, rowb, rows, tiles_n);
  hipEventRecord(b, st); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1e3;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const long ld = 4096;    // bytes per matrix row (2048 bf16)
  char* buf; CK(hipMalloc(&buf, 4096L * ld * 2));
  // tile 256 rows x 512 B (= 128 KB per block, the dgrad epilogue's share); 128 blocks = half the matrix, 256 = all
  for (int nblk : {128, 256}) {
    const int tiles_n = 8;
    printf("%d blocks x 128 KB (%5.1f MB):\n", nblk, nblk * 0.131072);
#define ROW(SEG) printf("  %4d-B segments: plain %6.1f us | write-through %6.1f us | nt %6.1f us\n", SEG, \
    run<SEG, 0>(buf, ld, 512, 256, nblk, tiles_n, st), run<SEG, 1>(buf, ld, 512, 256, nblk, tiles_n, st), run<SEG, 2>(buf, ld, 512, 256, nblk, tiles_n, st));
    ROW(64) ROW(128) ROW(256) ROW(512)
  }
  // an empty-ish kernel for the launch floor
  printf("floor (1 store per lane): %.1f us\n", run<64, 0>(buf, ld, 64, 128, 256, 8, st));
  return 0;
}
