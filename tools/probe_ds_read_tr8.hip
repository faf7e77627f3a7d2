// probe: semantics of ds_read_b64_tr_b8 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(unsigned char* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 64];
  // byte at [row][col] of a 64 x 64 byte image = row * 64 + col encoded as (row << 4 | col & 15) ... keep it simple: row*16+ (col&15) won't fit.
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned char)(((i / 64) & 15) << 4 | ((i % 64) & 15));   // hi nibble = row & 15, lo nibble = col & 15
  __syncthreads();
  const int lane = threadIdx.x;
  int addr;
  if (mode == 0) addr = (lane & 15) * 64;                        // every lane of a 16-group: its own row, col 0..7
  else if (mode == 1) addr = ((lane & 15) >> 1) * 64 + (lane & 1) * 8;   // 8 rows x 16 cols block: lane i -> row i>>1, cols 8(i&1)..
  else addr = (lane & 7) * 64 + ((lane & 15) >> 3) * 8;          // lane i -> row i&7, cols 8(i>>3)..
  addr += (lane >> 4) * 16 * 64 * 0 + (lane >> 4) * 16;          // lane groups: next 16 columns
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + addr));
  unsigned char* o = out + lane * 8;
  for (int j = 0; j < 8; ++j) o[j] = (unsigned char)((j < 4 ? (unsigned)r[0] >> (8 * j) : (unsigned)r[1] >> (8 * (j - 4))) & 0xff);
}
int main() {
  unsigned char* d; hipMalloc(&d, 512);
  unsigned char h[512];
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64>>>(d, mode); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d (each entry: row,col nibbles of the 8 bytes lane gets)\n", mode);
    for (int l = 0; l < 20; ++l) { printf(" lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %x,%x", h[l * 8 + j] >> 4, h[l * 8 + j] & 15); printf("\n"); }
  }
  return 0;
}
