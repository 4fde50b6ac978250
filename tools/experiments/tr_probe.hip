// What does ds_read_b64_tr_b16 return?  LDS holds a [16 rows][64 cols] bf16-sized (16-bit) matrix with value = row * 64 + col;
// lane l supplies the address of the 8-byte piece (row = 4 * (l >> 4 ... ) see below) and prints its four 16-bit results.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tr_probe tools/experiments/tr_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned short* out) {
  __shared__ unsigned short m[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) m[i] = (unsigned short)i;     // value = row * 64 + col
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, t = l & 15;
  // group g reads the 4-row x 16-col block at rows 4 * (g >> 1) .. +3, cols 16 * (g & 1) .. +15: lane t supplies the address of
  // its 8-byte piece: row (t >> 2), cols 4 * (t & 3) .. +3 of that block
  const int row = 4 * (g >> 1) + (t >> 2), col = 16 * (g & 1) + 4 * (t & 3);
  const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned short*)(m + row * 64 + col);
  unsigned v0, v1;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
  v0 = r[0]; v1 = r[1];
  out[l * 4 + 0] = v0 & 0xffff; out[l * 4 + 1] = v0 >> 16; out[l * 4 + 2] = v1 & 0xffff; out[l * 4 + 3] = v1 >> 16;
}

int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (r%2d,c%2d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
    printf("\n");
  }
  return 0;
}
