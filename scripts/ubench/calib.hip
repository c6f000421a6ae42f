// Known-byte-count kernels with the library's access width (8 B per lane, 512 B per wave instruction) to calibrate
// rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: both are uncalibrated for this width).
//   calib_read8 : reads  N*8 bytes, writes ~nothing
//   calib_copy8 : reads  N*8 bytes, writes N*8 bytes
//   calib_read4 / calib_read16 / calib_read1 (round 5): the same bytes read with 4-, 16- and 1-byte loads per lane -- the
//   5-byte first pass reads 16-byte groups of low words and 4-byte groups of high bytes, the partition stores single bytes:
//   does the FETCH_SIZE scale depend on the access width?   calib_copy4 / calib_write1: WRITE_SIZE with 4- and 1-byte stores
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__global__ __launch_bounds__(256) void calib_read8(const u64 *in, u64 n, u64 *out) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) acc ^= in[i];
  if (acc == 0x1234567) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_copy8(const u64 *in, u64 n, u64 *out) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) out[i] = in[i];
}
typedef unsigned int u32;
__global__ __launch_bounds__(256) void calib_read4(const u32 *in, u64 n, u32 *out) {
  u32 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) acc ^= in[i];
  if (acc == 0x1234567) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read16(const uint4 *in, u64 n, u32 *out) {
  u32 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x1234567) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read1(const unsigned char *in, u64 n, u32 *out) {
  u32 acc = 0;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) acc ^= in[i];
  if (acc == 0x1234567) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_copy4(const u32 *in, u64 n, u32 *out) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void calib_write1(u64 n, unsigned char *out) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) out[i] = (unsigned char)i;
}
int main() {
  const u64 n = 1ull << 28;                      // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
  u64 *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 1, n * 8); hipMemset(b, 0, n * 8);
  for (int r = 0; r < 3; r++) {
    calib_read8<<<4096, 256>>>(a, n, b);
    calib_copy8<<<4096, 256>>>(a, n, b);
    calib_read4<<<4096, 256>>>((const u32 *)a, n * 2, (u32 *)b);               // the same 2 GiB
    calib_read16<<<4096, 256>>>((const uint4 *)a, n / 2, (u32 *)b);
    calib_read1<<<4096, 256>>>((const unsigned char *)a, n * 8, (u32 *)b);
    calib_copy4<<<4096, 256>>>((const u32 *)a, n * 2, (u32 *)b);
    calib_write1<<<4096, 256>>>(n * 8, (unsigned char *)b);
  }
  hipDeviceSynchronize();
  printf("calib: %llu bytes per buffer\n", n * 8);
  return 0;
}
