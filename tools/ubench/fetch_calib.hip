// tools/ubench/fetch_calib.hip -- what does rocprofv3's FETCH_SIZE / WRITE_SIZE count for the access shapes of k_vote_labels?
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reads HALF the bytes of a wide (16 B per lane) coalesced streaming read on gfx950 -- "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Four kernels over buffers far larger
// than the Infinity Cache (1 GiB each, touched once), each with a byte count that is known exactly:
//   k_read16   16 B per lane, coalesced            (the shape the guide calibrated: expect FETCH_SIZE = bytes / 2)
//   k_read4     4 B per lane, coalesced            (k_vote_labels' winner reads, votes / counts rows)
//   k_gather1   1 B per lane at a stride of 52 B   (its label gathers: one byte per visible face, neighbours 13 pixels apart)
//   k_write4    4 B per lane, every other dword    (its winner resets: scattered dword stores)
// hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib;  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read16(const uint4 *p, size_t n, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_read4(const unsigned *p, size_t n, unsigned *out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_gather1(const unsigned char *p, size_t n, unsigned *out) {   // n gathers, 52 bytes apart
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i * 52];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write4(unsigned *p, size_t n) {                             // every other dword
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[2 * i] = 0u;
}

int main() {
  const size_t bytes = 1ull << 30;
  void *buf; unsigned *out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
  void *spoil; CK(hipMalloc(&spoil, bytes));   // a second gigabyte written between the kernels: nothing of `buf` stays cached
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(spoil, rep, bytes)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, out); CK(hipDeviceSynchronize());
    CK(hipMemset(spoil, rep + 2, bytes)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read4, dim3(4096), dim3(256), 0, 0, (const unsigned *)buf, bytes / 4, out); CK(hipDeviceSynchronize());
    CK(hipMemset(spoil, rep + 4, bytes)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_gather1, dim3(4096), dim3(256), 0, 0, (const unsigned char *)buf, bytes / 52, out); CK(hipDeviceSynchronize());
    CK(hipMemset(spoil, rep + 6, bytes)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write4, dim3(4096), dim3(256), 0, 0, (unsigned *)buf, bytes / 8); CK(hipDeviceSynchronize());
  }
  printf("bytes touched: k_read16 %zu, k_read4 %zu, k_gather1 %zu gathers (1 B each; %zu B at 32-B sectors, %zu at 64), k_write4 %zu B stored\n",
         bytes, bytes, bytes / 52, (bytes / 52) * 32, (bytes / 52) * 64, bytes / 2);
  return 0;
}
