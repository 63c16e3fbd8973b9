// tools/ubench/store_pattern.hip -- how fast can the id images (50 x 3000 x 4000 int32 = 2.4 GB) be written, by store
// shape and tile order?  The tile kernel's memory floor.  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int H = 3000, W = 4000, NV = 50;

__global__ __launch_bounds__(256) void k_linear(int4 *p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_int4(-1, -1, -1, -1);
}

// tile TWxTH per workgroup of 256 threads; VEC = dwords per lane per store; XCD = remap tile index so that blocks b, b+8, ... (one XCD) take consecutive tiles
template <int TW, int TH, int VEC, int XCD>
__global__ __launch_bounds__(256) void k_tile(int *ids, int TX, int T) {
  int tile = blockIdx.x;
  if (XCD) {  // block b runs on XCD b % 8 (observed): give XCD x the tiles [x*T/8, (x+1)*T/8)
    const int per = (T + 7) / 8;
    tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (tile >= T) return;
  }
  const int tx = tile % TX, ty = tile / TX;
  const int px0 = tx * TW, py0 = ty * TH;
  int *plane = ids + (size_t)blockIdx.y * H * W;
  constexpr int LPR = TW / VEC;        // lanes per row
  constexpr int RPP = 256 / LPR;       // rows per pass
  const int c = (threadIdx.x % LPR) * VEC, r0 = threadIdx.x / LPR;
  if (px0 + c >= W) return;
  for (int r = r0; r < TH && py0 + r < H; r += RPP) {
    int *dst = plane + (size_t)(py0 + r) * W + px0 + c;
    if (VEC == 4) *reinterpret_cast<int4 *>(dst) = make_int4(-1, -1, -1, -1);
    else if (VEC == 2) *reinterpret_cast<int2 *>(dst) = make_int2(-1, -1);
    else *dst = -1;
  }
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

template <int TW, int TH, int VEC, int XCD> void run(int *ids, const char *name) {
  const int TX = (W + TW - 1) / TW, TY = (H + TH - 1) / TH, T = TX * TY;
  const int gx = XCD ? ((T + 7) / 8) * 8 : T;
  float ms = timeit([&] { hipLaunchKernelGGL((k_tile<TW, TH, VEC, XCD>), dim3(gx, NV), dim3(256), 0, 0, ids, TX, T); });
  printf("%-28s %7.3f ms  %6.2f us/view  %5.2f TB/s\n", name, ms, ms * 1e3 / NV, (double)NV * H * W * 4 / (ms * 1e-3) / 1e12);
}

int main() {
  int *ids; const size_t n = (size_t)NV * H * W;
  CK(hipMalloc(&ids, n * 4));
  float ms = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(8192), dim3(256), 0, 0, (int4 *)ids, n / 4); });
  printf("%-28s %7.3f ms  %6.2f us/view  %5.2f TB/s\n", "linear 16B", ms, ms * 1e3 / NV, (double)n * 4 / (ms * 1e-3) / 1e12);
  ms = timeit([&] { CK(hipMemsetAsync(ids, 0xFF, n * 4, 0)); });
  printf("%-28s %7.3f ms  %6.2f us/view  %5.2f TB/s\n", "hipMemset", ms, ms * 1e3 / NV, (double)n * 4 / (ms * 1e-3) / 1e12);
  run<64, 32, 1, 0>(ids, "tile 64x32 dword");
  run<64, 32, 4, 0>(ids, "tile 64x32 dwordx4");
  run<64, 32, 4, 1>(ids, "tile 64x32 dwordx4 xcd");
  run<64, 32, 1, 1>(ids, "tile 64x32 dword xcd");
  run<128, 16, 4, 0>(ids, "tile 128x16 dwordx4");
  run<128, 16, 4, 1>(ids, "tile 128x16 dwordx4 xcd");
  run<256, 8, 4, 0>(ids, "tile 256x8 dwordx4");
  run<256, 8, 4, 1>(ids, "tile 256x8 dwordx4 xcd");
  run<64, 64, 4, 0>(ids, "tile 64x64 dwordx4");
  run<32, 32, 4, 0>(ids, "tile 32x32 dwordx4");
  run<1024, 2, 4, 0>(ids, "tile 1024x2 dwordx4");
  return 0;
}
