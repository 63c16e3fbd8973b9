// tools/ubench/atomic_rate.hip -- what does an integer global atomicAdd cost on gfx950 when many waves hit the SAME
// address, the same 128-byte line, or addresses spread over many lines -- returning and non-returning?  This is the
// question behind the set-up kernel of the hostile workload (k_setup_cull: one big-list counter per view, hot tile
// counters).  hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// every wave issues ITER atomics from ONE lane (the leader of a wave-aggregated counter), each dependent on the last when
// RET (the next address is computed from the returned value, as a list position would be)
template <bool RET>
__global__ __launch_bounds__(256) void k(unsigned *ctr, int n_addr, int stride_words, int iters, unsigned *sink) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
    const unsigned a = (unsigned)((wave * 7 + i * 13 + (RET ? (acc & 0u) : 0u)) % n_addr);
    if (lane == 0) {
      if (RET) acc += atomicAdd(&ctr[(size_t)a * stride_words], 1u);
      else atomicAdd(&ctr[(size_t)a * stride_words], 1u);
    }
    acc = __shfl(acc, 0);
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <bool RET>
void run(const char *name, unsigned *ctr, unsigned *sink, int n_addr, int stride_words, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(ctr, 0, 64u << 20);
  hipLaunchKernelGGL(k<RET>, dim3(blocks), dim3(256), 0, 0, ctr, n_addr, stride_words, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<RET>, dim3(blocks), dim3(256), 0, 0, ctr, n_addr, stride_words, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)blocks * 4 * iters;
  printf("%-44s addr %6d stride %3d B  waves %6d x %4d : %8.3f ms  %8.2f ns per atomic (chip), %8.1f ns per atomic per address\n",
         name, n_addr, stride_words * 4, blocks * 4, iters, ms, ms * 1e6 / n, ms * 1e6 / (n / n_addr));
}

int main() {
  unsigned *ctr, *sink;
  hipMalloc(&ctr, 64u << 20);
  hipMalloc(&sink, 64);
  const int blocks = 2048, iters = 64;
  for (int ret = 0; ret < 2; ++ret) {
    const char *r = ret ? "returning" : "no return";
    auto go = [&](const char *what, int n_addr, int stride) {
      char name[96];
      snprintf(name, sizeof(name), "%s, %s", what, r);
      if (ret) run<true>(name, ctr, sink, n_addr, stride, blocks, iters);
      else run<false>(name, ctr, sink, n_addr, stride, blocks, iters);
    };
    go("one address", 1, 1);
    go("20 addresses, one per 1 KiB (20 views)", 20, 256);
    go("32 addresses in one 128-B line", 32, 1);
    go("32 addresses, one per 128-B line", 32, 32);
    go("640 addresses in 20 lines", 640, 1);
    go("6000 addresses, packed (tile counters)", 6000, 1);
    go("6000 addresses, one per 64 B", 6000, 16);
    go("120000 addresses, packed", 120000, 1);
  }
  return 0;
}
