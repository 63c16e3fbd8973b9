// tools/ubench/ext_events.hip -- what the start / stop events of hipExtLaunchKernelGGL measure on this runtime: two dependent kernels
// of ~200 us each on one stream; elapsed times between every pair of events, next to events recorded between the launches.
// hipcc --offload-arch=gfx950 -O2 -o ext_events tools/ubench/ext_events.hip && ./ext_events
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(unsigned long long cycles, unsigned long long *out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long t = t0;
  while (t - t0 < cycles) t = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = t - t0;
}
static float el(hipEvent_t a, hipEvent_t b, const char *what) {
  float ms = -1.f;
  hipError_t rc = hipEventElapsedTime(&ms, a, b);
  if (rc != hipSuccess) { printf("%-28s error: %s\n", what, hipGetErrorString(rc)); (void)hipGetLastError(); return -1.f; }
  printf("%-28s %8.1f us\n", what, ms * 1e3);
  return ms;
}
int main() {
  unsigned long long *d;
  hipMalloc(&d, 8);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t s1, e1, s2, e2, r0, r1, r2;
  for (hipEvent_t *e : {&s1, &e1, &s2, &e2, &r0, &r1, &r2}) hipEventCreate(e);
  const unsigned long long cyc = 20000;  // s_memtime ticks at 100 MHz: 200 us
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, cyc, d);  // something in front
    hipEventRecord(r0, s);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, s1, e1, 0, cyc, d);
    hipEventRecord(r1, s);
    hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, s2, e2, 0, cyc, d);
    hipEventRecord(r2, s);
    hipStreamSynchronize(s);
    printf("--- rep %d (each kernel spins 200 us)\n", rep);
    el(r0, r1, "recorded r0 -> r1 (K1)");
    el(r1, r2, "recorded r1 -> r2 (K2)");
    el(s1, e1, "ext start1 -> stop1");
    el(s2, e2, "ext start2 -> stop2");
    el(s1, e2, "ext start1 -> stop2");
    el(e1, e2, "ext stop1 -> stop2");
    el(s1, s2, "ext start1 -> start2");
    el(e1, s2, "ext stop1 -> start2");
    el(r0, e1, "recorded r0 -> ext stop1");
    el(r0, e2, "recorded r0 -> ext stop2");
    el(e1, r2, "ext stop1 -> recorded r2");
  }
  // a chain like the library's: init (short), set-up (long), clip (short), tile kernel (long), back to back, no records between
  hipEvent_t e[4], w0, w1;
  for (auto &x : e) hipEventCreate(&x);
  hipEventCreate(&w0); hipEventCreate(&w1);
  const unsigned long long len[4] = {8000, 500000, 8000, 1300000};  // ~4, 250, 4, 650 us
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 1300000ull, d);  // the previous call's tile kernel
    hipStreamSynchronize(s);
    const auto h0 = std::chrono::steady_clock::now();
    hipEventRecord(w0, s);
    for (int k = 0; k < 4; ++k) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, (hipEvent_t) nullptr, e[k], 0, len[k], d);
    hipEventRecord(w1, s);
    hipStreamSynchronize(s);
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    printf("--- chain rep %d: kernels of ~4 / 250 / 4 / 650 us; host clock around it (sync to sync): %.1f us\n", rep, host_us);
    el(w0, w1, "recorded around the chain");
    el(e[0], e[1], "stop0 -> stop1 (250)");
    el(e[1], e[2], "stop1 -> stop2 (4)");
    el(e[2], e[3], "stop2 -> stop3 (650)");
    el(e[0], e[3], "stop0 -> stop3 (904)");
    el(w0, e[3], "recorded w0 -> stop3");
  }
  // the library's mix: init (stop event), cull + set-up (PLAIN launches, no events), clip (stop event), tile kernel (stop event)
  for (int rep = 0; rep < 3; ++rep) {
    hipStreamSynchronize(s);
    const auto h0 = std::chrono::steady_clock::now();
    for (int call = 0; call < 10; ++call) {
      hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, (hipEvent_t) nullptr, e[0], 0, 8000ull, d);
      hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 8000ull, d);
      hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, 500000ull, d);
      hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, (hipEvent_t) nullptr, e[1], 0, 8000ull, d);
      hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s, (hipEvent_t) nullptr, e[2], 0, 1300000ull, d);
    }
    hipStreamSynchronize(s);
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    printf("--- mixed chain rep %d, 10 calls back to back (events of the LAST call); host clock per call: %.1f us\n", rep, host_us / 10);
    el(e[0], e[1], "stop init -> stop clip");
    el(e[1], e[2], "stop clip -> stop tile");
  }
  return 0;
}
