// tools/ubench/valu_rate.hip -- how many cycles does one wave64 VALU instruction cost per SIMD on gfx950, for the
// instruction mix of the tile rasterizer (int add, fp32 mul/add, cvt, v_pk, mul24, cndmask, LDS u64 atomic)?
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(int *out, int seed) {
  int a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 5, d = a + 7;
  float fa = (float)a, fb = (float)b, fc = 1.0001f, fd = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa = {fa, fb}, pb = {fc, fd}, pc = {0.25f, 0.75f};
  __shared__ unsigned long long lds[65 * 32];
  for (int i = threadIdx.x; i < 65 * 32; i += 256) lds[i] = 0;
  __syncthreads();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) { a += b; c += d; b += c; d += a; }                       // 4 v_add_u32
      if (KIND == 1) { fa = fa * fc; fb = fb * fc; fa = fa + fd; fb = fb + fd; } // 2 mul 2 add f32
      if (KIND == 2) { fa = (float)a; fb = (float)b; a += 1; b += 3; }         // 2 cvt 2 add
      if (KIND == 3) { pa = pa * pb; pa = pa + pc; pa = pa * pb; pa = pa + pc; } // 4 v_pk
      if (KIND == 4) { a = __mul24(a, 3) + b; c = __mul24(c, 5) + d; b ^= a; d ^= c; } // mad24 + xor
      if (KIND == 5) { a = (b > c) ? a : d; c = (a > d) ? c : b; b += 1; d += 1; }     // cmp+cndmask
      if (KIND == 6) { atomicMax(&lds[((a + u) & 31) * 65 + (threadIdx.x & 63)], (unsigned long long)b << 32 | (unsigned)c); a += 1; b += 1; c += 1; d += 1; }
      if (KIND == 7) { a = a * b + c; c = c * d + a; b += 1; d += 1; }         // v_mul_lo_u32 (full 32-bit)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (int)fa + (int)fb + (int)pa.x + (int)pa.y;
}

template <int KIND>
double run(const char *name, int vinst_per_iter, int *out) {
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 2);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = blocks * 4.0 / (256 * 4);
  const double inst_per_simd = waves_per_simd * ITER * 8.0 * vinst_per_iter;
  const double ns_per_inst = ms * 1e6 / inst_per_simd;
  printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, ms, ns_per_inst,
         ns_per_inst * 2.4);
  return ns_per_inst;
}

int main() {
  int *out;
  hipMalloc(&out, sizeof(int) * 256 * 8 * 256);
  run<0>("v_add_u32 x4", 4, out);
  run<1>("v_mul_f32/v_add_f32 x4", 4, out);
  run<2>("v_cvt_f32_i32 x2 + add x2", 4, out);
  run<3>("v_pk_mul/add_f32 x4", 4, out);
  run<4>("v_mad24 x2 + xor x2", 4, out);
  run<5>("v_cmp+v_cndmask x2 + add x2", 6, out);
  run<6>("ds_max_u64 + 4 add", 5, out);
  run<7>("v_mul_lo_u32 x2 (+2 add)", 6, out);
  return 0;
}
