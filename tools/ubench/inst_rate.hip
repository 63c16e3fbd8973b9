// tools/ubench/inst_rate.hip -- issue cost of single VALU instructions on gfx950 (GPU box only):
//   hipcc --offload-arch=gfx950 -O2 -o inst_rate tools/ubench/inst_rate.hip && ./inst_rate
// Every case is one instruction in inline asm, repeated 64 times per loop trip on 8 independent register sets (no dependent
// chain shorter than 8 instructions), 8 waves per SIMD on every SIMD of the chip, so the time per instruction is its ISSUE
// cost, not its latency.  Printed: SIMD cycles per wave64 instruction at the clock measured in the kernel (s_memtime over
// s_memrealtime).  What it is for: the set-up kernel's instruction budget prices 64-bit integer
// multiply-adds, double-precision arithmetic and reciprocals at what they cost, not at one slot each.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int CASE>
__global__ __launch_bounds__(256) void k_rate(int trips, unsigned long long *out, unsigned *sink) {
  unsigned a[8], b[8];
  unsigned long long w[8];
  double d[8], e[8];
  float f[8];
  unsigned sg[8];
  unsigned long long sm[8];
  typedef int v4 __attribute__((ext_vector_type(4)));
  v4 q4[8];
  __shared__ int lds_buf[4096];
  lds_buf[threadIdx.x] = threadIdx.x;
  const unsigned lds_addr = (threadIdx.x & 255) * 16;
  const float sf = 1.0000001f;
  for (int i = 0; i < 8; ++i) {
    sg[i] = 0; sm[i] = 0; q4[i] = v4{0, 0, 0, 0};
    a[i] = threadIdx.x * 2654435761u + i * 40503u + 1u;
    b[i] = threadIdx.x * 40503u + i + 3u;
    w[i] = ((unsigned long long)a[i] << 32) | b[i];
    d[i] = 1.0 + 1e-9 * (double)a[i];
    e[i] = 1.0 + 1e-12 * (double)b[i];
    f[i] = 1.0f + 1e-6f * (float)(b[i] & 1023u);
  }
  const unsigned long long mask = __ballot((threadIdx.x & 3) != 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (CASE == 0) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 1) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 2) {
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 3) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 4) {
#define X(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 5) {
#define X(i) asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 6) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e[i]));
        REP8(X)
#undef X
      } else if (CASE == 7) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
        REP8(X)
#undef X
      } else if (CASE == 8) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
        REP8(X)
#undef X
      } else if (CASE == 9) {
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        REP8(X)
#undef X
      } else if (CASE == 10) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 11) {
#define X(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 12) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(mask));
        REP8(X)
#undef X
      } else if (CASE == 13) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 14) {
#define X(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(d[i]) : "v"(e[i]) : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 15) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(w[i]));
        REP8(X)
#undef X
      } else if (CASE == 16) {
#define X(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 17) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 18) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
        REP8(X)
#undef X
      } else if (CASE == 19) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 20) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 21) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 22) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 23) {
#define X(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 24) {
#define X(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 25) {
#define X(i) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 26) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 27) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 28) {
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(b[i]), "+v"(f[i]) : : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 29) {
#define X(i) asm volatile("v_ashrrev_i64 %0, 8, %0" : "+v"(w[i]));
        REP8(X)
#undef X
      } else if (CASE == 30) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[i]));
        REP8(X)
#undef X
      } else if (CASE == 31) {
#define X(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 32) {
#define X(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 33) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 34) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 35) {
#define X(i) asm volatile("v_bfe_i32 %0, %0, 3, 12" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 36) {
#define X(i) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 37) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(f[i]) : "s"(sf));
        REP8(X)
#undef X
      } else if (CASE == 38) {
#define X(i) asm volatile("v_floor_f32 %0, %0\n\tv_cvt_i32_f32 %1, %0" : "+v"(f[i]), "=v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 39) {
#define X(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(f[i]), "v"(b[i]) : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 40) {
#define X(i) asm volatile("v_div_fixup_f32 %0, %0, %0, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 41) {
#define X(i) asm volatile("v_min3_i32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 42) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 43) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 44) {
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 45) {
#define X(i) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 50) {
#define X(i) asm volatile("v_add_u32 %0, 7, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 51) {
#define X(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 52) {
#define X(i) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 53) {
#define X(i) asm volatile("v_mul_f32 %0, 0x43800000, %0" : "+v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 54) {
#define X(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i]) : "s"(sf));
        REP8(X)
#undef X
      } else if (CASE == 55) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(sf));
        REP8(X)
#undef X
      } else if (CASE == 56) {
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 57) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 58) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 59) {
#define X(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 60) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i+1)&7]));
        REP8(X)
#undef X
      } else if (CASE == 61) {
#define X(i) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(sm[i]) : "v"(a[i]), "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 62) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
        REP8(X)
#undef X
      } else if (CASE == 63) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 64) {
#define X(i) asm volatile("v_ashrrev_i32 %0, 8, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 65) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 66) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i+1)&7]));
        REP8(X)
#undef X
      } else if (CASE == 67) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i+1)&7]));
        REP8(X)
#undef X
      } else if (CASE == 68) {
#define X(i) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(X)
#undef X
      } else if (CASE == 69) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(f[i]) : "v"(f[(i+1)&7]));
        REP8(X)
#undef X
      } else if (CASE == 70) {
#define X(i) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (CASE == 71) {
#define X(i) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(f[i]) : "v"(f[(i+1)&7]));
        REP8(X)
#undef X
      } else if (CASE == 72) {
#define X(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
        REP8(X)
#undef X
      } else if (CASE == 73) {
#define X(i) asm volatile("v_mul_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(f[(i+1)&7]));
        REP8(X)
#undef X
      } else if (CASE == 74) {
#define X(i) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q4[i]) : "v"(lds_addr));
        REP8(X)
#undef X
      } else if (CASE == 75) {
#define X(i) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w[i]) : "v"(lds_addr));
        REP8(X)
#undef X
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  unsigned acc = 0;
  for (int i = 0; i < 8; ++i) acc ^= (unsigned)sm[i] ^ (unsigned)q4[i].x ^ (unsigned)q4[i].w ^ sg[i] ^ a[i] ^ b[i] ^ (unsigned)w[i] ^ (unsigned)(w[i] >> 32) ^ (unsigned)d[i] ^ (unsigned)e[i] ^ (unsigned)f[i];
  if (acc == 0x12345679u) sink[0] = acc;
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], t1 - t0);
    atomicAdd(&out[1], r1 - r0);
    atomicAdd(&out[2], 1ull);
  }
}

template <int CASE>
void run(const char *name, int cus, unsigned long long *out_d, unsigned *sink_d) {
  const int trips = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // 16 workgroups of 256 threads per CU (whatever the occupancy, every SIMD has several waves to issue from)
  hipLaunchKernelGGL(k_rate<CASE>, dim3(cus * 16), dim3(256), 0, 0, 10, out_d, sink_d);
  hipMemset(out_d, 0, 3 * sizeof(unsigned long long));
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_rate<CASE>, dim3(cus * 16), dim3(256), 0, 0, trips, out_d, sink_d);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long o[3];
  hipMemcpy(o, out_d, sizeof(o), hipMemcpyDeviceToHost);
  const double clk_ghz = (double)o[0] / ((double)o[1] * 10.0);  // shader cycles per ns while the loop ran (s_memrealtime ticks are 10 ns)
  const double simd_cycles = (double)ms * 1e6 * clk_ghz;        // cycles every SIMD had during the launch
  const double inst_per_simd = (double)cus * 16.0 * 4.0 * (double)trips * 64.0 / ((double)cus * 4.0);
  printf("%-22s %6.2f SIMD cycles per wave64 instruction   (clock %.2f GHz, %.3f ms)\n", name, simd_cycles / inst_per_simd, clk_ghz, ms);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  unsigned long long *out_d;
  unsigned *sink_d;
  hipMalloc(&out_d, 64);
  hipMalloc(&sink_d, 64);
  const int cus = p.multiProcessorCount;
  run<0>("v_add_u32", cus, out_d, sink_d);
  run<19>("v_mad_u32_u24", cus, out_d, sink_d);
  run<4>("v_mul_i32_i24", cus, out_d, sink_d);
  run<5>("v_mul_hi_i32_i24", cus, out_d, sink_d);
  run<3>("v_mul_lo_u32", cus, out_d, sink_d);
  run<1>("v_mad_u64_u32", cus, out_d, sink_d);
  run<2>("v_mad_i64_i32", cus, out_d, sink_d);
  run<15>("v_lshl_add_u64", cus, out_d, sink_d);
  run<12>("v_cndmask_b32", cus, out_d, sink_d);
  run<17>("v_mov_b32_dpp", cus, out_d, sink_d);
  run<13>("v_mul_f32", cus, out_d, sink_d);
  run<10>("v_rcp_f32", cus, out_d, sink_d);
  run<6>("v_fma_f64", cus, out_d, sink_d);
  run<7>("v_mul_f64", cus, out_d, sink_d);
  run<8>("v_add_f64", cus, out_d, sink_d);
  run<9>("v_rcp_f64", cus, out_d, sink_d);
  run<14>("v_div_scale_f64", cus, out_d, sink_d);
  run<11>("v_cvt_f64_i32", cus, out_d, sink_d);
  run<18>("v_cvt_f32_f64", cus, out_d, sink_d);
  run<20>("v_mov_b32", cus, out_d, sink_d);
  run<21>("v_and_b32", cus, out_d, sink_d);
  run<22>("v_lshlrev_b32", cus, out_d, sink_d);
  run<23>("v_max_i32", cus, out_d, sink_d);
  run<24>("v_sub_u32", cus, out_d, sink_d);
  run<25>("v_cmp_lt_i32 (vcc)", cus, out_d, sink_d);
  run<26>("v_fma_f32", cus, out_d, sink_d);
  run<27>("v_cvt_f32_i32", cus, out_d, sink_d);
  run<28>("v_add_co_u32+addc (2 instructions)", cus, out_d, sink_d);
  run<29>("v_ashrrev_i64", cus, out_d, sink_d);
  run<30>("v_pk_mul_f32", cus, out_d, sink_d);
  run<31>("v_add_f32", cus, out_d, sink_d);
  run<32>("v_mad_i32_i24", cus, out_d, sink_d);
  run<33>("v_readlane_b32", cus, out_d, sink_d);
  run<34>("v_and_or_b32", cus, out_d, sink_d);
  run<35>("v_bfe_i32", cus, out_d, sink_d);
  run<36>("v_mbcnt_lo+hi (2 instructions)", cus, out_d, sink_d);
  run<37>("v_mul_f32 sgpr operand", cus, out_d, sink_d);
  run<38>("v_cvt_flr/floor+cvt (2 instructions)", cus, out_d, sink_d);
  run<39>("v_cmp_class_f32", cus, out_d, sink_d);
  run<40>("v_div_fixup_f32", cus, out_d, sink_d);
  run<41>("v_min3_i32", cus, out_d, sink_d);
  run<42>("v_add3_u32", cus, out_d, sink_d);
  run<43>("v_lshl_add_u32", cus, out_d, sink_d);
  run<44>("v_xad/v_or3", cus, out_d, sink_d);
  run<45>("v_sub_u32 sdwa", cus, out_d, sink_d);
  run<50>("v_add_u32 inline const", cus, out_d, sink_d);
  run<51>("v_add_u32 literal", cus, out_d, sink_d);
  run<52>("v_mul_f32 inline 0.5", cus, out_d, sink_d);
  run<53>("v_mul_f32 literal 256.0", cus, out_d, sink_d);
  run<54>("v_add_f32 sgpr operand", cus, out_d, sink_d);
  run<55>("v_mov_b32 from sgpr", cus, out_d, sink_d);
  run<56>("v_or_b32", cus, out_d, sink_d);
  run<57>("v_xor_b32", cus, out_d, sink_d);
  run<58>("v_min_u32", cus, out_d, sink_d);
  run<59>("v_add_u32_e64", cus, out_d, sink_d);
  run<60>("v_sub_f32", cus, out_d, sink_d);
  run<61>("v_cmp_lt_i32_e64 (sgpr pair)", cus, out_d, sink_d);
  run<62>("v_cndmask_b32 vcc", cus, out_d, sink_d);
  run<63>("v_perm_b32", cus, out_d, sink_d);
  run<64>("v_ashrrev_i32", cus, out_d, sink_d);
  run<65>("v_mul_u32_u24", cus, out_d, sink_d);
  run<66>("v_max_f32", cus, out_d, sink_d);
  run<67>("v_med3_f32", cus, out_d, sink_d);
  run<68>("v_subrev_u32", cus, out_d, sink_d);
  run<69>("v_fmac_f32", cus, out_d, sink_d);
  run<70>("v_not_b32", cus, out_d, sink_d);
  run<71>("v_add_f32 neg modifier e64", cus, out_d, sink_d);
  run<72>("v_cvt_u32_f32", cus, out_d, sink_d);
  run<73>("v_mul_f32 dpp row_shr", cus, out_d, sink_d);
  run<74>("ds_read_b128 x1 + wait", cus, out_d, sink_d);
  run<75>("ds_read_b64 x1 + wait", cus, out_d, sink_d);
  run<16>("ds_bpermute_b32+wait", cus, out_d, sink_d);
  return 0;
}
