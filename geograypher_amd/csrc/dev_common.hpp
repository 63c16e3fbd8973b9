#pragma once
// geograypher_amd/csrc/dev_common.hpp -- device helpers shared by the binning kernels and the tile kernel (gfx950, wave64).
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

__device__ __forceinline__ int imin3(int a, int b, int c) { return min(a, min(b, c)); }
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(a, max(b, c)); }

// Inclusive prefix sum over the 64 lanes with DPP moves only: the LDS pipe (ds_bpermute shuffles included) is the tile
// kernel's scarcest resource, VALU issue is not (one extra ds_bpermute per 64-item batch costs 0.34 us per C2 view, 48
// extra VALU instructions 0.9).  Sources outside a row / masked rows contribute the `old` operand, 0.
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);  // rows 1, 3 += total of rows 0, 2
  x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);  // rows 2, 3 += total of rows 0-1
  return x;
}

// Inclusive prefix maximum (unsigned), same DPP pattern.
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t x) {
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
  // the two row-broadcast steps as single instructions (the compiler makes three of each): rows 1, 3 take the maximum
  // with lane 15 of the row before, then rows 2, 3 with lane 31; masked-out rows keep their value
  asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
  return x;
}

}  // namespace
