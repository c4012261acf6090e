// Inclusive prefix operations over the 64 lanes of a wavefront with DPP row shifts / row broadcasts
// (6 VALU instructions, no LDS traffic; a shuffle-based scan is 6 dependent ds_bpermute round trips).
// All 64 lanes must be active.
#ifndef SPA_WAVE_SCAN_H
#define SPA_WAVE_SCAN_H
#include <stdint.h>

namespace spa {

__device__ __forceinline__ uint32_t waveScanAdd( uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x111, 0xF, 0xF, true);	// row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x112, 0xF, 0xF, true);	// row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x114, 0xF, 0xF, true);	// row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x118, 0xF, 0xF, true);	// row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x142, 0xA, 0xF, false);	// row_bcast:15 into rows 1,3
	v += (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x143, 0xC, 0xF, false);	// row_bcast:31 into rows 2,3
	return v;
}

__device__ __forceinline__ uint32_t waveScanMax( uint32_t v)		// unsigned maximum (identity 0)
{
	uint32_t t;
	t = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x111, 0xF, 0xF, true); v = t > v ? t : v;
	t = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x112, 0xF, 0xF, true); v = t > v ? t : v;
	t = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x114, 0xF, 0xF, true); v = t > v ? t : v;
	t = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x118, 0xF, 0xF, true); v = t > v ? t : v;
	t = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;
	t = (uint32_t)__builtin_amdgcn_update_dpp( 0, (int)v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;
	return v;
}

} // namespace
#endif
