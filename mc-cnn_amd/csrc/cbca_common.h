// Shared pieces of the cross-based cost aggregation kernels (cbca.hip).
#pragma once
#include "mc_common.h"

namespace mc {

typedef unsigned cb_u32;
typedef unsigned cb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned cb_u2 __attribute__((ext_vector_type(2)));
typedef float cb_f4 __attribute__((ext_vector_type(4)));
typedef float cb_f2 __attribute__((ext_vector_type(2)));

typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bytemin4(uint32_t a, uint32_t b)
{
	// per-byte unsigned minimum through two packed 16-bit minima (even / odd bytes)
	const uint32_t ae = a & 0x00ff00ffu, be = b & 0x00ff00ffu;
	const uint32_t ao = (a >> 8) & 0x00ff00ffu, bo = (b >> 8) & 0x00ff00ffu;
	const ushort2v me = __builtin_elementwise_min(__builtin_bit_cast(ushort2v, ae), __builtin_bit_cast(ushort2v, be));
	const ushort2v mo = __builtin_elementwise_min(__builtin_bit_cast(ushort2v, ao), __builtin_bit_cast(ushort2v, bo));
	return __builtin_bit_cast(uint32_t, me) | (__builtin_bit_cast(uint32_t, mo) << 8);
}

struct CbcaArgs {
	const uint32_t *p0, *p1;      // packed arm lengths (H,W)
	const float *vin;
	float *vout;
	int D, H, W, direction;
	int rb;                       // output rows per strip
	const uint32_t *overflow;     // optional, cbca_pack's flags: [0] an arm saturated the packed form -> do nothing; [1] an arm > 4; [2] an arm > 13
	int by_arm;                   // 1: flag [1] selects the kernel (window kernel iff no arm > 4, strip kernel otherwise)
	int gx, gy;                   // strips per row, row chunks
	int d0, nd;                   // planes [d0, d0 + nd) of the volume are processed by this launch
};



constexpr int CS_COLS = 256;
constexpr int CS_STEP = 252;   // output columns per strip
constexpr int CS_PAD = 1024;   // words of padding around p0 / p1 in the scratch (shifted dwordx4 reads may start outside)

constexpr int CS_FLAGS = 4;    // flag words behind the packed lengths (three used)

// scratch = [pad | p0 (H*W) | pad | p1 (H*W) | pad | CS_FLAGS flag words], pad = CS_PAD words
size_t cbca_scratch_bytes(int H, int W);   // cbca.hip

struct CbcaScratch { uint32_t *p0, *p1, *flag; };
static inline CbcaScratch cbca_scratch(const void *scratch, int H, int W)
{
	CbcaScratch s;
	s.p0 = (uint32_t *)scratch + CS_PAD;
	s.p1 = s.p0 + (size_t)H * W + CS_PAD;
	s.flag = s.p1 + (size_t)H * W + CS_PAD;
	return s;
}


}  // namespace mc
