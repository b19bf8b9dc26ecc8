// Shared pieces of the cross-based cost aggregation kernels (cbca.hip, cbca_fused.hip).
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
	const uint32_t *overflow;     // optional: set by cbca_pack when an arm saturated the packed form -> do nothing
	int gx, gy;                   // strips per row, row chunks
	int d0, nd;                   // planes [d0, d0 + nd) of the volume are processed by this launch
};



constexpr int CS_COLS = 256;
constexpr int CS_STEP = 252;   // output columns per strip
constexpr int CS_PAD = 1024;   // words of padding around p0 / p1 in the scratch (shifted dwordx4 reads may start outside)

__device__ __forceinline__ cb_u4 bytemin4x4_sdwa(cb_u4 a, cb_u4 b)
{
	// byte-lane minima of four words, each byte written in place (the other bytes of the destination are preserved).
	// The four words are interleaved so that an instruction never reads the register the previous one wrote: gfx940+
	// needs a wait state between a partial (dst_sel) write and its consumer, and nothing inserts one inside inline asm.
	cb_u32 r0 = a.x, r1 = a.y, r2 = a.z, r3 = a.w;
#define MC_SDWA_MIN(B) \
	"v_min_u32_sdwa %0, %4, %8 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
	"v_min_u32_sdwa %1, %5, %9 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
	"v_min_u32_sdwa %2, %6, %10 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t" \
	"v_min_u32_sdwa %3, %7, %11 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #B " src1_sel:BYTE_" #B "\n\t"
	asm(MC_SDWA_MIN(0) MC_SDWA_MIN(1) MC_SDWA_MIN(2) MC_SDWA_MIN(3) "s_nop 0"
	    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)
	    : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
#undef MC_SDWA_MIN
	return cb_u4{r0, r1, r2, r3};
}

// s / 9 for two sums at once.  MC_DIV9_LO <= |s| < MC_DIV9_HI (bit patterns) is the range in which the three-operation
// form equals the IEEE quotient for EVERY float (mc_selftest_div9 walks all of them); `ok` reports per element whether
// s lies inside it.
constexpr cb_u32 MC_DIV9_LO = 0x10000000u;   // 2^-95
constexpr cb_u32 MC_DIV9_HI = 0x7e000000u;   // 2^125
__device__ __forceinline__ cb_f2 div9_pk(cb_f2 s)
{
	const float r9 = 0x1.c71c72p-4f;  // RN(1/9)
	const cb_f2 r = cb_f2{r9, r9};
	const cb_f2 q = s * r;
	const cb_f2 e = __builtin_elementwise_fma(cb_f2{-9.0f, -9.0f}, q, s);
	return __builtin_elementwise_fma(e, r, q);
}
__device__ __forceinline__ bool div9_in_range(float s)
{
	return ((__float_as_uint(s) & 0x7fffffffu) - MC_DIV9_LO) < (MC_DIV9_HI - MC_DIV9_LO);
}

struct C2Row { cb_f2 A, B, C, D, E; };   // columns (-1,0) (0,1) (1,2) (2,3) (3,4) relative to the lane's first column

// scratch = [pad | p0 (H*W) | pad | p1 (H*W) | pad | overflow flag], pad = CS_PAD words
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
