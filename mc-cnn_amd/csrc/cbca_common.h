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
	const uint32_t *flags;        // optional, cbca_pack's flag words: the launch runs only if cbca_gate(flags, route)
	int route;
	int gx, gy;                   // strips per row, row chunks
	int d0, nd;                   // planes [d0, d0 + nd) of the volume are processed by this launch
	void *plan;                   // tile kernel: the pair's item order per (plane, region, step), written by one launch and read by the others
	int spr;                      // ... steps per region
	size_t plan_m, plan_ud;       // ... byte offsets of the combined runs / vertical arms behind the tables
	int wp;                       // ... their row length in pixels
	int lean_rb;                  // strip kernel as the lean kernels' fallback: rows per wave the pair's list must have been written for
};



constexpr int CS_COLS = 256;
constexpr int CS_STEP = 252;   // output columns per strip
constexpr int CS_PAD = 1024;   // words of padding around p0 / p1 in the scratch (shifted dwordx4 reads may start outside)

// flag words behind the packed lengths, written by cbca_pack: what the pair's arms look like and the kernel they call for
constexpr int CS_FLAGS = 8;
enum { CF_SATURATED = 0,      // an arm longer than 254 pixels (the packed form saturates): one thread per voxel
       CF_ARM_GT4 = 1, CF_ARM_GT13 = 2, CF_ROUTE = 3,
       CF_UNIT_PIXELS = 4 };  // [4], [5]: pixels of image 0 / 1 whose four arms are all <= 1
enum { CR_TILE4 = 0,          // every arm <= 4 (L1 <= 5): tile kernel, short-arm instance
       CR_TILE13 = 1,         // every arm <= 13 (L1 <= 14), real-scene statistics: tile kernel, long-arm instance
       CR_STRIP = 2,          // longer arms, or nearly every support the minimal 3 x 3 (textures): strip kernel
       CR_DIRECT = 3,         // saturated packed form
       // launch conditions that are not a single route (CbcaArgs::route):
       CR_ARMS_LE4 = 16, CR_ARMS_LE13 = 17,   // the forced tile instances of the test hook: any pair whose arms fit
       CR_NOT_DIRECT = 18,                    // the forced strip kernel: any pair the packed form holds
       CR_STRIP_OR_TILE13 = 19,               // strip kernel where L1 > 14 is known: the routes of longer arms and of arms <= 13 alike
       CR_STRIP_IF_NO_LIST = 20,              // strip kernel as the fallback of the lean + list kernels (cbca_lean.hip): route CR_STRIP and no usable list
       CR_NOT_DIRECT_IF_NO_LIST = 21,         // ... of the forced lean + list kernels of the test hook
       // mc_predict's passes where it keeps a plan area and knows 5 <= L1 <= 14 (the list is classified before the first pass of a direction):
       CR_PLANNED_TILE13 = 22,                // tile kernel, long-arm instance: the tile routes, and route CR_STRIP whose list is unusable (a texture with flat regions)
       CR_STRIP_IF_LIST = 23 };               // strip kernel for a SINGLE pass of route CR_STRIP with a usable list (pairs of passes: cbca_lean2x)

// head of the pair's plan area when the route is CR_STRIP (cbca_lean.hip): the list of outputs whose support is not the minimal 3 x 3
enum { LH_COUNT = 0, LH_OVERFLOW = 1, LH_D = 2, LH_H = 3, LH_W = 4, LH_DIR = 5, LH_MAGIC = 6, LH_RB = 7, LH_WORDS = 64 };   // (wave table from word LH_WORDS on, then the slots)
constexpr uint32_t LH_MAGIC_VALUE = 0x4c495354u;
// written by cbca_classify_kernel for exactly this problem and wave geometry (rb rows per wave), and complete.
// INVARIANT every caller keeps: the answer means something only together with `route == CR_STRIP` for the CURRENT pair -- the head is reset
// (cbca_list_reset_kernel) and rewritten only when the pair takes that route, so a cached workspace may still hold a valid-looking head from an
// EARLIER pair of the same shape whenever the current one takes a tile route.  A consumer that trusted list_valid alone would read another
// image's records.
__device__ __forceinline__ bool list_valid(const uint32_t *__restrict__ hdr, int D, int H, int W, int direction, int rb)
{
	return hdr[LH_MAGIC] == LH_MAGIC_VALUE && hdr[LH_D] == (uint32_t)D && hdr[LH_H] == (uint32_t)H && hdr[LH_W] == (uint32_t)W &&
	       hdr[LH_DIR] == (uint32_t)(direction + 1) && hdr[LH_RB] == (uint32_t)rb && !hdr[LH_OVERFLOW];
}

// the launch conditions of mc_predict's planned passes (see CR_PLANNED_TILE13): hdr = the pair's plan area, whose first LH_WORDS words are the list's head
__device__ __forceinline__ bool cbca_gate_planned(const uint32_t *__restrict__ flags, int route, const uint32_t *__restrict__ hdr, int D, int H, int W,
                                                  int direction, int rb)
{
	const uint32_t r = flags[CF_ROUTE];
	if (route == CR_PLANNED_TILE13) return r == CR_TILE13 || r == CR_TILE4 || (r == CR_STRIP && !flags[CF_ARM_GT13] && !list_valid(hdr, D, H, W, direction, rb));
	return r == CR_STRIP && list_valid(hdr, D, H, W, direction, rb);   // CR_STRIP_IF_LIST
}

// does this launch run? (flags == nullptr: the caller knows the arms and launched exactly the right kernel)
__device__ __forceinline__ bool cbca_gate(const uint32_t *__restrict__ flags, int route)
{
	if (!flags) return true;
	const uint32_t r = flags[CF_ROUTE];
	switch (route) {
	case CR_ARMS_LE4: return !flags[CF_ARM_GT4];
	case CR_ARMS_LE13: return !flags[CF_ARM_GT13];
	case CR_NOT_DIRECT: return r != CR_DIRECT;
	case CR_STRIP_OR_TILE13: return r == CR_STRIP || r == CR_TILE13;
	default: return r == (uint32_t)route;
	}
}

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
