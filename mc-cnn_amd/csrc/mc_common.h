// Shared helpers for the gfx950 kernels of libmcadcensus.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/mc_adcensus.h"

namespace mc {

// ---- error reporting (mc_last_error) ---------------------------------------
void set_error(const char *fmt, ...);
int check_launch(const char *what);  // hipPeekAtLastError -> rc, like checkCudaError (adcensus.cu:31-36)

#define MC_REQUIRE(cond, ...)                 \
	do {                                      \
		if (!(cond)) {                        \
			mc::set_error(__VA_ARGS__);       \
			return MC_EINVAL;                 \
		}                                     \
	} while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// launch configuration of the cbca strip / tile kernels; the defaults derive everything from the problem size
struct CbcaCfg {
	int rb = 0;        // output rows per strip (0 = auto)
	int nt = -1;       // volume cache policy: -1 auto, 0 default, 1 non-temporal
	int d0 = 0, nd = 0;// planes [d0, d0 + nd) only (nd = 0: all)
	int variant = 0;   // tile kernel: geometry variant (0 = the product's choice)
	void *plan = nullptr;   // tile kernel: cbca_plan_bytes() bytes holding the item order of this pair and direction, or null
	int plan_mode = 0; // 0 = no plan (every launch sorts its items), 1 = this launch sorts and writes the plan, 2 = this launch reads it
	size_t plan_bytes = 0;  // size of *plan
	int lean_rb = 0;   // ... rows per wave of the lean kernels that wrote / read the list (0 = the product's choice)
	int lean_variant = -1;  // ... launch variant of the lean kernels (-1 = the product's choice; cbca_lean.hip)
	bool lean = false; // route CR_STRIP is served by the lean + list kernels (cbca_lean.hip) out of the list cbca_classify wrote to *plan
	bool lean_two_pass = false;   // ... by cbca_lean2x (two passes per launch, launched by the caller), whose list has another wave geometry: cbca_by_arms
	                              // then launches only the kernels of the other routes and the strip kernel as that list's fallback
	bool planned = false;         // mc_predict where it keeps a plan area and 5 <= L1 <= 14: the list was classified before the direction's first pass, and a
	                              // pass is the tile kernel's long-arm instance (CR_PLANNED_TILE13) + for a single pass the strip kernel (CR_STRIP_IF_LIST)
};

// ---- wave64 cross-lane primitives (DPP, no LDS round trip) -------------------
// gfx9 DPP controls.
constexpr int DPP_QUAD_1032 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_2301 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;
constexpr int DPP_WAVE_SHL1 = 0x130;       // lane i <- lane i+1
constexpr int DPP_WAVE_SHR1 = 0x138;       // lane i <- lane i-1

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float old, float src)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
	                                                  ROW_MASK, 0xF, false));
}

// min over the 64 lanes, NaN-ignoring (fminf), returned wave-uniform.
__device__ __forceinline__ float wave_min(float v)
{
	v = fminf(v, dpp_mov<DPP_QUAD_1032>(v, v));
	v = fminf(v, dpp_mov<DPP_QUAD_2301>(v, v));
	v = fminf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v, v));
	v = fminf(v, dpp_mov<DPP_ROW_MIRROR>(v, v));
	v = fminf(v, dpp_mov<DPP_ROW_BCAST15, 0xA>(v, v));
	v = fminf(v, dpp_mov<DPP_ROW_BCAST31, 0xC>(v, v));
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// value of lane-1 (lane 0 gets `fill`) / lane+1 (lane 63 gets `fill`)
__device__ __forceinline__ float lane_from_below(float v, float fill) { return dpp_mov<DPP_WAVE_SHR1>(fill, v); }
__device__ __forceinline__ float lane_from_above(float v, float fill) { return dpp_mov<DPP_WAVE_SHL1>(fill, v); }

}  // namespace mc
