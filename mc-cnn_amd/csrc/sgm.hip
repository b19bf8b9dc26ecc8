// Semiglobal matching for gfx950: one scan LINE per wave64, lanes across the
// disparity axis (VPL consecutive d per lane), the whole line walked inside one
// launch with the previous step's L_r held in VGPRs.
//
// Replaces adcensus.sgm2 (adcensus.cu:620-697; kernels sgm2<0..3>, 535-618),
// which launches one grid per scan STEP (2W+2H launches per call) and keeps
// L_r in a global `tmp` array.  Here a call is 1 prep launch + 4 direction
// sweeps; the recurrence itself (min / + / - only) is evaluated with the same
// operations, so results are bit-identical.
//
// Data layout: (H,W,D) with D contiguous and a pixel stride `ds` (>= D).  A
// wave reads one pixel's D costs as one coalesced run (VPL=4 -> dwordx4 per
// lane, 1 KiB per wave-load at D=256) for every direction.
//
// The intensity tests D1/D2 < tau_so (adcensus.cu:586-605) only select one of
// three penalty pairs, so they are precomputed by sgm_prep_kernel as 2-bit
// classes: cls0[r][y][x] for the reference pixel (wave-uniform) and, for the
// partner pixel x + d*direction, "window" bytes win[o][r][y][s] that pack the
// classes of 4 consecutive pixels s..s+3 (ascending for direction +1,
// descending for -1) so that a lane fetches the classes of its VPL
// disparities with VPL/4 byte loads.
#include "mc_common.h"

// cache policy of the volume accesses: nt (bit 1) -- every cost run is read or written once per sweep; streaming them keeps
// the edge-class maps (re-read by every line) in L2
#define MC_SGM_VOL_AUX 2

namespace mc {

constexpr int SGM_PADW = 520;  // >= 63*8 + 7 + slack: window starts reach -(VPL*63+VPL-1)

struct SgmPassArgs {
	const float *C[2];    // input cost volume(s), (H,W,ds)
	const float *accin[2];// running sum read by MODE 1,2,3
	const float *accin2[2];// second partial sum read by MODE 3
	float *out[2];        // where this sweep writes
	float *out2[2];       // DUAL: where the concurrent second direction writes
	float *disp[2];       // ARGMIN output (H,W), may be null
	int direction[2];
	int nvol;
	int H, W, D, ds;
	const uint8_t *cls0;  // [4][cls_plane], plane = H*W bytes padded to a dword multiple
	int64_t cls_plane;
	const uint8_t *win;   // [2][4][H][Wm]
	int Wm;
	float P1[3], P2[3], P1a[3];  // 0: both < tau, 1: mixed, 2: both > tau ; P1a = P1 / alpha1
};

__device__ __forceinline__ int cls_of(float v, float tau) { return v < tau ? 0 : (v > tau ? 2 : 1); }

// r: 0 right (dx=1), 1 left (dx=-1), 2 down (dy=1), 3 up (dy=-1)  (adcensus.cu:541-565)
__global__ void __launch_bounds__(256) sgm_prep_kernel(const float *__restrict__ x0, const float *__restrict__ x1,
                                                       uint8_t *__restrict__ cls0, uint8_t *__restrict__ win,
                                                       int H, int W, int Wm, int64_t cls_plane, float tau_so)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t total = (int64_t)4 * H * Wm;
	if (id >= total) return;
	const int i = (int)(id % Wm);
	const int y = (int)((id / Wm) % H);
	const int r = (int)(id / ((int64_t)Wm * H));
	const int dx = r == 0 ? 1 : (r == 1 ? -1 : 0);
	const int dy = r == 2 ? 1 : (r == 3 ? -1 : 0);
	const bool yok = (y - dy >= 0) && (y - dy < H);

	if (i < W) {
		const int x = i;
		int c = 1;
		if (yok && x - dx >= 0 && x - dx < W) {
			// D1 = COLOR_DIFF(x0, ind2, ind2 - dy*size2 - dx), adcensus.cu:587
			c = cls_of(fabsf(x0[y * W + x] - x0[(y - dy) * W + x - dx]), tau_so);
		}
		cls0[(int64_t)r * cls_plane + (int64_t)y * W + x] = (uint8_t)c;
	}
	const int s = i - SGM_PADW;
	unsigned asc = 0, desc = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int xx = s + k;
		float D2 = 10.0f;  // adcensus.cu:590-591
		if (yok && xx >= 0 && xx < W && xx - dx >= 0 && xx - dx < W) {
			D2 = fabsf(x1[y * W + xx] - x1[(y - dy) * W + xx - dx]);  // adcensus.cu:593
		}
		const unsigned c = (unsigned)cls_of(D2, tau_so);
		asc |= c << (2 * k);
		desc |= c << (2 * (3 - k));
	}
	win[((int64_t)(0 * 4 + r) * H + y) * Wm + i] = (uint8_t)asc;
	win[((int64_t)(1 * 4 + r) * H + y) * Wm + i] = (uint8_t)desc;
}

typedef unsigned uint4v __attribute__((ext_vector_type(4)));

template <int VPL, int NACC>
struct StepData {
	float c[VPL];
	float a[NACC > 0 ? VPL : 1];
	float a2[NACC > 1 ? VPL : 1];
	unsigned pk;
	unsigned a0;   // class of the reference pixel's edge (same value in every lane)
};

// One pixel's run of costs as a raw buffer: base = vol + pix*ds, num_records = the run's bytes, so that
// lanes whose disparities lie beyond the run are dropped (stores) / zero-filled (loads) by the hardware
// range check instead of by exec-masked branches -- the steady-state loop stays straight-line code and
// the compiler's s_waitcnt vmcnt(N) counts the steps in flight exactly.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pixel_rsrc(const float *base, int64_t elem_off, int bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc((void *)(base + elem_off), 0, bytes, 0x00020000);
}

// What a sweep does with its directional cost L_r (val), given the running sum(s):
// MODE 0: out = 0 + L_r            (first direction: no read of the sum, folds out:zero(), main.lua:1014)
// MODE 1: out = accin + L_r        (reference-compatible accumulate, adcensus.cu:569,616)
// MODE 2: out = (accin + L_r)/4    (last direction; folds vol:copy(out):div(4), main.lua:1017,1020)
// MODE 3: out = (accin + accin2) + L_r   (third direction after a DUAL first launch)
// DUAL (DIRN 0 only): waves [0,n) sweep right and write out = 0 + L_0; waves [n,2n) sweep LEFT over the same
//   lines and write out2 = L_1.  The two horizontal directions then run concurrently (twice the waves in
//   flight where a direction alone has fewer lines than the chip has SIMDs); the sum order of the
//   reference, ((0 + L_0) + L_1) + L_2) + L_3, is restored by MODE 3.
// U = steps kept in flight per wave (register ring): the scan is a strict recurrence, so memory latency is
//   covered by prefetch depth, not by occupancy.
template <int DIRN, int VPL, int MODE, bool ARGMIN, bool VEC, int U, bool DUAL>
__global__ void __launch_bounds__(256) sgm_pass_kernel(const SgmPassArgs A)
{
	constexpr int NACC = MODE == 0 ? 0 : (MODE == 3 ? 2 : 1);
	const int lane = threadIdx.x & 63;
	int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	const int H = A.H, W = A.W, D = A.D, ds = A.ds, Wm = A.Wm;
	const int nlines = DIRN <= 1 ? H : W;
	const int nsteps = DIRN <= 1 ? W : H;
	const int nw = A.nvol * nlines;
	bool second = false;
	if (DUAL) {
		if (wave >= 2 * nw) return;
		second = wave >= nw;
		if (second) wave -= nw;
	} else if (wave >= nw) {
		return;
	}
	const int dirn = (DUAL && second) ? 1 : DIRN;
	const int v = wave / nlines;
	const int line = wave - v * nlines;
	const int direction = A.direction[v];
	const float *__restrict__ Cp = A.C[v];
	const float *__restrict__ Ain = A.accin[v];
	const float *__restrict__ Ain2 = A.accin2[v];
	float *__restrict__ Out = (DUAL && second) ? A.out2[v] : A.out[v];
	float *__restrict__ Disp = A.disp[v];

	// line geometry: pixel of step s is (y0 + s*sy, x0 + s*sx)
	const int x0s = dirn == 0 ? 0 : (dirn == 1 ? W - 1 : line);
	const int y0s = dirn == 2 ? 0 : (dirn == 3 ? H - 1 : line);
	const int sx = dirn == 0 ? 1 : (dirn == 1 ? -1 : 0);
	const int sy = dirn == 2 ? 1 : (dirn == 3 ? -1 : 0);

	const int dbase = VPL * lane;
	const uint8_t *__restrict__ cls0 = A.cls0 + (int64_t)dirn * A.cls_plane;
	// window bytes: buffer base = win row + x + SGM_PADW - WBIAS (wave-uniform), per-lane constant voffset
	constexpr int WBIAS = 1024;  // >= the most negative window start (VPL*63 + VPL - 1 + 3)
	const uint8_t *__restrict__ win = A.win + ((int64_t)((direction > 0 ? 0 : 1) * 4 + dirn) * H) * Wm + SGM_PADW - WBIAS;
	// window start of chunk q at image column x: direction +1: x + VPL*lane + 4q ; -1: x - VPL*lane - 4q - 3
	const int woff = (direction > 0 ? dbase : -dbase - 3) + WBIAS;
	const int wq = direction > 0 ? 4 : -4;
	const int run_bytes = (VEC ? ds : D) * 4;

	const float INF = __builtin_inff();
	// the nine penalties live in SGPRs for the whole sweep
	const float P1mid = A.P1[1], P2mid = A.P2[1], P1amid = A.P1a[1];
	const float P1lo = A.P1[0], P2lo = A.P2[0], P1alo = A.P1a[0];
	const float P1hi = A.P1[2], P2hi = A.P2[2], P1ahi = A.P1a[2];

	auto load_run = [&](float (&dst)[VPL], const float *base, int64_t pix) {
		const __amdgpu_buffer_rsrc_t r = pixel_rsrc(base, pix * ds, run_bytes);
		if (VEC) {
#pragma unroll
			for (int q = 0; q < VPL / 4; ++q) {
				const uint4v t = __builtin_amdgcn_raw_buffer_load_b128(r, (dbase + 4 * q) * 4, 0, MC_SGM_VOL_AUX);
				dst[4 * q + 0] = __uint_as_float(t.x); dst[4 * q + 1] = __uint_as_float(t.y);
				dst[4 * q + 2] = __uint_as_float(t.z); dst[4 * q + 3] = __uint_as_float(t.w);
			}
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j) dst[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (dbase + j) * 4, 0, 0));
		}
	};

	auto load_step = [&](StepData<VPL, NACC> &sd, int s) {
		const int x = x0s + s * sx, y = y0s + s * sy;
		const int64_t pix = (int64_t)y * W + x;
		load_run(sd.c, Cp, pix);
		if constexpr (NACC >= 1) load_run(sd.a, Ain, pix);
		if constexpr (NACC >= 2) load_run(sd.a2, Ain2, pix);
		unsigned pk = 0;
		const __amdgpu_buffer_rsrc_t rw =
		    __builtin_amdgcn_make_buffer_rsrc((void *)(win + (int64_t)y * Wm + x), 0, 2 * WBIAS + 64, 0x00020000);
#pragma unroll
		for (int q = 0; q < VPL / 4; ++q) pk |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rw, woff + q * wq, 0, 0) << (8 * q);
		sd.pk = pk;
		const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(cls0 + pix), 0, 1, 0x00020000);
		sd.a0 = __builtin_amdgcn_raw_buffer_load_b8(rc, 0, 0, 0);  // every lane reads the same byte
	};

	auto store_step = [&](const float (&o)[VPL], int s) {
		const int x = x0s + s * sx, y = y0s + s * sy;
		const int64_t pix = (int64_t)y * W + x;
		const __amdgpu_buffer_rsrc_t r = pixel_rsrc(Out, pix * ds, run_bytes);
		if (VEC) {
#pragma unroll
			for (int q = 0; q < VPL / 4; ++q) {
				uint4v t;
				t.x = __float_as_uint(o[4 * q + 0]); t.y = __float_as_uint(o[4 * q + 1]);
				t.z = __float_as_uint(o[4 * q + 2]); t.w = __float_as_uint(o[4 * q + 3]);
				__builtin_amdgcn_raw_buffer_store_b128(t, r, (dbase + 4 * q) * 4, 0, MC_SGM_VOL_AUX);
			}
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[j]), r, (dbase + j) * 4, 0, 0);
		}
		if (ARGMIN) {
			// torch.min(vol,2) - 1 (main.lua:1049-1050) on the finished pixel: first strict
			// minimum from +INF, NaN never wins (spatial_argmin convention, adcensus.cu:251-260)
			float best = INF;
			int bi = 0;
#pragma unroll
			for (int j = 0; j < VPL; ++j) {
				if (dbase + j < D && o[j] < best) {
					best = o[j];
					bi = dbase + j;
				}
			}
			const float mall = wave_min(best);
			const unsigned long long cand = __ballot(best == mall && best < INF);
			// branch-free: ffs = 0 when no lane holds a finite minimum (all-NaN pixel -> index 0)
			const int f = __builtin_ffsll((long long)cand);
			const int got = __builtin_amdgcn_readlane(bi, (f - 1) & 63);
			const int idx = f ? got : 0;
			// one float per pixel: a 4-byte buffer at disp + pix, so only lane 0 is in range
			const __amdgpu_buffer_rsrc_t rd = pixel_rsrc(Disp, pix, 4);
			__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)idx), rd, lane * 4, 0, 0);
		}
	};

	float prev[VPL];
	float m = 0.0f;
#pragma unroll
	for (int j = 0; j < VPL; ++j) prev[j] = 0.0f;

	auto process = [&](const StepData<VPL, NACC> &sd, int s) {
		float val[VPL], o[VPL];
		const bool first = s == 0;  // border branch, adcensus.cu:567-572: L_r = C
		const int a0 = __builtin_amdgcn_readfirstlane((int)sd.a0);
		const int amatch = a0 == 1 ? 3 : a0;
		const float P1x = a0 == 0 ? P1lo : P1hi;
		const float P2x = a0 == 0 ? P2lo : P2hi;
		const float P1ax = a0 == 0 ? P1alo : P1ahi;
		const float down = lane_from_below(prev[VPL - 1], INF);
		const float up = lane_from_above(prev[0], INF);
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			const int b = (sd.pk >> (2 * j)) & 3;
			const bool match = b == amatch;
			const float P2 = match ? P2x : P2mid;
			const float P1 = match ? P1x : P1mid;
			const float P1a = match ? P1ax : P1amid;
			const float pm = j > 0 ? prev[j > 0 ? j - 1 : 0] : down;
			const float pp = j < VPL - 1 ? prev[j < VPL - 1 ? j + 1 : 0] : up;
			// adcensus.cu:607-613
			float cost = fminf(prev[j], m + P2);
			cost = fminf(cost, pm + (DIRN == 2 ? P1a : P1));
			cost = fminf(cost, pp + (DIRN == 3 ? P1a : P1));
			const float rec = (sd.c[j] + cost) - m;  // adcensus.cu:615
			val[j] = first ? sd.c[j] : rec;
		}
		float nm = INF;
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			if (MODE == 0) o[j] = (DUAL && second) ? val[j] : 0.0f + val[j];
			else if (MODE == 1) o[j] = sd.a[j] + val[j];
			else if (MODE == 2) o[j] = (sd.a[j] + val[j]) * 0.25f;
			else o[j] = (sd.a[j] + sd.a2[j]) + val[j];
			prev[j] = (dbase + j < D) ? fminf(val[j], INF) : INF;  // NaN -> +INF: fminf semantics of the recurrence
			nm = fminf(nm, prev[j]);
		}
		m = wave_min(nm);
		store_step(o, s);
	};

	StepData<VPL, NACC> ring[U];
	const int last = nsteps - 1;
#pragma unroll
	for (int u = 0; u < U; ++u) load_step(ring[u], u < last ? u : last);

	int g = 0;
	// steady state: straight-line code, every slot is consumed and immediately refilled U steps ahead
	// (refills past the end of the line re-read its last pixel; they are never consumed)
	for (; g + U <= nsteps; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			// keep each step's work behind its own s_waitcnt: without the barrier the machine scheduler hoists the
			// recurrence-independent adds of ALL ring slots to the loop head, i.e. waits for every prefetch at once
			__builtin_amdgcn_sched_barrier(0);
			process(ring[u], s);
			const int sn = s + U;
			load_step(ring[u], sn < last ? sn : last);
		}
	}
#pragma unroll
	for (int u = 0; u < U; ++u) {
		if (g + u < nsteps) process(ring[u], g + u);
	}
}

// ---------------------------------------------------------------------------

size_t sgm_maps_bytes(int H, int W)
{
	const size_t Wm = (size_t)W + 2 * SGM_PADW;
	const size_t plane = ((size_t)H * W + 3 + 4) / 4 * 4;  // dword-aligned class planes (+4: the dword read may straddle)
	size_t b = 4 * plane + (size_t)8 * H * Wm;
	return (b + 255) & ~(size_t)255;
}

int sgm_prep(const float *x0, const float *x1, void *maps, int H, int W, float tau_so, hipStream_t st)
{
	const int Wm = W + 2 * SGM_PADW;
	const size_t plane = ((size_t)H * W + 3 + 4) / 4 * 4;
	uint8_t *cls0 = (uint8_t *)maps;
	uint8_t *win = cls0 + 4 * plane;
	const int64_t total = (int64_t)4 * H * Wm;
	hipLaunchKernelGGL(sgm_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, x0, x1, cls0, win, H, W, Wm, (int64_t)plane,
	                   tau_so);
	return check_launch("sgm_prep");
}

template <int DIRN, int MODE, bool ARGMIN, bool DUAL>
static void launch_pass(const SgmPassArgs &A, bool vec, hipStream_t st)
{
	const int nlines = DIRN <= 1 ? A.H : A.W;
	const int waves = A.nvol * nlines * (DUAL ? 2 : 1);
	const dim3 grid(cdiv(waves, 4)), block(256);
	// fewer waves than SIMDs (1024): nothing but prefetch depth hides HBM latency
	// measured on MI355X, A/B on one box (round 4): the depth matters little once >= 4 -- KITTI 370x1226x228 all four
	// sweeps 2.016 ms at (horizontal 4, up 4), 2.024 at (8, 4), 2.009 at (4, 8), 1.982 at (8, 16), 2.008 at (16, 8), the down
	// sweep (three loads per step) at 16 throughout; 1500x1000x256 within 1 % either way.  The sweeps run at what the
	// memory system gives long-lived waves that each walk their own line (4.3-5.1 TB/s), not at a latency bound.
#ifndef MC_SGM_U_H
#define MC_SGM_U_H 8
#endif
#ifndef MC_SGM_U_UP
#define MC_SGM_U_UP 16
#endif
	const int U = DIRN == 2 ? 16 : (DIRN == 3 ? MC_SGM_U_UP : MC_SGM_U_H);
#define MC_SGM_GO(VPL_, VEC_, U_) \
	hipLaunchKernelGGL((sgm_pass_kernel<DIRN, VPL_, MODE, ARGMIN, VEC_, U_, DUAL>), grid, block, 0, st, A)
	if (A.D <= 256) {
		if (vec) { if (U == 16) MC_SGM_GO(4, true, 16); else if (U == 8) MC_SGM_GO(4, true, 8); else MC_SGM_GO(4, true, 4); }
		else MC_SGM_GO(4, false, 4);
	} else {
		if (vec) { if (U >= 8) MC_SGM_GO(8, true, 8); else MC_SGM_GO(8, true, 4); }
		else MC_SGM_GO(8, false, 2);
	}
#undef MC_SGM_GO
}

// Four direction sweeps over nvol (1 or 2) volumes.
//   fused = false: every sweep does out += L_r (adcensus.sgm2 contract, out pre-zeroed by caller)
//   fused = true : right and left sweeps run concurrently (out = 0 + L_0, out2 = L_1), the down sweep
//                  writes (out + out2) + L_2 to out, the up sweep writes (out + L_3)/4 and, if disp[] is
//                  set, the argmin of the finished pixel.  `out2` is scratch of the same size as out.
int sgm_sweeps(const float *const C[2], float *const out[2], float *const out2[2], float *const disp[2],
               const int direction[2], int nvol, int H, int W, int D, int ds, const void *maps, float pi1, float pi2,
               float alpha1, float q1, float q2, bool fused, hipStream_t st)
{
	SgmPassArgs A;
	for (int v = 0; v < 2; ++v) {
		const int k = v < nvol ? v : 0;
		A.C[v] = C[k];
		A.accin[v] = out[k];
		A.accin2[v] = out2 ? out2[k] : nullptr;
		A.out[v] = out[k];
		A.out2[v] = out2 ? out2[k] : nullptr;
		A.disp[v] = disp ? disp[k] : nullptr;
		A.direction[v] = direction[k];
	}
	A.nvol = nvol;
	A.H = H; A.W = W; A.D = D; A.ds = ds;
	A.Wm = W + 2 * SGM_PADW;
	A.cls0 = (const uint8_t *)maps;
	A.cls_plane = (int64_t)(((size_t)H * W + 3 + 4) / 4 * 4);
	A.win = A.cls0 + 4 * A.cls_plane;
	// adcensus.cu:595-605 -- float divisions exactly as written there
	A.P1[0] = pi1; A.P2[0] = pi2;
	A.P1[1] = pi1 / q1; A.P2[1] = pi2 / q1;
	A.P1[2] = pi1 / (q1 * q2); A.P2[2] = pi2 / (q1 * q2);
	for (int k = 0; k < 3; ++k) A.P1a[k] = A.P1[k] / alpha1;  // adcensus.cu:609,612

	bool vec = (ds % 4 == 0);
	for (int v = 0; v < nvol; ++v) {
		vec = vec && ((uintptr_t)C[v] % 16 == 0) && ((uintptr_t)out[v] % 16 == 0);
		if (out2) vec = vec && ((uintptr_t)out2[v] % 16 == 0);
	}
	const bool am = fused && disp && disp[0];
	if (!fused || !out2) {
		const bool f = fused;  // fused without scratch: sequential sweeps with the zero and /4 still folded
		if (f) launch_pass<0, 0, false, false>(A, vec, st); else launch_pass<0, 1, false, false>(A, vec, st);
		launch_pass<1, 1, false, false>(A, vec, st);
		launch_pass<2, 1, false, false>(A, vec, st);
		if (!f) launch_pass<3, 1, false, false>(A, vec, st);
		else if (am) launch_pass<3, 2, true, false>(A, vec, st);
		else launch_pass<3, 2, false, false>(A, vec, st);
	} else {
		launch_pass<0, 0, false, true>(A, vec, st);
		launch_pass<2, 3, false, false>(A, vec, st);
		if (am) launch_pass<3, 2, true, false>(A, vec, st);
		else launch_pass<3, 2, false, false>(A, vec, st);
	}
	return check_launch("sgm_pass");
}

}  // namespace mc
