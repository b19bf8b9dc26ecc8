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
#define MC_SGM_ST_AUX MC_SGM_VOL_AUX   // (the stores' policy on its own: profiles/r05_ab_sgm_store_policy.txt)

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


// v_min_f32 / v_min3_f32 as written: fminf() through the compiler canonicalises every operand it cannot prove quiet (a v_max x, x
// each) -- no signalling NaN can reach a minimum here: the operands are results of additions, of other minima, or +INF.
__device__ __forceinline__ float vmin2(float a, float b)
{
	float r;
	asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ float vmin3(float a, float b, float c)
{
	float r;
	asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// min over the 64 lanes of quiet values, wave-uniform (mc::wave_min with the DPP folded into the minimum: 13 instructions
// instead of 31 -- a sweep runs one or two waves per SIMD, where every instruction is ~8 cycles of the recurrence)
__device__ __forceinline__ float wave_min_q(float v)
{
	float r;
	asm volatile("s_nop 1\n\tv_min_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_min_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_min_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_min_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_min_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	             "s_nop 1\n\tv_min_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	             "s_nop 0\n\tv_readlane_b32 %0, %1, 63"
	             : "=s"(r), "+v"(v));
	return r;
}

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
template <int DIRN, int VPL, int MODE, bool ARGMIN, bool VEC, int U, bool DUAL, bool FAR>
__device__ __forceinline__ void sgm_line(const SgmPassArgs &A, int wave)
{
	constexpr int NACC = MODE == 0 ? 0 : (MODE == 3 ? 2 : 1);
	const int lane = threadIdx.x & 63;
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

	// Addressing.  A line's pixels are pix0 + s*dp (dp = +-1 or +-W).  FAR = false (every line spans < 2 GiB of a volume): one
	// descriptor per volume for the whole sweep -- base = the line's lowest-addressed pixel, num_records = the line's span --
	// and a step costs one scalar multiply-add per array: rel(s)*ds*4 goes into the instruction's scalar offset (which the
	// range check includes on this hardware: a per-pixel num_records cannot be combined with it).  The lanes beyond a pixel's
	// run carry an out-of-range offset instead (loop-invariant), so their loads still return 0 and their stores are dropped.
	// FAR = true rebuilds a 64-bit base per pixel (13 scalar instructions per array and step, a third of the loop).
	const int last = nsteps - 1;
	const int pix0 = y0s * W + x0s, dp = sy * W + sx;
	const int minpix = dp > 0 ? pix0 : pix0 + last * dp;
	const unsigned c1v = (unsigned)(dp * ds * 4);                       // modular: c0v + s*c1v is the true offset in [0, 2^31)
	const unsigned c0v = dp > 0 ? 0u : (unsigned)last * (unsigned)(-dp * ds * 4);
	const unsigned span = (unsigned)last * (unsigned)((dp > 0 ? dp : -dp) * ds * 4) + (unsigned)run_bytes;
	constexpr unsigned OOBV = 0x80000000u;                              // + any scalar offset < 2^31: beyond every span, no wrap
	const __amdgpu_buffer_rsrc_t rC = pixel_rsrc(Cp, (int64_t)minpix * ds, FAR ? 0 : span);
	const __amdgpu_buffer_rsrc_t rA = pixel_rsrc(NACC >= 1 ? Ain : Cp, (int64_t)minpix * ds, FAR ? 0 : span);
	const __amdgpu_buffer_rsrc_t rA2 = pixel_rsrc(NACC >= 2 ? Ain2 : Cp, (int64_t)minpix * ds, FAR ? 0 : span);
	const __amdgpu_buffer_rsrc_t rO = pixel_rsrc(Out, (int64_t)minpix * ds, FAR ? 0 : span);
	const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void *)win, 0, H * Wm + 2 * WBIAS + 64, 0x00020000);
	const int w0 = y0s * Wm + x0s, dw = sy * Wm + sx;
	const __amdgpu_buffer_rsrc_t rCls = __builtin_amdgcn_make_buffer_rsrc((void *)cls0, 0, H * W, 0x00020000);
	const __amdgpu_buffer_rsrc_t rDisp = pixel_rsrc(ARGMIN ? Disp : Out, 0, H * W * 4);
	unsigned lane_off[VEC ? VPL / 4 : VPL];                             // byte offset of this lane's values inside a run, or OOBV
#pragma unroll
	for (int q = 0; q < (VEC ? VPL / 4 : VPL); ++q) {
		const unsigned b = (unsigned)(dbase + (VEC ? 4 * q : q)) * 4u;
		lane_off[q] = (FAR || b < (unsigned)run_bytes) ? b : OOBV;
	}
	const unsigned lane0_off = lane == 0 ? 0u : OOBV;

	// rs: the volume's descriptor (FAR: ignored), so: the pixel's scalar byte offset (FAR: ignored)
	auto load_run = [&](float (&dst)[VPL], const float *base, const __amdgpu_buffer_rsrc_t &rs, unsigned so, int64_t pix) {
		const __amdgpu_buffer_rsrc_t r = FAR ? pixel_rsrc(base, pix * ds, run_bytes) : rs;
		const unsigned soff = FAR ? 0u : so;
		if (VEC) {
#pragma unroll
			for (int q = 0; q < VPL / 4; ++q) {
				const uint4v t = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off[q], soff, MC_SGM_VOL_AUX);
				dst[4 * q + 0] = __uint_as_float(t.x); dst[4 * q + 1] = __uint_as_float(t.y);
				dst[4 * q + 2] = __uint_as_float(t.z); dst[4 * q + 3] = __uint_as_float(t.w);
			}
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j) dst[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, lane_off[j], soff, 0));
		}
	};

	auto load_step = [&](StepData<VPL, NACC> &sd, int s) {
		const unsigned so = c0v + (unsigned)s * c1v;
		const int64_t pix = FAR ? (int64_t)(y0s + s * sy) * W + (x0s + s * sx) : 0;
		load_run(sd.c, Cp, rC, so, pix);
		if constexpr (NACC >= 1) load_run(sd.a, Ain, rA, so, pix);
		if constexpr (NACC >= 2) load_run(sd.a2, Ain2, rA2, so, pix);
		unsigned pk = 0;
		const unsigned sw = (unsigned)(w0 + s * dw);
#pragma unroll
		for (int q = 0; q < VPL / 4; ++q) pk |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rW, woff + q * wq, sw, 0) << (8 * q);
		sd.pk = pk;
		sd.a0 = __builtin_amdgcn_raw_buffer_load_b8(rCls, 0, (unsigned)(pix0 + s * dp), 0);  // every lane reads the same byte
	};

	auto store_step = [&](const float (&o)[VPL], int s) {
		const int64_t pix = FAR ? (int64_t)(y0s + s * sy) * W + (x0s + s * sx) : 0;
		const __amdgpu_buffer_rsrc_t r = FAR ? pixel_rsrc(Out, pix * ds, run_bytes) : rO;
		const unsigned soff = FAR ? 0u : c0v + (unsigned)s * c1v;
		if (VEC) {
#pragma unroll
			for (int q = 0; q < VPL / 4; ++q) {
				uint4v t;
				t.x = __float_as_uint(o[4 * q + 0]); t.y = __float_as_uint(o[4 * q + 1]);
				t.z = __float_as_uint(o[4 * q + 2]); t.w = __float_as_uint(o[4 * q + 3]);
				__builtin_amdgcn_raw_buffer_store_b128(t, r, lane_off[q], soff, MC_SGM_ST_AUX);
			}
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[j]), r, lane_off[j], soff, 0);
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
			const float mall = wave_min_q(best);
			const unsigned long long cand = __ballot(best == mall && best < INF);
			// branch-free: ffs = 0 when no lane holds a finite minimum (all-NaN pixel -> index 0)
			const int f = __builtin_ffsll((long long)cand);
			const int got = __builtin_amdgcn_readlane(bi, (f - 1) & 63);
			const int idx = f ? got : 0;
			// one float per pixel, the pixel in the scalar offset: only lane 0 carries an in-range offset
			__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)idx), rDisp, lane0_off, (unsigned)(pix0 + s * dp) * 4u, 0);
		}
	};

	float prev[VPL];
	float m = 0.0f;
#pragma unroll
	for (int j = 0; j < VPL; ++j) prev[j] = 0.0f;

	// loop-invariant VALU operands, pinned in VGPRs (the compiler re-materialises them from SGPRs in every step otherwise)
	float P1midv = P1mid, P2midv = P2mid, P1amidv = P1amid, INFv = INF;
	asm volatile("" : "+v"(P1midv), "+v"(P2midv), "+v"(P1amidv), "+v"(INFv));
	// MODE 0 writes 0 + L_r (out:zero() then +=, main.lua:1014), its concurrent second direction L_r itself: x + (+0) and x + (-0)
	const float zadd = (DUAL && second) ? -0.0f : 0.0f;

	// keep != nullptr: the step's outputs go there instead of to memory (the horizontal sweeps' batched stores below)
	auto process = [&](const StepData<VPL, NACC> &sd, int s, float *keep = nullptr) {
		float val[VPL], o[VPL];
		const int a0 = __builtin_amdgcn_readfirstlane((int)sd.a0);
		const int amatch = a0 == 1 ? 3 : a0;
		// scalar selects, as written: the compiler sends a select of two float SGPRs to the VALU (two moves and a v_cndmask each)
		float P1x, P2x, P1ax = 0.0f;
		if (DIRN >= 2)
			asm("s_cmp_eq_u32 %3, 0\n\ts_cselect_b32 %0, %4, %5\n\ts_cselect_b32 %1, %6, %7\n\ts_cselect_b32 %2, %8, %9"
			    : "=s"(P1x), "=s"(P2x), "=s"(P1ax)
			    : "s"(a0), "s"(P1lo), "s"(P1hi), "s"(P2lo), "s"(P2hi), "s"(P1alo), "s"(P1ahi)
			    : "scc");
		else
			asm("s_cmp_eq_u32 %2, 0\n\ts_cselect_b32 %0, %3, %4\n\ts_cselect_b32 %1, %5, %6"
			    : "=s"(P1x), "=s"(P2x)
			    : "s"(a0), "s"(P1lo), "s"(P1hi), "s"(P2lo), "s"(P2hi)
			    : "scc");
		const float down = lane_from_below(prev[VPL - 1], INF);
		const float up = lane_from_above(prev[0], INF);
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			const int b = (sd.pk >> (2 * j)) & 3;
			const bool match = b == amatch;
			const float P2 = match ? P2x : P2midv;
			const float P1 = match ? P1x : P1midv;
			const float P1a = match ? P1ax : P1amidv;
			const float pm = j > 0 ? prev[j > 0 ? j - 1 : 0] : down;
			const float pp = j < VPL - 1 ? prev[j < VPL - 1 ? j + 1 : 0] : up;
			// adcensus.cu:607-613: min(min(min(prev, m + P2), pm + P1), pp + P1) as one v_min_f32 + v_min3_f32
			float cost;
			asm("v_min_f32 %0, %1, %2\n\tv_min3_f32 %0, %0, %3, %4"
			    : "=&v"(cost)
			    : "v"(prev[j]), "v"(m + P2), "v"(pm + (DIRN == 2 ? P1a : P1)), "v"(pp + (DIRN == 3 ? P1a : P1)));
			val[j] = (sd.c[j] + cost) - m;  // adcensus.cu:615
		}
		if (s == 0) {   // border branch, adcensus.cu:567-572: L_r = C.  A uniform branch taken once per line, not VPL selects per step
			asm volatile("; border step");
#pragma unroll
			for (int j = 0; j < VPL; ++j) val[j] = sd.c[j];
		}
		float q[VPL];
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			if (MODE == 0) o[j] = val[j] + zadd;
			else if (MODE == 1) o[j] = sd.a[j] + val[j];
			else if (MODE == 2) o[j] = (sd.a[j] + val[j]) * 0.25f;
			else o[j] = (sd.a[j] + sd.a2[j]) + val[j];
			q[j] = vmin2(val[j], INFv);                       // NaN -> +INF: fminf semantics of the recurrence
		}
#pragma unroll
		for (int j = 0; j < VPL; ++j) prev[j] = (dbase + j < D) ? q[j] : INFv;
		float nm = vmin3(prev[0], prev[1], prev[2]);
#pragma unroll
		for (int j = 3; j < VPL; j += 2) nm = j + 1 < VPL ? vmin3(nm, prev[j], prev[j + 1]) : vmin2(nm, prev[j]);
		m = wave_min_q(nm);
		if (keep) {
#pragma unroll
			for (int j = 0; j < VPL; ++j) keep[j] = o[j];
		} else {
			store_step(o, s);
		}
	};

	StepData<VPL, NACC> ring[U];
#pragma unroll
	for (int u = 0; u < U; ++u) load_step(ring[u], u < last ? u : last);

	int g = 0;
	// steady state: straight-line code, every slot is consumed and immediately refilled U steps ahead
	// (refills past the end of the line re-read its last pixel; they are never consumed)
	// Horizontal sweeps (SB / LB > 1): consecutive steps of a line are consecutive pixels, i.e. CONTIGUOUS memory -- the outputs of SB steps
	// are stored together (SB runs back to back: one piece of SB x ds floats instead of SB pieces a step's recurrence apart) and the refills of LB slots requested together
	// (A/B on one box, profiles/r05_ab_sgm_batched.txt, the horizontal launch at KITTI size: 772 us at 1 / 1, 761 at SB 4, 751 at LB 4, 746 at 4 / 4, 751 at 8 / 8; 1000 x 1500: 2 470 -> 2 462)
	constexpr int SB = (DIRN <= 1 && !ARGMIN && U % 4 == 0) ? 4 : 1;
	constexpr int LB = (DIRN <= 1 && U % 4 == 0) ? 4 : 1;
	float obuf[SB][VPL];
	for (; g + U <= nsteps; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			// keep each step's work behind its own s_waitcnt: without the barrier the machine scheduler hoists the
			// recurrence-independent adds of ALL ring slots to the loop head, i.e. waits for every prefetch at once
			__builtin_amdgcn_sched_barrier(0);
			if (SB > 1) {
				process(ring[u], s, obuf[u % SB]);
				if (u % SB == SB - 1) {
#pragma unroll
					for (int k = 0; k < SB; ++k) store_step(obuf[k], s - (SB - 1) + k);
				}
			} else {
				process(ring[u], s);
			}
			if (u % LB == LB - 1) {
#pragma unroll
				for (int k = 0; k < LB; ++k) {
					const int sn = s - (LB - 1) + k + U;
					load_step(ring[u - (LB - 1) + k], sn < last ? sn : last);
				}
			}
		}
	}
#pragma unroll
	for (int u = 0; u < U; ++u) {
		if (g + u < nsteps) process(ring[u], g + u);
	}
}

template <int DIRN, int VPL, int MODE, bool ARGMIN, bool VEC, int U, bool DUAL, bool FAR>
__global__ void __launch_bounds__(256) sgm_pass_kernel(const SgmPassArgs A)
{
	sgm_line<DIRN, VPL, MODE, ARGMIN, VEC, U, DUAL, FAR>(A, __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
}

// ---------------------------------------------------------------------------

// mc_sgm2's contract on a caller's volume (debug aid): NaNs form a tail in d and d = 0 is finite
__global__ void __launch_bounds__(256) sgm_contract_kernel(const float *__restrict__ vol, int64_t pixels, int D, unsigned *__restrict__ count)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool bad = false;
	if (p < pixels) {
		const float *v = vol + p * D;
		bad = !(fabsf(v[0]) < __builtin_inff());
		bool seen_nan = false;
		for (int d = 0; d < D; ++d) {
			const bool isn = v[d] != v[d];
			bad = bad || (seen_nan && !isn);
			seen_nan = seen_nan || isn;
		}
	}
	const unsigned long long b = __ballot(bad);
	if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned)__builtin_popcountll(b));
}

int sgm_contract_violations(const float *vol, int H, int W, int D, unsigned *count, hipStream_t st)
{
	const int64_t pixels = (int64_t)H * W;
	hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned), st);
	if (e != hipSuccess) {
		set_error("mc_sgm2_contract_violations: hipMemsetAsync: %s", hipGetErrorString(e));
		return (int)e;
	}
	hipLaunchKernelGGL(sgm_contract_kernel, dim3(cdiv(pixels, 256)), dim3(256), 0, st, vol, pixels, D, count);
	return check_launch("sgm_contract");
}

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

// steps kept in flight per wave and sweep (see launch_pass)
constexpr int SGM_U_H = 8, SGM_U_UP = 16;
constexpr int SGM_U_DOWN = 16;   // (A/B on one box: profiles/r05_ab_sgm_depth_per_launch.txt -- 16 steps in flight cost 236 VGPRs, i.e. two waves per SIMD)

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
	const int U = DIRN == 2 ? SGM_U_DOWN : (DIRN == 3 ? SGM_U_UP : SGM_U_H);
#define MC_SGM_GO(VPL_, VEC_, U_) \
	hipLaunchKernelGGL((sgm_pass_kernel<DIRN, VPL_, MODE, ARGMIN, VEC_, U_, DUAL, false>), grid, block, 0, st, A)
#define MC_SGM_GO_FAR(VPL_, VEC_, U_) \
	hipLaunchKernelGGL((sgm_pass_kernel<DIRN, VPL_, MODE, ARGMIN, VEC_, U_, DUAL, true>), grid, block, 0, st, A)
	// a volume of 2 GiB or more: 64-bit pixel addresses in every step (one prefetch depth only)
	const bool far = (int64_t)A.H * A.W * A.ds * 4 >= ((int64_t)1 << 31);
	if (far) {
		if (A.D <= 256) { if (vec) MC_SGM_GO_FAR(4, true, 4); else MC_SGM_GO_FAR(4, false, 4); }
		else { if (vec) MC_SGM_GO_FAR(8, true, 4); else MC_SGM_GO_FAR(8, false, 2); }
	} else if (A.D <= 256) {
		if (vec) { if (U == 16) MC_SGM_GO(4, true, 16); else if (U == 8) MC_SGM_GO(4, true, 8); else MC_SGM_GO(4, true, 4); }
		else MC_SGM_GO(4, false, 4);
	} else {
		if (vec) { if (U >= 8) MC_SGM_GO(8, true, 8); else MC_SGM_GO(8, true, 4); }
		else MC_SGM_GO(8, false, 2);
	}
#undef MC_SGM_GO_FAR
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
	// 32-bit pixel indices and line strides in the sweeps (a volume may still exceed 4 GiB: the FAR instances)
	MC_REQUIRE((int64_t)H * W < ((int64_t)1 << 29) && (int64_t)W * ds * 4 < ((int64_t)1 << 31), "sgm: image %dx%d (pixel stride %d) exceeds the sweeps' 32-bit line arithmetic", H, W, ds);
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
