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
				const uint4v t = __builtin_amdgcn_raw_buffer_load_b128(r, (dbase + 4 * q) * 4, 0, 0);
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
				__builtin_amdgcn_raw_buffer_store_b128(t, r, (dbase + 4 * q) * 4, 0, 0);
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

// =====================================================================================================
// Fused sweeps: 7.x V of traffic instead of 11 V
// =====================================================================================================
// The four directional recurrences need C four times and the running sum three more times when every direction is its
// own pass.  Two of them can share a pass if the pixels are visited on a diagonal wavefront:
//   forward  sweep: wave <-> image row y, walking x = 0..W-1, computes L_right(y,x) from its own previous step AND
//                   L_down(y,x) from L_down(y-1,x), which the wave of row y-1 produced when it passed column x.
//                   Writes P = 0 + L_right and Q = L_down.                                     (reads V, writes 2 V)
//   backward sweep: rows from the bottom, x = W-1..0: L_left from its own state, L_up from row y+1.
//                   Writes (((P + L_left) + Q) + L_up) / 4 over P and the arg-min.             (reads 3 V, writes V)
// The reference's summation order ((0+L_r)+L_l)+L_d)+L_u (adcensus.cu:639-693 launches right, left, down, up and every
// launch does out += L) is kept, so results stay bit-identical.
// Row y can only be one step behind row y-1, so all rows of a volume are in flight at once, staggered.  Rows are
// grouped G per workgroup: inside a group L_down travels through an LDS ring (slot = one pixel's D values) with LDS
// progress counters; between groups it travels through global memory -- for the forward sweep that is simply Q itself
// (the last row of a group stores it write-through, sc0 sc1), for the backward sweep a small boundary buffer -- with
// one agent-scope progress word per row, published U steps late (in-order vmcnt: the stores of step s have completed
// once the loads issued after them have been waited for) and polled by the consumer through inline asm (so that
// hipcc's vmcnt bookkeeping of the prefetch ring stays exact).  Every spin is bounded: on timeout the error word is
// set and the wave carries on with whatever data it finds (the host checks the word: the result is then rejected).
// Dispatch order is the only liveness assumption: workgroup b needs b-1 to have started, which in-order dispatch gives
// (at most nvol*ceil(H/G) workgroups, i.e. <= 256 at every BASELINE size, all resident).
struct SgmFusedArgs {
	const float *C[2];
	float *P[2];          // forward: written; backward: read, then overwritten with the result
	float *Q[2];          // forward: written; backward: read
	float *Bnd[2];        // backward: boundary rows of L_up, [ngroups][W][ds]
	float *disp[2];
	int direction[2];
	int nvol, H, W, D, ds;
	const uint8_t *cls0;
	int64_t cls_plane;
	const uint8_t *win;
	int Wm;
	float P1[3], P2[3], P1a[3];
	int *gprog;           // [2 phases][nvol][H] progress words, zeroed by the host before the call
	int *err;             // set to 1 if a bounded spin timed out
	int ngroups;
};

constexpr int SGF_R = 4;          // LDS ring slots per row hand-off
constexpr int SGF_SPIN = 1 << 20; // bound of every spin loop

__device__ __forceinline__ int poll_global(const int *p)
{
	int v;
	asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	return v;
}

template <int PHASE, int VPL, int G, int U>
__global__ void __launch_bounds__(G * 64) sgm_fused_kernel(const SgmFusedArgs A)
{
	extern __shared__ __attribute__((aligned(16))) float sgf_lds[];
	// layout: ring[G][SGF_R][64*VPL] floats (dynamic); progress words in their own static LDS array so that the
	// compiler keeps them in the LDS address space (a pointer carved out of the dynamic block decays to flat)
	float *ringbase = sgf_lds;
	__shared__ int prog[G];  // accessed with relaxed workgroup-scope atomics (volatile accesses would stay flat)

	const int lane = threadIdx.x & 63;
	const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int H = A.H, W = A.W, D = A.D, ds = A.ds, Wm = A.Wm;
	const int b = blockIdx.x;
	const int v = b / A.ngroups, g = b - v * A.ngroups;
	const int lr = g * G + w;                       // logical row: 0 is the first row of the sweep
	if (lane == 0) __hip_atomic_store(&prog[w], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	__syncthreads();
	if (lr >= H) return;
	const int y = PHASE == 0 ? lr : H - 1 - lr;
	const int direction = A.direction[v];
	const float *__restrict__ Cp = A.C[v];
	float *__restrict__ Pp = A.P[v];
	float *__restrict__ Qp = A.Q[v];
	float *__restrict__ Bp = A.Bnd[v];
	float *__restrict__ Disp = A.disp[v];
	constexpr int DH = PHASE == 0 ? 0 : 1;  // horizontal direction index (right / left)
	constexpr int DV = PHASE == 0 ? 2 : 3;  // vertical direction index (down / up)
	const bool has_prod = lr > 0;                    // a row above (in sweep order) exists
	const bool prod_lds = has_prod && w > 0;         // ... in this workgroup
	const bool prod_glb = has_prod && w == 0;        // ... in the previous workgroup
	const bool has_cons = lr + 1 < H;
	const bool cons_lds = has_cons && w + 1 < G;
	const bool cons_glb = has_cons && w + 1 == G;
	int *gp_mine = A.gprog + ((size_t)PHASE * A.nvol + v) * H + lr;
	const int *gp_prod = gp_mine - 1;

	const int dbase = VPL * lane;
	const uint8_t *__restrict__ clsH = A.cls0 + (int64_t)DH * A.cls_plane;
	const uint8_t *__restrict__ clsV = A.cls0 + (int64_t)DV * A.cls_plane;
	constexpr int WBIAS = 1024;
	const uint8_t *__restrict__ winH = A.win + ((int64_t)((direction > 0 ? 0 : 1) * 4 + DH) * H) * Wm + SGM_PADW - WBIAS;
	const uint8_t *__restrict__ winV = A.win + ((int64_t)((direction > 0 ? 0 : 1) * 4 + DV) * H) * Wm + SGM_PADW - WBIAS;
	const int woff = (direction > 0 ? dbase : -dbase - 3) + WBIAS;
	const int run_bytes = ds * 4;
	const float INF = __builtin_inff();
	const float P1mid = A.P1[1], P2mid = A.P2[1], P1amid = A.P1a[1];
	const float P1lo = A.P1[0], P2lo = A.P2[0], P1alo = A.P1a[0];
	const float P1hi = A.P1[2], P2hi = A.P2[2], P1ahi = A.P1a[2];

	// The prefetch ring is loaded from inline asm: with the spin loops in every step hipcc's s_waitcnt model degrades to
	// vmcnt(0) at each use, i.e. it would wait for the loads it has just issued.  Hidden from the compiler, the ring is
	// waited for by hand with a counted vmcnt (vmcnt is in-order on gfx9): see land().
	static_assert(VPL == 4, "the fused sweeps cover D <= 256 (one dwordx4 per lane)");
	struct Step {
		uint4v c, p, q, h;  // cost, P, Q (backward only), hand-off from the previous workgroup
		unsigned pkH, pkV, a0H, a0V;
	};
	const unsigned vo16 = (unsigned)dbase * 4u;   // this lane's byte offset inside a pixel's run
	const unsigned vzero = 0u;
	auto ld128 = [&](uint4v &dst, const float *base, int64_t elem_off, bool through) {
		const __amdgpu_buffer_rsrc_t r = pixel_rsrc(base, elem_off, run_bytes);
		if (through) {
			asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen sc0 sc1" : "=v"(dst) : "v"(vo16), "s"(r) : "memory");
		} else {
			asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(vo16), "s"(r) : "memory");
		}
	};
	auto ld8 = [&](unsigned &dst, const uint8_t *base, unsigned voffset) {
		const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 2 * 1024 + 64, 0x00020000);
		asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voffset), "s"(r) : "memory");
	};
	auto store_run = [&](const float (&src)[VPL], float *base, int64_t elem_off, bool through) {
		const __amdgpu_buffer_rsrc_t r = pixel_rsrc(base, elem_off, run_bytes);
		uint4v t;
		t.x = __float_as_uint(src[0]); t.y = __float_as_uint(src[1]);
		t.z = __float_as_uint(src[2]); t.w = __float_as_uint(src[3]);
		if (through) __builtin_amdgcn_raw_buffer_store_b128(t, r, dbase * 4, 0, 17);
		else __builtin_amdgcn_raw_buffer_store_b128(t, r, dbase * 4, 0, 0);
	};
	auto xof = [&](int s) { return PHASE == 0 ? s : W - 1 - s; };

	bool dead = false;   // a spin timed out: stop waiting (the error word is set, the result will be rejected)
	int known_glb = 0;   // progress of the producing row in the previous workgroup, as last seen
	// VMEM operations every wave issues per step, in this order: ring loads (LOADS), then the step's stores (>= STORES).
	// Waves at a workgroup boundary issue one more of each; counting the minimum only makes land() wait a little longer.
	constexpr int LOADS = (PHASE == 0 ? 1 : 3) + 4;
	constexpr int STORES = PHASE == 0 ? 2 : 1;
	auto load_step = [&](Step &sd, int s) {
		const int x = xof(s);
		const int64_t pix = (int64_t)y * W + x;
		ld128(sd.c, Cp, pix * ds, false);
		if (PHASE == 1) {
			ld128(sd.p, Pp, pix * ds, false);
			ld128(sd.q, Qp, pix * ds, false);
		}
		if (prod_glb) {  // wave-uniform
			if (known_glb < s + 1 && !dead) {
				int spins = 0;
				while ((known_glb = poll_global(gp_prod)) < s + 1) {
					__builtin_amdgcn_s_sleep(8);
					if (++spins > SGF_SPIN) { *A.err = 1; dead = true; break; }
				}
			}
			if (PHASE == 0) ld128(sd.h, Qp, ((int64_t)(y - 1) * W + x) * ds, true);
			else ld128(sd.h, Bp, ((int64_t)(g - 1) * W + x) * ds, true);
		}
		ld8(sd.pkH, winH + (int64_t)y * Wm + x, (unsigned)woff);
		ld8(sd.pkV, winV + (int64_t)y * Wm + x, (unsigned)woff);
		ld8(sd.a0H, clsH + pix, vzero);
		ld8(sd.a0V, clsV + pix, vzero);
	};
	// the loads of the step about to be processed have landed once at most (U-1) steps' worth of younger operations
	// are outstanding
	// mode 2: steady state -- after this step's loads came U-1 x (stores of a step + loads of a step);
	// mode 1: the first U steps -- only the U-1 later prologue loads are certain to have followed;
	// mode 0: tail -- no younger loads any more.
	auto land = [&](Step &sd, int mode) {
		constexpr int N2 = (U - 1) * (LOADS + STORES) < 63 ? (U - 1) * (LOADS + STORES) : 63;
		constexpr int N1 = (U - 1) * LOADS < 63 ? (U - 1) * LOADS : 63;
		if (mode == 2) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N2) : "memory");
		else if (mode == 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N1) : "memory");
		else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
		asm volatile("" : "+v"(sd.c), "+v"(sd.p), "+v"(sd.q), "+v"(sd.h), "+v"(sd.pkH), "+v"(sd.pkV), "+v"(sd.a0H), "+v"(sd.a0V));
	};

	// one step of the recurrence (adcensus.cu:574-615) for a whole pixel: val = (C + cost) - m
	auto recur = [&](float (&val)[VPL], const float (&c)[VPL], const float (&prev)[VPL], float m, unsigned pk, unsigned a0v,
	                 bool alpha_minus, bool alpha_plus) {
		const int a0 = __builtin_amdgcn_readfirstlane((int)a0v);
		const int amatch = a0 == 1 ? 3 : a0;
		const float P1x = a0 == 0 ? P1lo : P1hi, P2x = a0 == 0 ? P2lo : P2hi, P1ax = a0 == 0 ? P1alo : P1ahi;
		const float down = lane_from_below(prev[VPL - 1], INF);
		const float up = lane_from_above(prev[0], INF);
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			const int bcls = (pk >> (2 * j)) & 3;
			const bool match = bcls == amatch;
			const float P2 = match ? P2x : P2mid;
			const float P1 = match ? P1x : P1mid;
			const float P1a = match ? P1ax : P1amid;
			const float pm = j > 0 ? prev[j > 0 ? j - 1 : 0] : down;
			const float pp = j < VPL - 1 ? prev[j < VPL - 1 ? j + 1 : 0] : up;
			float cost = fminf(prev[j], m + P2);
			cost = fminf(cost, pm + (alpha_minus ? P1a : P1));
			cost = fminf(cost, pp + (alpha_plus ? P1a : P1));
			val[j] = (c[j] + cost) - m;
		}
	};
	auto clean_min = [&](float (&dst)[VPL], const float (&src)[VPL]) {  // NaN / beyond D -> +INF; returns the wave minimum
		float nm = INF;
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			dst[j] = (dbase + j < D) ? fminf(src[j], INF) : INF;
			nm = fminf(nm, dst[j]);
		}
		return wave_min(nm);
	};

	float prevH[VPL];
	float mH = 0.0f;
#pragma unroll
	for (int j = 0; j < VPL; ++j) prevH[j] = 0.0f;
	int known_prod = 0, known_cons = 0;  // LDS progress of the neighbouring rows, as last seen
	float *ring_in = ringbase + (size_t)(w > 0 ? w - 1 : 0) * SGF_R * 64 * VPL;  // written by row w-1
	float *ring_out = ringbase + (size_t)w * SGF_R * 64 * VPL;                      // read by row w+1

	auto process = [&](const Step &sd, int s) {
		const int x = xof(s);
		const int64_t pix = (int64_t)y * W + x;
		float sc[VPL] = {__uint_as_float(sd.c.x), __uint_as_float(sd.c.y), __uint_as_float(sd.c.z), __uint_as_float(sd.c.w)};
		// ---- vertical direction: previous row's L at this column ----
		float vprev[VPL];
		if (prod_lds) {
			if (known_prod < s + 1 && !dead) {
				int spins = 0;
				while ((known_prod = __hip_atomic_load(&prog[w - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < s + 1) {
					__builtin_amdgcn_s_sleep(1);
					if (++spins > SGF_SPIN) { *A.err = 1; dead = true; break; }
				}
			}
			asm volatile("" ::: "memory");  // the slot is read only after the flag said it is there (LDS executes in order)
			const float4 t = *reinterpret_cast<const float4 *>(ring_in + ((size_t)(s % SGF_R) * 64 + lane) * VPL);
			vprev[0] = t.x; vprev[1] = t.y; vprev[2] = t.z; vprev[3] = t.w;
		} else {
			vprev[0] = __uint_as_float(sd.h.x); vprev[1] = __uint_as_float(sd.h.y);
			vprev[2] = __uint_as_float(sd.h.z); vprev[3] = __uint_as_float(sd.h.w);
		}
		float valV[VPL], valH[VPL];
		if (has_prod) {
			float vclean[VPL];
			const float mV = clean_min(vclean, vprev);
			recur(valV, sc, vclean, mV, sd.pkV, sd.a0V, DV == 2, DV == 3);
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j) valV[j] = sc[j];  // first row of the sweep: L = C (adcensus.cu:567-572)
		}
		// ---- horizontal direction ----
		{
			float rec[VPL];
			recur(rec, sc, prevH, mH, sd.pkH, sd.a0H, false, false);
#pragma unroll
			for (int j = 0; j < VPL; ++j) valH[j] = s == 0 ? sc[j] : rec[j];
		}
		mH = clean_min(prevH, valH);
		// ---- hand L_vertical(y, x) to the next row ----
		if (cons_lds) {
			if (known_cons + SGF_R < s + 1 && !dead) {  // slot s % R still holds step s - R until the consumer has finished it
				int spins = 0;
				while ((known_cons = __hip_atomic_load(&prog[w + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) + SGF_R < s + 1) {
					__builtin_amdgcn_s_sleep(1);
					if (++spins > SGF_SPIN) { *A.err = 1; dead = true; break; }
				}
			}
			asm volatile("" ::: "memory");
			*reinterpret_cast<float4 *>(ring_out + ((size_t)(s % SGF_R) * 64 + lane) * VPL) = make_float4(valV[0], valV[1], valV[2], valV[3]);
		}
		// ---- outputs ----
		if (PHASE == 0) {
			float o[VPL];
#pragma unroll
			for (int j = 0; j < VPL; ++j) o[j] = 0.0f + valH[j];
			store_run(o, Pp, pix * ds, false);
			store_run(valV, Qp, pix * ds, cons_glb);
		} else {
			const float sp[VPL] = {__uint_as_float(sd.p.x), __uint_as_float(sd.p.y), __uint_as_float(sd.p.z), __uint_as_float(sd.p.w)};
			const float sq[VPL] = {__uint_as_float(sd.q.x), __uint_as_float(sd.q.y), __uint_as_float(sd.q.z), __uint_as_float(sd.q.w)};
			float o[VPL];
#pragma unroll
			for (int j = 0; j < VPL; ++j) o[j] = (((sp[j] + valH[j]) + sq[j]) + valV[j]) * 0.25f;
			store_run(o, Pp, pix * ds, false);
			if (cons_glb) store_run(valV, Bp, ((int64_t)g * W + x) * ds, true);
			if (Disp) {
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
				const int f = __builtin_ffsll((long long)cand);
				const int got = __builtin_amdgcn_readlane(bi, (f - 1) & 63);
				const int idx = f ? got : 0;
				const __amdgpu_buffer_rsrc_t rd = pixel_rsrc(Disp, pix, 4);
				__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)idx), rd, lane * 4, 0, 0);
			}
		}
		// ---- publish progress ----
		if (prod_lds || cons_lds) {
			// LDS operations of one wave are serviced in order, so the ring write above is visible before this flag;
			// only the compiler must be kept from reordering them (a release fence would also drain vmcnt)
			asm volatile("" ::: "memory");
			if (lane == 0) __hip_atomic_store(&prog[w], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
		if (cons_glb && s >= U) {
			// the stores of step s-U were issued before the loads of step s, which process() has just consumed
			if (lane == 0) __hip_atomic_store(gp_mine, s - U + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};

	// The ring is filled by the SAME asm statements that refill it in steady state (the loop starts U steps early with
	// processing switched off): a separate prologue would give every slot a second definition, and the register copies
	// the compiler may place where the two meet would read a slot before its hidden load has landed.
	Step ring[U];
#pragma unroll
	for (int u = 0; u < U; ++u) {
		ring[u].c = ring[u].p = ring[u].q = ring[u].h = uint4v{0u, 0u, 0u, 0u};
		ring[u].pkH = ring[u].pkV = ring[u].a0H = ring[u].a0V = 0u;
	}
	const int last = W - 1;
	int gidx = -U;
	for (; gidx + U <= W; gidx += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = gidx + u;
			__builtin_amdgcn_sched_barrier(0);
			if (s >= 0) {
				land(ring[u], gidx > 0 ? 2 : 1);
				process(ring[u], s);
			}
			const int sn = s + U;
			load_step(ring[u], sn < last ? sn : last);
		}
	}
#pragma unroll
	for (int u = 0; u < U; ++u) {
		if (gidx + u < W) {
			land(ring[u], 0);
			process(ring[u], gidx + u);
		}
	}
	if (cons_glb) {  // everything stored: publish the whole row
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (lane == 0) __hip_atomic_store(gp_mine, W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

static int sgm_depth(bool horizontal, int deflt)
{
	// MC_SGM_UH / MC_SGM_UV = <4|8|16> override the prefetch depth of the horizontal / vertical sweeps (tuning aid)
	static const int fh = [] { const char *e = getenv("MC_SGM_UH"); return e ? atoi(e) : 0; }();
	static const int fv = [] { const char *e = getenv("MC_SGM_UV"); return e ? atoi(e) : 0; }();
	const int f = horizontal ? fh : fv;
	return (f == 4 || f == 8 || f == 16) ? f : deflt;
}

template <int DIRN, int MODE, bool ARGMIN, bool DUAL>
static void launch_pass(const SgmPassArgs &A, bool vec, hipStream_t st)
{
	const int nlines = DIRN <= 1 ? A.H : A.W;
	const int waves = A.nvol * nlines * (DUAL ? 2 : 1);
	const dim3 grid(cdiv(waves, 4)), block(256);
	// fewer waves than SIMDs (1024): nothing but prefetch depth hides HBM latency
	const int U = sgm_depth(DIRN <= 1, DIRN <= 1 ? 8 : 4);
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

constexpr int SGF_G = 8, SGF_U = 4;

size_t sgm_fused_scratch_bytes(int nvol, int H, int W, int ds)
{
	const size_t ngroups = (size_t)(H + SGF_G - 1) / SGF_G;
	size_t b = (size_t)nvol * ngroups * W * ds * sizeof(float);  // boundary rows of the backward sweep
	b = (b + 255) & ~(size_t)255;
	b += ((size_t)2 * nvol * H + 64) * sizeof(int);               // progress words + error word
	return (b + 255) & ~(size_t)255;
}

// Fused forward/backward sweeps (see sgm_fused_kernel).  out[] receives the result (what sgm_sweeps' fused mode leaves
// in out[]), out2[] is scratch of the same size (Q), `scratch` holds sgm_fused_scratch_bytes().  Returns -1 when the
// problem is outside what the fused kernels cover (caller then uses the separate sweeps).
int sgm_fused(const float *const C[2], float *const out[2], float *const out2[2], float *const disp[2], const int direction[2],
              int nvol, int H, int W, int D, int ds, const void *maps, void *scratch, float pi1, float pi2, float alpha1, float q1,
              float q2, hipStream_t st)
{
	static const int env_off = [] { const char *e = getenv("MC_SGM_FUSED"); return e ? atoi(e) == 0 : 0; }();
	if (env_off || D > 256 || ds % 4 != 0 || !out2 || !scratch) return -1;
	for (int v = 0; v < nvol; ++v)
		if ((uintptr_t)C[v] % 16 || (uintptr_t)out[v] % 16 || (uintptr_t)out2[v] % 16) return -1;
	const int ngroups = (H + SGF_G - 1) / SGF_G;
	if (nvol * ngroups > 256) return -1;  // every workgroup must be resident (one per CU)
	SgmFusedArgs A;
	const size_t bnd_per_vol = (size_t)ngroups * W * ds;
	float *bnd = (float *)scratch;
	size_t off = ((size_t)nvol * bnd_per_vol * sizeof(float) + 255) & ~(size_t)255;
	int *words = (int *)((char *)scratch + off);
	for (int v = 0; v < 2; ++v) {
		const int k = v < nvol ? v : 0;
		A.C[v] = C[k];
		A.P[v] = out[k];
		A.Q[v] = out2[k];
		A.Bnd[v] = bnd + (size_t)k * bnd_per_vol;
		A.disp[v] = disp ? disp[k] : nullptr;
		A.direction[v] = direction[k];
	}
	A.nvol = nvol; A.H = H; A.W = W; A.D = D; A.ds = ds;
	A.cls0 = (const uint8_t *)maps;
	A.cls_plane = (int64_t)(((size_t)H * W + 3 + 4) / 4 * 4);
	A.win = A.cls0 + 4 * A.cls_plane;
	A.Wm = W + 2 * SGM_PADW;
	A.P1[0] = pi1; A.P2[0] = pi2;
	A.P1[1] = pi1 / q1; A.P2[1] = pi2 / q1;
	A.P1[2] = pi1 / (q1 * q2); A.P2[2] = pi2 / (q1 * q2);
	for (int k = 0; k < 3; ++k) A.P1a[k] = A.P1[k] / alpha1;
	A.gprog = words;
	A.err = words + (size_t)2 * nvol * H;
	A.ngroups = ngroups;
	const hipError_t e = hipMemsetAsync(words, 0, (size_t)2 * nvol * H * sizeof(int), st);  // progress words; the error word is sticky
	if (e != hipSuccess) {
		set_error("sgm_fused: %s", hipGetErrorString(e));
		return (int)e;
	}
	const size_t lds = (size_t)SGF_G * SGF_R * 64 * 4 * sizeof(float) + SGF_G * sizeof(int) + 64;
	const dim3 grid((unsigned)(nvol * ngroups)), block(SGF_G * 64);
	hipLaunchKernelGGL((sgm_fused_kernel<0, 4, SGF_G, SGF_U>), grid, block, lds, st, A);
	hipLaunchKernelGGL((sgm_fused_kernel<1, 4, SGF_G, SGF_U>), grid, block, lds, st, A);
	return check_launch("sgm_fused");
}

int sgm_fused_clear_error(void *scratch, int nvol, int H, int W, int ds, hipStream_t st)
{
	const size_t ngroups = (size_t)(H + SGF_G - 1) / SGF_G;
	size_t off = ((size_t)nvol * ngroups * W * ds * sizeof(float) + 255) & ~(size_t)255;
	int *err = (int *)((char *)scratch + off) + (size_t)2 * nvol * H;
	const hipError_t e = hipMemsetAsync(err, 0, sizeof(int), st);
	if (e != hipSuccess) {
		set_error("sgm_fused: %s", hipGetErrorString(e));
		return (int)e;
	}
	return 0;
}

// sets disp[0] to NaN if a fused sweep timed out (so that a broken hand-off can never pass for a result)
__global__ void sgm_fused_guard_kernel(const int *err, float *disp, int n)
{
	if (*err) {
		for (int i = threadIdx.x; i < n; i += blockDim.x) disp[i] = __builtin_nanf("");
	}
}

int sgm_fused_guard(const void *scratch, int nvol, int H, int W, int ds, float *disp_out, hipStream_t st)
{
	const size_t ngroups = (size_t)(H + SGF_G - 1) / SGF_G;
	size_t off = ((size_t)nvol * ngroups * W * ds * sizeof(float) + 255) & ~(size_t)255;
	const int *err = (const int *)((const char *)scratch + off) + (size_t)2 * nvol * H;
	hipLaunchKernelGGL(sgm_fused_guard_kernel, dim3(1), dim3(256), 0, st, err, disp_out, H * W);
	return check_launch("sgm_fused_guard");
}

}  // namespace mc
