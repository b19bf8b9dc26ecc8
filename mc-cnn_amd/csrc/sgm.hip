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
	const float *accin[2];// running sum read by MODE>=1
	float *out[2];        // where this sweep writes
	float *disp[2];       // ARGMIN output (H,W), may be null
	int direction[2];
	int nvol;
	int H, W, D, ds;
	const uint8_t *cls0;  // [4][H][W]
	const uint8_t *win;   // [2][4][H][Wm]
	int Wm;
	float P1[3], P2[3], P1a[3];  // 0: both < tau, 1: mixed, 2: both > tau ; P1a = P1 / alpha1
};

__device__ __forceinline__ int cls_of(float v, float tau) { return v < tau ? 0 : (v > tau ? 2 : 1); }

// r: 0 right (dx=1), 1 left (dx=-1), 2 down (dy=1), 3 up (dy=-1)  (adcensus.cu:541-565)
__global__ void __launch_bounds__(256) sgm_prep_kernel(const float *__restrict__ x0, const float *__restrict__ x1,
                                                       uint8_t *__restrict__ cls0, uint8_t *__restrict__ win,
                                                       int H, int W, int Wm, float tau_so)
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
		cls0[((int64_t)r * H + y) * W + x] = (uint8_t)c;
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

template <int VPL>
struct StepData {
	float c[VPL];
	float a[VPL];
	unsigned pk;
	int a0;
};

// MODE 0: out = 0 + L_r          (first direction, no read of the sum)
// MODE 1: out = accin + L_r      (reference-compatible accumulate)
// MODE 2: out = (accin + L_r)/4  (last direction; folds vol:copy(out):div(4), main.lua:1017,1020)
template <int DIRN, int VPL, int MODE, bool ARGMIN, bool VEC, int U>
__global__ void __launch_bounds__(256) sgm_pass_kernel(const SgmPassArgs A)
{
	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	const int H = A.H, W = A.W, D = A.D, ds = A.ds, Wm = A.Wm;
	const int nlines = DIRN <= 1 ? H : W;
	const int nsteps = DIRN <= 1 ? W : H;
	if (wave >= A.nvol * nlines) return;
	const int v = wave / nlines;
	const int line = wave - v * nlines;
	const int direction = A.direction[v];
	const float *__restrict__ Cp = A.C[v];
	const float *__restrict__ Ain = A.accin[v];
	float *__restrict__ Out = A.out[v];

	// line geometry: pixel of step s is (y0 + s*sy, x0 + s*sx)
	const int x0s = DIRN == 0 ? 0 : (DIRN == 1 ? W - 1 : line);
	const int y0s = DIRN == 2 ? 0 : (DIRN == 3 ? H - 1 : line);
	constexpr int sx = DIRN == 0 ? 1 : (DIRN == 1 ? -1 : 0);
	constexpr int sy = DIRN == 2 ? 1 : (DIRN == 3 ? -1 : 0);

	const int dbase = VPL * lane;
	const uint8_t *__restrict__ cls0 = A.cls0 + (int64_t)DIRN * H * W;
	const uint8_t *__restrict__ win = A.win + ((int64_t)((direction > 0 ? 0 : 1) * 4 + DIRN) * H) * Wm + SGM_PADW;
	// window start of chunk q at image column x: direction +1: x + VPL*lane + 4q ; -1: x - VPL*lane - 4q - 3
	const int woff = direction > 0 ? dbase : -dbase - 3;
	const int wq = direction > 0 ? 4 : -4;

	const float INF = __builtin_inff();
	// wave-uniform penalties of the "both classes equal and != mixed" case are picked per step
	const float P1mid = A.P1[1], P2mid = A.P2[1], P1amid = A.P1a[1];

	auto load_step = [&](StepData<VPL> &sd, int s) {
		const int sc = s < nsteps ? s : nsteps - 1;
		const int x = x0s + sc * sx, y = y0s + sc * sy;
		const int64_t pix = (int64_t)y * W + x;
		const int64_t off = pix * ds + dbase;
		if (VEC) {
#pragma unroll
			for (int q = 0; q < VPL / 4; ++q) {
				if (dbase + 4 * q < ds) {  // ds % 4 == 0: the 16-byte chunk lies inside the pixel's run
					const float4 t = *reinterpret_cast<const float4 *>(Cp + off + 4 * q);
					sd.c[4 * q + 0] = t.x; sd.c[4 * q + 1] = t.y; sd.c[4 * q + 2] = t.z; sd.c[4 * q + 3] = t.w;
					if (MODE >= 1) {
						const float4 u = *reinterpret_cast<const float4 *>(Ain + off + 4 * q);
						sd.a[4 * q + 0] = u.x; sd.a[4 * q + 1] = u.y; sd.a[4 * q + 2] = u.z; sd.a[4 * q + 3] = u.w;
					}
				}
			}
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j) {
				if (dbase + j < D) {
					sd.c[j] = Cp[off + j];
					if (MODE >= 1) sd.a[j] = Ain[off + j];
				}
			}
		}
		unsigned pk = 0;
		const uint8_t *wrow = win + (int64_t)y * Wm + x + woff;
#pragma unroll
		for (int q = 0; q < VPL / 4; ++q) pk |= (unsigned)wrow[q * wq] << (8 * q);
		sd.pk = pk;
		sd.a0 = cls0[pix];
	};

	auto store_step = [&](const float (&o)[VPL], int s) {
		const int x = x0s + s * sx, y = y0s + s * sy;
		const int64_t pix = (int64_t)y * W + x;
		const int64_t off = pix * ds + dbase;
		if (VEC) {
#pragma unroll
			for (int q = 0; q < VPL / 4; ++q) {
				if (dbase + 4 * q < ds) {
					*reinterpret_cast<float4 *>(Out + off + 4 * q) =
					    make_float4(o[4 * q + 0], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
				}
			}
		} else {
#pragma unroll
			for (int j = 0; j < VPL; ++j)
				if (dbase + j < D) Out[off + j] = o[j];
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
			int idx = 0;
			if (cand) {
				const int fl = __builtin_ctzll(cand);
				idx = __builtin_amdgcn_readlane(bi, fl);
			}
			if (lane == 0) A.disp[v][pix] = (float)idx;
		}
	};

	float prev[VPL];
	float m;

	auto finish = [&](const float (&val)[VPL], const float (&ain)[VPL], float (&o)[VPL]) {
		float nm = INF;
#pragma unroll
		for (int j = 0; j < VPL; ++j) {
			o[j] = MODE == 0 ? 0.0f + val[j] : (MODE == 1 ? ain[j] + val[j] : (ain[j] + val[j]) * 0.25f);
			prev[j] = (dbase + j < D) ? fminf(val[j], INF) : INF;  // NaN -> +INF: fminf semantics of the recurrence
			nm = fminf(nm, prev[j]);
		}
		m = wave_min(nm);
	};

	StepData<VPL> cur[U], nxt[U];
#pragma unroll
	for (int u = 0; u < U; ++u) load_step(cur[u], u);

	for (int g = 0; g < nsteps; g += U) {
		if (g + U < nsteps) {
#pragma unroll
			for (int u = 0; u < U; ++u) load_step(nxt[u], g + U + u);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			if (s < nsteps) {
				float val[VPL], o[VPL];
				if (s == 0) {
					// border branch, adcensus.cu:567-572: L_r = C
#pragma unroll
					for (int j = 0; j < VPL; ++j) val[j] = cur[u].c[j];
				} else {
					const int a0 = __builtin_amdgcn_readfirstlane(cur[u].a0);
					const int amatch = a0 == 1 ? 3 : a0;
					const float P1x = a0 == 0 ? A.P1[0] : A.P1[2];
					const float P2x = a0 == 0 ? A.P2[0] : A.P2[2];
					const float P1ax = a0 == 0 ? A.P1a[0] : A.P1a[2];
					const float down = lane_from_below(prev[VPL - 1], INF);
					const float up = lane_from_above(prev[0], INF);
#pragma unroll
					for (int j = 0; j < VPL; ++j) {
						const int b = (cur[u].pk >> (2 * j)) & 3;
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
						val[j] = (cur[u].c[j] + cost) - m;  // adcensus.cu:615
					}
				}
				finish(val, cur[u].a, o);
				store_step(o, s);
			}
		}
#pragma unroll
		for (int u = 0; u < U; ++u) cur[u] = nxt[u];
	}
}

// ---------------------------------------------------------------------------

size_t sgm_maps_bytes(int H, int W)
{
	const size_t Wm = (size_t)W + 2 * SGM_PADW;
	size_t b = (size_t)4 * H * W + (size_t)8 * H * Wm;
	return (b + 255) & ~(size_t)255;
}

int sgm_prep(const float *x0, const float *x1, void *maps, int H, int W, float tau_so, hipStream_t st)
{
	const int Wm = W + 2 * SGM_PADW;
	uint8_t *cls0 = (uint8_t *)maps;
	uint8_t *win = cls0 + (size_t)4 * H * W;
	const int64_t total = (int64_t)4 * H * Wm;
	hipLaunchKernelGGL(sgm_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, x0, x1, cls0, win, H, W, Wm, tau_so);
	return check_launch("sgm_prep");
}

template <int DIRN, int MODE, bool ARGMIN>
static void launch_pass(const SgmPassArgs &A, bool vec, hipStream_t st)
{
	const int nlines = DIRN <= 1 ? A.H : A.W;
	const int waves = A.nvol * nlines;
	const dim3 grid(cdiv(waves, 4)), block(256);
	if (A.D <= 256) {
		if (vec) hipLaunchKernelGGL((sgm_pass_kernel<DIRN, 4, MODE, ARGMIN, true, 4>), grid, block, 0, st, A);
		else hipLaunchKernelGGL((sgm_pass_kernel<DIRN, 4, MODE, ARGMIN, false, 4>), grid, block, 0, st, A);
	} else {
		if (vec) hipLaunchKernelGGL((sgm_pass_kernel<DIRN, 8, MODE, ARGMIN, true, 2>), grid, block, 0, st, A);
		else hipLaunchKernelGGL((sgm_pass_kernel<DIRN, 8, MODE, ARGMIN, false, 2>), grid, block, 0, st, A);
	}
}

// Four direction sweeps over nvol (1 or 2) volumes.
//   fused = false: every sweep does out += L_r (adcensus.sgm2 contract, out pre-zeroed by caller)
//   fused = true : sweep 0 writes 0+L_0, sweeps 1,2 accumulate, sweep 3 writes (acc+L_3)/4 and,
//                  if disp[] is set, the argmin of the finished pixel.
int sgm_sweeps(const float *const C[2], float *const out[2], float *const disp[2], const int direction[2], int nvol,
               int H, int W, int D, int ds, const void *maps, float pi1, float pi2, float alpha1, float q1, float q2,
               bool fused, hipStream_t st)
{
	SgmPassArgs A;
	for (int v = 0; v < 2; ++v) {
		const int k = v < nvol ? v : 0;
		A.C[v] = C[k];
		A.accin[v] = out[k];
		A.out[v] = out[k];
		A.disp[v] = disp ? disp[k] : nullptr;
		A.direction[v] = direction[k];
	}
	A.nvol = nvol;
	A.H = H; A.W = W; A.D = D; A.ds = ds;
	A.Wm = W + 2 * SGM_PADW;
	A.cls0 = (const uint8_t *)maps;
	A.win = A.cls0 + (size_t)4 * H * W;
	// adcensus.cu:595-605 -- float divisions exactly as written there
	A.P1[0] = pi1; A.P2[0] = pi2;
	A.P1[1] = pi1 / q1; A.P2[1] = pi2 / q1;
	A.P1[2] = pi1 / (q1 * q2); A.P2[2] = pi2 / (q1 * q2);
	for (int k = 0; k < 3; ++k) A.P1a[k] = A.P1[k] / alpha1;  // adcensus.cu:609,612

	bool vec = (ds % 4 == 0);
	for (int v = 0; v < nvol; ++v) {
		vec = vec && ((uintptr_t)C[v] % 16 == 0) && ((uintptr_t)out[v] % 16 == 0);
	}
	const bool am = fused && disp && disp[0];
	if (!fused) {
		launch_pass<0, 1, false>(A, vec, st);
		launch_pass<1, 1, false>(A, vec, st);
		launch_pass<2, 1, false>(A, vec, st);
		launch_pass<3, 1, false>(A, vec, st);
	} else {
		launch_pass<0, 0, false>(A, vec, st);
		launch_pass<1, 1, false>(A, vec, st);
		launch_pass<2, 1, false>(A, vec, st);
		if (am) launch_pass<3, 2, true>(A, vec, st);
		else launch_pass<3, 2, false>(A, vec, st);
	}
	return check_launch("sgm_pass");
}

}  // namespace mc
