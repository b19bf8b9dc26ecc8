// H*W-sized disparity kernels and layout helpers (gfx950).
// Each kernel cites the adcensus.cu kernel whose arithmetic it reproduces.
#include "mc_common.h"

namespace mc {

// ---- fill / scale / transposes ---------------------------------------------------

__global__ void __launch_bounds__(256) fill_nan_kernel(float *__restrict__ p, int64_t n)
{
	const float nanv = __builtin_nanf("");
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = nanv;
}

__global__ void __launch_bounds__(256) scale_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n, float s)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i] * s;
}
// ... ONE 16-byte element per thread, blocks in address order: the form in which a copy streams fastest on this chip (6.3 TB/s at
// 2 x 2 GB against 5.0 - 5.5 for a grid-stride loop over a few thousand blocks, profiles/r04_bw_sizes.txt)
typedef float f4v_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) scale4_kernel(const f4v_t *__restrict__ in, f4v_t *__restrict__ out, int64_t n4, float s)
{
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n4) out[i] = in[i] * s;
}

// out[c*ldout + r] = in[r*ldin + c] * s for an R x Cn matrix, 64x64 tiles through LDS.
// (D,H,W)->(H,W,ds) is R=D, Cn=H*W, ldin=H*W, ldout=ds; the inverse is R=H*W, Cn=D, ldin=ds, ldout=H*W.
// NT: non-temporal accesses for volumes far larger than the 256 MB MALL (the next kernel cannot find them there anyway)
template <bool NT>
__global__ void __launch_bounds__(256) transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t R,
                                                        int64_t Cn, int64_t ldin, int64_t ldout, float s)
{
	__shared__ float tile[64][65];
	const int64_t c0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	for (int k = ty; k < 64; k += 4) {
		const int64_t r = r0 + k, c = c0 + tx;
		if (r < R && c < Cn) tile[k][tx] = NT ? __builtin_nontemporal_load(in + r * ldin + c) : in[r * ldin + c];
	}
	__syncthreads();
	for (int k = ty; k < 64; k += 4) {
		const int64_t c = c0 + k, r = r0 + tx;
		if (r < R && c < Cn) {
			if (NT) __builtin_nontemporal_store(tile[tx][k] * s, out + c * ldout + r);
			else out[c * ldout + r] = tile[tx][k] * s;
		}
	}
}

// ... 16 bytes per lane on both sides where every extent and leading dimension is a multiple of 4 and the bases are aligned
// (the volumes of the pipeline: H*W and ds are; D mostly is): a quarter of the memory instructions
typedef float post_f4 __attribute__((ext_vector_type(4)));
// A block walks ALL tiles along the short dimension (LOOP_R: the rows, else the columns) of its 64 columns / rows: the 64-float pieces it writes to
// (or reads from) one pixel's run of disparities then follow each other within microseconds, so the L2 sees whole 128-byte lines instead of thirds
// of them from blocks that run at different times (a 228-float run is 7.1 lines; measured at 370x1226x228: 1.25x the bytes written)
template <bool NT, bool LOOP_R>
__global__ void __launch_bounds__(256) transpose4_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t R,
                                                         int64_t Cn, int64_t ldin, int64_t ldout, float s)
{
	__shared__ float tile[64][65];
	const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
	const int64_t fixed0 = (int64_t)blockIdx.x * 64;
	const int64_t nloop = LOOP_R ? R : Cn;
	for (int64_t t0 = 0; t0 < nloop; t0 += 64) {
		const int64_t c0 = LOOP_R ? fixed0 : t0, r0 = LOOP_R ? t0 : fixed0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int row = ly + 16 * k;
			const int64_t r = r0 + row, c = c0 + 4 * lx;
			if (r < R && c < Cn) {
				const post_f4 *p = (const post_f4 *)(in + r * ldin + c);
				const post_f4 v = NT ? __builtin_nontemporal_load(p) : *p;
				tile[row][4 * lx + 0] = v.x; tile[row][4 * lx + 1] = v.y; tile[row][4 * lx + 2] = v.z; tile[row][4 * lx + 3] = v.w;
			}
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int col = ly + 16 * k;
			const int64_t c = c0 + col, r = r0 + 4 * lx;
			if (c < Cn && r < R) {
				const post_f4 v = {tile[4 * lx + 0][col] * s, tile[4 * lx + 1][col] * s, tile[4 * lx + 2][col] * s, tile[4 * lx + 3][col] * s};
				post_f4 *q = (post_f4 *)(out + c * ldout + r);
				if (NT) __builtin_nontemporal_store(v, q);
				else *q = v;
			}
		}
		__syncthreads();
	}
}


int fill_nan(float *p, int64_t n, hipStream_t st)
{
	if (n <= 0) return 0;
	const unsigned blocks = cdiv(n, 256) < 4096u ? cdiv(n, 256) : 4096u;
	hipLaunchKernelGGL(fill_nan_kernel, dim3(blocks), dim3(256), 0, st, p, n);
	return check_launch("fill_nan");
}

int scale(const float *in, float *out, int64_t n, float s, hipStream_t st)
{
	if (n <= 0) return 0;
	if (n % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0 && n / 4 / 256 < 0x7fffffff) {
		hipLaunchKernelGGL(scale4_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, (const f4v_t *)in, (f4v_t *)out, n / 4, s);
		return check_launch("scale");
	}
	const unsigned blocks = cdiv(n, 256) < 4096u ? cdiv(n, 256) : 4096u;
	hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, st, in, out, n, s);
	return check_launch("scale");
}

int transpose(const float *in, float *out, int64_t R, int64_t Cn, int64_t ldin, int64_t ldout, float s, hipStream_t st, int nt)
{
	// the long axis goes to grid.x (grid.y is limited to 65535 blocks)
	const bool use_nt = nt >= 0 ? nt != 0 : R * Cn * 4 > ((int64_t)768 << 20);
	const bool aligned16 = Cn % 4 == 0 && ldin % 4 == 0 && ldout % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0;
	if (R % 4 == 0 && aligned16) {
		if (R <= Cn) {
			if (use_nt) hipLaunchKernelGGL((transpose4_kernel<true, true>), dim3(cdiv(Cn, 64)), dim3(256), 0, st, in, out, R, Cn, ldin, ldout, s);
			else hipLaunchKernelGGL((transpose4_kernel<false, true>), dim3(cdiv(Cn, 64)), dim3(256), 0, st, in, out, R, Cn, ldin, ldout, s);
		} else {
			if (use_nt) hipLaunchKernelGGL((transpose4_kernel<true, false>), dim3(cdiv(R, 64)), dim3(256), 0, st, in, out, R, Cn, ldin, ldout, s);
			else hipLaunchKernelGGL((transpose4_kernel<false, false>), dim3(cdiv(R, 64)), dim3(256), 0, st, in, out, R, Cn, ldin, ldout, s);
		}
		return check_launch("transpose");
	}
	if (use_nt) hipLaunchKernelGGL(transpose_kernel<true>, dim3(cdiv(Cn, 64), cdiv(R, 64)), dim3(256), 0, st, in, out, R, Cn, ldin, ldout, s);
	else hipLaunchKernelGGL(transpose_kernel<false>, dim3(cdiv(Cn, 64), cdiv(R, 64)), dim3(256), 0, st, in, out, R, Cn, ldin, ldout, s);
	return check_launch("transpose");
}

// ---- fix_border (main.lua:922-927) ---------------------------------------------------
// (D,H,W): vol[.., dst] = vol[.., src] for the n outermost columns on one side.
__global__ void __launch_bounds__(256) fix_border_kernel(float *__restrict__ vol, int64_t rows, int W, int n, int direction)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= rows * n) return;
	const int i = (int)(id % n) + 1;
	const int64_t r = id / n;
	const int dst = direction < 0 ? W - i : i - 1;
	const int src = direction < 0 ? W - (n + 1) : n;
	vol[r * W + dst] = vol[r * W + src];
}

int fix_border(float *vol, int D, int H, int W, int n, int direction, hipStream_t st)
{
	if (n <= 0) return 0;
	const int64_t rows = (int64_t)D * H;
	hipLaunchKernelGGL(fix_border_kernel, dim3(cdiv(rows * n, 256)), dim3(256), 0, st, vol, rows, W, n, direction);
	return check_launch("fix_border");
}

// ---- argmin -------------------------------------------------------------------------------
// torch.min(vol,2)-1 (main.lua:1049-1050) with the spatial_argmin convention (adcensus.cu:244-262).
// (D,H,W): one thread per pixel, coalesced along x.  base1 = 1 gives spatial_argmin's 1-based output.
__global__ void __launch_bounds__(256) argmin_dhw_kernel(const float *__restrict__ vol, float *__restrict__ out, int D, int64_t HW,
                                                         int base1)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= HW) return;
	int argmin = 0;
	float mn = __builtin_inff();
	for (int i = 0; i < D; ++i) {
		const float val = __builtin_nontemporal_load(vol + i * HW + p);
		if (val < mn) {
			mn = val;
			argmin = i;
		}
	}
	out[p] = (float)(argmin + base1);
}

// ... four pixels per thread (16-byte loads) where the plane stride and the bases allow
__global__ void __launch_bounds__(256) argmin_dhw4_kernel(const float *__restrict__ vol, float *__restrict__ out, int D, int64_t HW,
                                                          int base1)
{
	const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (p >= HW) return;
	int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	float m0 = __builtin_inff(), m1 = m0, m2 = m0, m3 = m0;
	for (int i = 0; i < D; ++i) {
		const post_f4 v = __builtin_nontemporal_load((const post_f4 *)(vol + i * HW + p));
		if (v.x < m0) { m0 = v.x; a0 = i; }
		if (v.y < m1) { m1 = v.y; a1 = i; }
		if (v.z < m2) { m2 = v.z; a2 = i; }
		if (v.w < m3) { m3 = v.w; a3 = i; }
	}
	*(post_f4 *)(out + p) = post_f4{(float)(a0 + base1), (float)(a1 + base1), (float)(a2 + base1), (float)(a3 + base1)};
}

// (H,W,ds) layout: one wave per pixel, lanes over d.
__global__ void __launch_bounds__(256) argmin_hwd_kernel(const float *__restrict__ vol, float *__restrict__ out, int D, int ds,
                                                         int64_t HW)
{
	const int lane = threadIdx.x & 63;
	const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (p >= HW) return;
	const float INF = __builtin_inff();
	float best = INF;
	int bi = 0;
	for (int d = lane; d < D; d += 64) {  // ascending d per lane keeps "first" semantics
		const float val = vol[p * ds + d];
		if (val < best) {
			best = val;
			bi = d;
		}
	}
	const float mall = wave_min(best);
	// smallest index among lanes holding the minimum
	int cand = (best == mall && best < INF) ? bi : 0x7fffffff;
	for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
	if (lane == 0) out[p] = (float)(cand == 0x7fffffff ? 0 : cand);
}

int argmin_dhw(const float *vol, float *out, int D, int H, int W, int base1, hipStream_t st)
{
	const int64_t HW = (int64_t)H * W;
	if (HW % 4 == 0 && (uintptr_t)vol % 16 == 0 && (uintptr_t)out % 16 == 0)
		hipLaunchKernelGGL(argmin_dhw4_kernel, dim3(cdiv(HW / 4, 256)), dim3(256), 0, st, vol, out, D, HW, base1);
	else
		hipLaunchKernelGGL(argmin_dhw_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, vol, out, D, HW, base1);
	return check_launch("argmin_dhw");
}

int argmin_hwd(const float *vol, float *out, int D, int ds, int H, int W, hipStream_t st)
{
	const int64_t HW = (int64_t)H * W;
	hipLaunchKernelGGL(argmin_hwd_kernel, dim3(cdiv(HW * 64, 256)), dim3(256), 0, st, vol, out, D, ds, HW);
	return check_launch("argmin_hwd");
}

// ---- outlier_detection, adcensus.cu:878-899 ------------------------------------------------
// The reference's inner loop asks, per left pixel x, whether some d in [0, disp_max) has |d - d1[x-d]| < 1.1.  Seen from
// the right map that is a SCATTER: pixel x' of d1 can only match the few integers d around d1[x'], i.e. the left pixels
// x = x' + d.  One block per image row marks those pixels in LDS (every candidate d is tested with the reference's own
// float expression, so the set of marked pixels is exactly the set the loop would find), then classifies the row:
// O(1) per pixel instead of O(disp_max).
__global__ void __launch_bounds__(256) outlier_rows_kernel(const float *__restrict__ d0, const float *__restrict__ d1,
                                                           float *__restrict__ outlier, int W, int disp_max)
{
	extern __shared__ unsigned char mark[];
	const int y = blockIdx.x;
	const int64_t row = (int64_t)y * W;
	for (int x = threadIdx.x; x < W; x += 256) mark[x] = 0;
	__syncthreads();
	for (int xp = threadIdx.x; xp < W; xp += 256) {
		const float v = d1[row + xp];
		// |d - v| < 1.1 with 0 <= d < disp_max needs -1.1 < v < disp_max + 0.1 (NaN fails both tests)
		if (v > -2.0f && v < (float)disp_max + 2.0f) {
			const int base = (int)floorf(v);
#pragma unroll
			for (int k = -2; k <= 3; ++k) {
				const int d = base + k;
				if (d >= 0 && d < disp_max && xp + d < W && (double)fabsf((float)d - v) < 1.1) mark[xp + d] = 1;
			}
		}
	}
	__syncthreads();
	for (int x = threadIdx.x; x < W; x += 256) {
		const int64_t id = row + x;
		const int d0i = (int)d0[id];
		float res;
		if (x - d0i < 0) res = 1;
		else if ((double)fabsf(d0[id] - d1[id - d0i]) < 1.1) res = 0;
		else res = mark[x] ? 2.0f : 1.0f;
		outlier[id] = res;
	}
}

int outlier_detection(const float *d0, const float *d1, float *outlier, int H, int W, int disp_max, hipStream_t st)
{
	hipLaunchKernelGGL(outlier_rows_kernel, dim3(H), dim3(256), (size_t)W, st, d0, d1, outlier, W, disp_max);
	return check_launch("outlier_detection");
}

// ---- interpolate_occlusion, adcensus.cu:1079-1105 --------------------------------------------
// An occluded pixel takes the disparity of the nearest pixel to its LEFT that passed the left-right check; if the row
// has none to its left, the reference's second scan finds the first one to the right, which is then the first valid
// pixel of the row.  Both are row scans: one block per row, each thread owns a short run of pixels, the "last valid
// index so far" is carried across threads by a max-scan in LDS.
__global__ void __launch_bounds__(256) interp_occ_rows_kernel(const float *__restrict__ d0, const float *__restrict__ outlier,
                                                              float *__restrict__ out, int W)
{
	__shared__ int carry[256];
	__shared__ int first_valid;
	const int y = blockIdx.x, t = threadIdx.x;
	const int64_t row = (int64_t)y * W;
	const int seg = (W + 255) / 256;
	const int xa = t * seg, xb = min(W, xa + seg);
	if (t == 0) first_valid = W;
	int last = -1;
	for (int x = xa; x < xb; ++x)
		if (outlier[row + x] == 0) last = x;
	carry[t] = last;
	__syncthreads();
	if (last >= 0) {
		int f = -1;
		for (int x = xa; x < xb; ++x)
			if (outlier[row + x] == 0) { f = x; break; }
		atomicMin(&first_valid, f);
	}
	// inclusive max-scan of carry[] (Hillis-Steele)
	for (int off = 1; off < 256; off <<= 1) {
		const int o = t >= off ? carry[t - off] : -1;
		__syncthreads();
		carry[t] = max(carry[t], o);
		__syncthreads();
	}
	int lv = t > 0 ? carry[t - 1] : -1;   // nearest valid pixel left of this thread's run
	const int fv = first_valid;
	for (int x = xa; x < xb; ++x) {
		const int64_t id = row + x;
		const float o = outlier[id];
		if (o == 0) lv = x;
		float r = d0[id];
		if (o == 1) {
			if (lv >= 0) r = d0[row + lv];
			else if (fv < W) r = d0[row + fv];
		}
		out[id] = r;
	}
}

int interpolate_occlusion(const float *d0, const float *outlier, float *out, int H, int W, hipStream_t st)
{
	hipLaunchKernelGGL(interp_occ_rows_kernel, dim3(H), dim3(256), 0, st, d0, outlier, out, W);
	return check_launch("interpolate_occlusion");
}

// ---- interpolate_mismatch, adcensus.cu:1001-1058 ----------------------------------------------
// 16 rays; coordinates accumulate in float and are rounded half away from zero (CUDA round()).
// If every ray leaves the image the reference reads an uninitialised value (assert compiled
// out); defined here as "keep d0".
// One LANE per ray: the 16 lanes of a DPP row walk the 16 rays of one pixel concurrently (a wave = 4 pixels), then
// every lane ranks its value against the other 15 with row rotations; the lane whose value has rank n/2 in the
// ascending order of the n rays that ended inside the image writes it (sort(), adcensus.cu:47-60: equal values are
// interchangeable, so rank selection returns the same number).
template <int K> __device__ __forceinline__ void ray_rank_step(float v, int inb, int &less, int &eq)
{
	const float ov = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x120 + K, 0xf, 0xf, false));
	const int oin = __builtin_amdgcn_update_dpp(0, inb, 0x120 + K, 0xf, 0xf, false);
	less += (oin && ov < v) ? 1 : 0;
	eq += (oin && ov == v) ? 1 : 0;
}

// The walk in integers: the reference accumulates xx += dx in float with dx in {0, +-0.5, +-1} from an integer start, so every
// xx is an exact multiple of 0.5 and round(xx) (half away from zero) = (X2 + 1 + (X2 >> 31)) >> 1 for X2 = 2 xx -- the same pixels,
// without the float rounding sequence.  The mark is fetched through a buffer whose range check answers 0 ("not a mismatch")
// for a position outside the image, so the loop has one exit test and no nested regions.
#define MC_MIS_NU 4
__global__ void __launch_bounds__(256) interp_mis_rays_kernel(const float *__restrict__ d0, const float *__restrict__ outlier,
                                                              float *__restrict__ out, int size, int H, int W)
{
	// direction (dx, dy) of ray k in half pixels, adcensus.cu:1013-1030:
	// dx = {0, -.5, -1, -1, -1, -1, -1, -.5, 0, .5, 1, 1, 1, 1, 1, .5}, dy = {1, 1, 1, .5, 0, -.5, -1, -1, -1, -1, -1, -.5, 0, .5, 1, 1}
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	const int id = t >> 4;
	const int ray = (int)(threadIdx.x & 15);
	if (id >= size) return;
	const bool mis = outlier[id] == 2;
	if (!mis) {
		if (ray == 0) out[id] = d0[id];
		return;
	}
	const int hx[16] = {0, -1, -2, -2, -2, -2, -2, -1, 0, 1, 2, 2, 2, 2, 2, 1};
	const int hy[16] = {2, 2, 2, 1, 0, -1, -2, -2, -2, -2, -2, -1, 0, 1, 2, 2};
	const int x = id % W, y = id / W;
	const int sx = hx[ray], sy = hy[ray];
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)outlier, 0, size * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)d0, 0, size * 4, 0x00020000);
	int X2 = 2 * x, Y2 = 2 * y;
	// The path does not depend on what it finds: the marks of the next MC_MIS_NU positions are requested together and tested in
	// order, so a step costs a fraction of a memory round trip (the start pixel is a mismatch: the walk begins one step out).
	constexpr unsigned OOB = 0x80000000u;
	unsigned stop = OOB;             // byte offset of the pixel the ray stops at, OOB if it left the image
	for (bool go = true; go;) {
		unsigned off[MC_MIS_NU], o[MC_MIS_NU];
#pragma unroll
		for (int i = 0; i < MC_MIS_NU; ++i) {
			X2 += sx;
			Y2 += sy;
			const int xi = (X2 + 1 + (X2 >> 31)) >> 1;
			const int yi = (Y2 + 1 + (Y2 >> 31)) >> 1;
			const bool in = (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W;
			off[i] = in ? (__umul24((unsigned)yi, (unsigned)W) + (unsigned)xi) * 4u : OOB;   // H*W < 2^27
			o[i] = __builtin_amdgcn_raw_buffer_load_b32(ro, off[i], 0, 0);
		}
#pragma unroll
		for (int i = MC_MIS_NU - 1; i >= 0; --i) {       // the first position whose mark is not 2 wins
			const bool hit = __uint_as_float(o[i]) != 2.0f;
			stop = hit ? off[i] : stop;
			go = hit ? false : go;
		}
	}
	const int inb = stop != OOB ? 1 : 0;
	const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, stop, 0, 0));   // 0 outside
	// all 16 lanes of this pixel are here (mis is uniform over the row of 16)
	int less = 0, eq = inb;
	ray_rank_step<1>(v, inb, less, eq); ray_rank_step<2>(v, inb, less, eq); ray_rank_step<3>(v, inb, less, eq);
	ray_rank_step<4>(v, inb, less, eq); ray_rank_step<5>(v, inb, less, eq); ray_rank_step<6>(v, inb, less, eq);
	ray_rank_step<7>(v, inb, less, eq); ray_rank_step<8>(v, inb, less, eq); ray_rank_step<9>(v, inb, less, eq);
	ray_rank_step<10>(v, inb, less, eq); ray_rank_step<11>(v, inb, less, eq); ray_rank_step<12>(v, inb, less, eq);
	ray_rank_step<13>(v, inb, less, eq); ray_rank_step<14>(v, inb, less, eq); ray_rank_step<15>(v, inb, less, eq);
	const uint64_t bal = __ballot(inb != 0);
	const int n = __builtin_popcount((unsigned)((bal >> ((threadIdx.x & 48))) & 0xffffu));
	if (n == 0) {
		if (ray == 0) out[id] = d0[id];
		return;
	}
	const int want = n / 2;
	if (inb && less <= want && want < less + eq) out[id] = v;   // lanes that qualify hold the same value
}

int interpolate_mismatch(const float *d0, const float *outlier, float *out, int H, int W, hipStream_t st)
{
	const int64_t size = (int64_t)H * W;
	MC_REQUIRE(size * 16 < ((int64_t)1 << 31), "interpolate_mismatch: %dx%d pixels x 16 rays do not fit a 32-bit index", H, W);
	MC_REQUIRE(H < (1 << 24) && W < (1 << 24), "interpolate_mismatch: %dx%d: a side beyond the 24-bit row / column arithmetic of the ray walk", H, W);
	hipLaunchKernelGGL(interp_mis_rays_kernel, dim3(cdiv(size * 16, 256)), dim3(256), 0, st, d0, outlier, out, (int)size, H, W);
	return check_launch("interpolate_mismatch");
}

// ---- subpixel_enchancement, adcensus.cu:1205-1220 ----------------------------------------------
// cost of (pixel p, disparity d) is vol[d*sd + p*sp]: (D,H,W): sd=HW, sp=1 ; (H,W,ds): sd=1, sp=ds.
__global__ void __launch_bounds__(256) subpixel_kernel(const float *__restrict__ d0, const float *__restrict__ c2,
                                                       float *__restrict__ out, int64_t size, int64_t sd, int64_t sp, int disp_max)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= size) return;
	const int d = (int)d0[id];
	float res = (float)d;
	if (1 <= d && d < disp_max - 1) {
		const float cn = c2[(d - 1) * sd + id * sp];
		const float cz = c2[d * sd + id * sp];
		const float cp = c2[(d + 1) * sd + id * sp];
		const float denom = 2 * (cp + cn - 2 * cz);
		if ((double)denom > 1e-5) {
			res = (float)((double)d - fmin(1.0, fmax(-1.0, (double)((cp - cn) / denom))));
		}
	}
	out[id] = res;
}

int subpixel(const float *d0, const float *vol, float *out, int D, int H, int W, int64_t sd, int64_t sp, hipStream_t st)
{
	const int64_t size = (int64_t)H * W;
	hipLaunchKernelGGL(subpixel_kernel, dim3(cdiv(size, 256)), dim3(256), 0, st, d0, vol, out, size, sd, sp, D);
	return check_launch("subpixel_enchancement");
}

// ---- median2d, adcensus.cu:1575-1594 ------------------------------------------------------------
// xs[n/2] of the ascending sort of the in-bounds taps (no NaNs reach this stage, so any correct selection equals
// the reference's selection sort).  Interior pixels of the 3x3 / 5x5 filters use forgetful selection in registers
// (repeatedly drop the minimum and maximum of a working set of N/2+2 values, feeding in the rest: ~130 compare-
// exchanges for N = 25); border pixels and larger kernels count ranks.
__device__ __forceinline__ void cswap(float &a, float &b)
{
	const float lo = fminf(a, b), hi = fmaxf(a, b);
	a = lo;
	b = hi;
}

template <int N>
__device__ __forceinline__ float median_forgetful(const float (&v)[N])
{
	constexpr int M0 = N / 2 + 2;
	float w[M0];
#pragma unroll
	for (int i = 0; i < M0; ++i) w[i] = v[i];
	int next = M0;
#pragma unroll
	for (int m = M0; m > 3; --m) {
		// minimum of w[0..m) to w[0], maximum to w[m-1]
#pragma unroll
		for (int i = 0; i < m / 2; ++i) cswap(w[i], w[m - 1 - i]);
#pragma unroll
		for (int i = 1; i < (m + 1) / 2; ++i) cswap(w[0], w[i]);
#pragma unroll
		for (int i = m / 2; i < m - 1; ++i) cswap(w[i], w[m - 1]);
		// forget both; the next unseen value takes the minimum's slot, the set shrinks by one from the top
		// (N - M0 == M0 - 3 for odd N: every round has a value to feed in)
		w[0] = v[next < N ? next : N - 1];
		++next;
	}
	// three values left: ranks N/2-1 .. N/2+1
	cswap(w[0], w[1]);
	cswap(w[1], w[2]);
	cswap(w[0], w[1]);
	return w[1];
}

template <int KR>
__global__ void __launch_bounds__(256) median_kernel(const float *__restrict__ img, float *__restrict__ out, int H, int W)
{
	constexpr int K = 2 * KR + 1;
	constexpr int TW = 64 + 2 * KR, TH = 4 + 2 * KR;
	__shared__ float tile[TH][TW + 1];
	const int bx = blockIdx.x * 64, by = blockIdx.y * 4;
	for (int i = threadIdx.x; i < TH * TW; i += 256) {
		const int ty = i / TW, tx = i % TW;
		const int gx = bx + tx - KR, gy = by + ty - KR;
		tile[ty][tx] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? img[(int64_t)gy * W + gx] : 0.0f;
	}
	__syncthreads();
	const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
	const int x = bx + lx, y = by + ly;
	if (x >= W || y >= H) return;
	const int xa = max(0, x - KR), xb = min(W - 1, x + KR), ya = max(0, y - KR), yb = min(H - 1, y + KR);
	const int n = (xb - xa + 1) * (yb - ya + 1);
	if constexpr (KR == 1 || KR == 2) {
		if (n == K * K) {  // interior pixel
			float v[K * K];
#pragma unroll
			for (int i = 0; i < K * K; ++i) v[i] = tile[ly + i % K][lx + i / K];
			out[(int64_t)y * W + x] = median_forgetful<K * K>(v);
			return;
		}
	}
	const int want = n / 2;
	float res = 0;
	for (int i = 0; i < K * K; ++i) {
		const int ix = x - KR + i / K, iy = y - KR + i % K;
		if (ix < xa || ix > xb || iy < ya || iy > yb) continue;
		const float vi = tile[iy - by + KR][ix - bx + KR];
		int less = 0, eq = 0;
		for (int yy = ya; yy <= yb; ++yy) {
			for (int xx = xa; xx <= xb; ++xx) {
				const float vj = tile[yy - by + KR][xx - bx + KR];
				less += vj < vi ? 1 : 0;
				eq += vj == vi ? 1 : 0;
			}
		}
		if (less <= want && want < less + eq) {
			res = vi;
			break;
		}
	}
	out[(int64_t)y * W + x] = res;
}

int median2d(const float *img, float *out, int H, int W, int k, hipStream_t st)
{
	const dim3 grid(cdiv(W, 64), cdiv(H, 4)), block(256);
	switch (k / 2) {
	case 0: hipLaunchKernelGGL(median_kernel<0>, grid, block, 0, st, img, out, H, W); break;
	case 1: hipLaunchKernelGGL(median_kernel<1>, grid, block, 0, st, img, out, H, W); break;
	case 2: hipLaunchKernelGGL(median_kernel<2>, grid, block, 0, st, img, out, H, W); break;
	case 3: hipLaunchKernelGGL(median_kernel<3>, grid, block, 0, st, img, out, H, W); break;
	case 4: hipLaunchKernelGGL(median_kernel<4>, grid, block, 0, st, img, out, H, W); break;
	default: hipLaunchKernelGGL(median_kernel<5>, grid, block, 0, st, img, out, H, W); break;
	}
	return check_launch("median2d");
}

// ---- mean2d, adcensus.cu:1241-1261 ----------------------------------------------------------------
// Range-gated Gaussian mean.  Tap order (xx outer, yy inner, running kernel index) and the FMA-contracted
// `sum += img*k` are the reference's, so the result is bit-identical.
// A 256-thread block produces a 64 x 8 output tile; a thread owns 2 vertically adjacent outputs, so one LDS read of
// a tap column feeds both (their windows overlap in all but one row).  (4 outputs per thread and 2 waves per block left the
// chip with 1.8 waves per SIMD at KITTI size -- a lone wave issues an instruction every ~9 cycles; 2 per thread: 0.49 -> 0.44 ms
// for the whole post-processing stage, 1 per thread the same.)  Image tile (+halo) and the (ks,ks)
// weights live in LDS.  A tap that does not count adds w' = -0.0 instead of branching: x + (-0.0) == x and
// fma(v, -0.0, s) == s for every finite v and every s this loop can hold (s starts at +0), so gated-out and
// out-of-image taps (staged as a huge finite value) are exact no-ops.
constexpr int M2_OY = 2;   // outputs per thread
constexpr int M2_TR = 4;   // thread rows per block
__global__ void __launch_bounds__(64 * M2_TR) mean2d_kernel(const float *__restrict__ img, const float *__restrict__ kernel,
                                                            float *__restrict__ out, int H, int W, int kr, float alpha2)
{
	extern __shared__ __attribute__((aligned(16))) float smem[];
	constexpr int NT = 64 * M2_TR;
	const int ks = 2 * kr + 1;
	const int TW = 64 + 2 * kr, TH = M2_TR * M2_OY + 2 * kr;
	const int TS = TW + 1;
	const int WS = ks + 2 * (M2_OY - 1);       // padded weight column: [ -3 .. ks+2 ]
	float *tile = smem;                          // [TH][TS]
	float *wts = smem + TH * TS;                 // [ks][WS], wts[ix*WS + (M2_OY-1) + iy]
	const int bx = blockIdx.x * 64, by = blockIdx.y * (M2_TR * M2_OY);
	const float FAR = 1e30f;                     // |FAR - c| < alpha2 is false for any disparity c
	for (int i = threadIdx.x; i < TH * TW; i += NT) {
		const int ty = i / TW, tx = i - ty * TW;
		const int gx = bx + tx - kr, gy = by + ty - kr;
		tile[ty * TS + tx] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? img[(int64_t)gy * W + gx] : FAR;
	}
	for (int i = threadIdx.x; i < ks * WS; i += NT) {
		const int ix = i / WS, j = i - ix * WS - (M2_OY - 1);
		wts[i] = (j >= 0 && j < ks) ? kernel[ix * ks + j] : -0.0f;
	}
	__syncthreads();
	const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
	const int x = bx + lx, yb = by + ly * M2_OY;
	if (x >= W || yb >= H) return;
	float c[M2_OY], sum[M2_OY], cnt[M2_OY];
#pragma unroll
	for (int o = 0; o < M2_OY; ++o) {
		c[o] = tile[(ly * M2_OY + o + kr) * TS + lx + kr];
		sum[o] = 0.0f;
		cnt[o] = 0.0f;
	}
	const int nj = ks + M2_OY - 1;  // tile rows a thread's 4 windows span
	for (int ix = 0; ix < ks; ++ix) {
		const float *col = tile + (ly * M2_OY) * TS + lx + ix;
		const float *wc = wts + ix * WS + (M2_OY - 1);
		float wprev[M2_OY > 1 ? M2_OY - 1 : 1];    // weights of taps j-1, j-2, ...
#pragma unroll
		for (int o = 0; o < M2_OY - 1; ++o) wprev[o] = -0.0f;
#pragma unroll 4
		for (int j = 0; j < nj; ++j) {
			const float v = col[0];
			col += TS;
			const float w0 = wc[j];                  // -0.0 beyond the kernel
			float ws[M2_OY];                         // output o sees this row as its tap j-o
			ws[0] = w0;
#pragma unroll
			for (int o = 1; o < M2_OY; ++o) ws[o] = wprev[o - 1];
#pragma unroll
			for (int o = 0; o < M2_OY; ++o) {
				const float w = fabsf(v - c[o]) < alpha2 ? ws[o] : -0.0f;
				sum[o] = fmaf(v, w, sum[o]);
				cnt[o] += w;
			}
#pragma unroll
			for (int o = M2_OY - 2; o > 0; --o) wprev[o] = wprev[o - 1];
			if (M2_OY > 1) wprev[0] = w0;
		}
	}
#pragma unroll
	for (int o = 0; o < M2_OY; ++o)
		if (yb + o < H) out[(int64_t)(yb + o) * W + x] = sum[o] / cnt[o];
}

// The table radii (kr = 6, 9, 14, 18, 24: every blur_sigma of main.lua:68-295) as compile-time sizes: one output per thread,
// 64 x 8 outputs per block of 8 waves (twice the waves of the kernel above: at KITTI size that one leaves 2.3 waves per
// SIMD on average and a VALU instruction every 5.6 cycles -- r03_kitti_fast_pmc.csv).  A tap is four VALU instructions: the gate
// narrows EXEC (v_cmpx) and the two accumulations run under it with the weight as a scalar operand, so a tap that does
// not count is skipped instead of adding -0.0, and the weights come through the scalar cache (one s_load_dwordx8 per 8 taps)
// instead of LDS.  Same taps in the same order, same FMA: bit-identical to the kernel above.
#define MC_M2_TAP(i) \
	"v_sub_f32 %[d], %[v" #i "], %[c]\n\tv_cmpx_lt_f32_e64 vcc, |%[d]|, %[a]\n\tv_fmac_f32 %[s], %[w" #i "], %[v" #i "]\n\t" \
	"v_add_f32 %[n], %[w" #i "], %[n]\n\ts_mov_b64 exec, %[e]\n\t"
__device__ __forceinline__ void mean2d_taps8(float &sum, float &cnt, const float (&v)[8], const float (&w)[8], float c, float alpha2,
                                             unsigned long long ex)
{
	float d;
	asm volatile(MC_M2_TAP(0) MC_M2_TAP(1) MC_M2_TAP(2) MC_M2_TAP(3) MC_M2_TAP(4) MC_M2_TAP(5) MC_M2_TAP(6) MC_M2_TAP(7)
	             : [s] "+v"(sum), [n] "+v"(cnt), [d] "=&v"(d)
	             : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]),
	               [v7] "v"(v[7]), [w0] "s"(w[0]), [w1] "s"(w[1]), [w2] "s"(w[2]), [w3] "s"(w[3]), [w4] "s"(w[4]), [w5] "s"(w[5]),
	               [w6] "s"(w[6]), [w7] "s"(w[7]), [c] "v"(c), [a] "s"(alpha2), [e] "s"(ex)
	             : "vcc");
}
__device__ __forceinline__ void mean2d_tap1(float &sum, float &cnt, float v0, float w0, float c, float alpha2, unsigned long long ex)
{
	float d;
	asm volatile(MC_M2_TAP(0) : [s] "+v"(sum), [n] "+v"(cnt), [d] "=&v"(d)
	             : [v0] "v"(v0), [w0] "s"(w0), [c] "v"(c), [a] "s"(alpha2), [e] "s"(ex)
	             : "vcc");
}
#undef MC_M2_TAP

template <int KR>
__global__ void __launch_bounds__(512) mean2d_taps_kernel(const float *__restrict__ img, const float *__restrict__ kernel,
                                                          float *__restrict__ out, int H, int W, float alpha2)
{
	constexpr int KS = 2 * KR + 1, TW = 64 + 2 * KR, TH = 8 + 2 * KR, TS = TW + 1;
	__shared__ float tile[TH * TS];
	const int bx = blockIdx.x * 64, by = blockIdx.y * 8;
	const float FAR = 1e30f;                     // |FAR - c| < alpha2 is false for any disparity c
	for (int i = threadIdx.x; i < TH * TW; i += 512) {
		const int ty = i / TW, tx = i - ty * TW;
		const int gx = bx + tx - KR, gy = by + ty - KR;
		tile[ty * TS + tx] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? img[(int64_t)gy * W + gx] : FAR;
	}
	__syncthreads();
	const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
	const int x = bx + lx, y = by + ly;
	if (x >= W || y >= H) return;
	const unsigned long long ex = __builtin_amdgcn_read_exec();   // the lanes that own an output: restored after every tap
	const float c = tile[(ly + KR) * TS + lx + KR];
	float sum = 0.0f, cnt = 0.0f;
	const float *col = tile + ly * TS + lx;
	for (int ix = 0; ix < KS; ++ix, ++col) {
		const float *wc = kernel + ix * KS;      // uniform: scalar loads
#pragma unroll
		for (int j0 = 0; j0 + 8 <= KS; j0 += 8) {
			float v[8], w[8];
#pragma unroll
			for (int g = 0; g < 8; ++g) {
				v[g] = col[(j0 + g) * TS];
				w[g] = wc[j0 + g];
			}
			__builtin_amdgcn_sched_barrier(0);
			mean2d_taps8(sum, cnt, v, w, c, alpha2, ex);
			__builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for (int j = KS / 8 * 8; j < KS; ++j) mean2d_tap1(sum, cnt, col[j * TS], wc[j], c, alpha2, ex);
	}
	out[(int64_t)y * W + x] = sum / cnt;
}

template <int KR>
static int mean2d_taps(const float *img, const float *kernel, float *out, int H, int W, float alpha2, hipStream_t st)
{
	hipLaunchKernelGGL(mean2d_taps_kernel<KR>, dim3(cdiv(W, 64), cdiv(H, 8)), dim3(512), 0, st, img, kernel, out, H, W, alpha2);
	return check_launch("mean2d");
}

int mean2d(const float *img, const float *kernel, float *out, int H, int W, int ks, float alpha2, hipStream_t st)
{
	switch (ks) {
	case 13: return mean2d_taps<6>(img, kernel, out, H, W, alpha2, st);
	case 19: return mean2d_taps<9>(img, kernel, out, H, W, alpha2, st);
	case 29: return mean2d_taps<14>(img, kernel, out, H, W, alpha2, st);
	case 37: return mean2d_taps<18>(img, kernel, out, H, W, alpha2, st);
	case 49: return mean2d_taps<24>(img, kernel, out, H, W, alpha2, st);
	default: break;   // any other size: the kernel with run-time sizes
	}
	const int kr = ks / 2;
	const size_t lds =
	    ((size_t)(M2_TR * M2_OY + 2 * kr) * (64 + 2 * kr + 1) + (size_t)ks * (ks + 2 * (M2_OY - 1))) * sizeof(float);
	if (lds > 160 * 1024) {
		set_error("mean2d: kernel size %d needs %zu B of LDS (> 160 KiB)", ks, lds);
		return MC_EINVAL;
	}
	if (lds > 64 * 1024) {
		hipError_t e = hipFuncSetAttribute((const void *)mean2d_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			set_error("mean2d: hipFuncSetAttribute: %s", hipGetErrorString(e));
			return (int)e;
		}
	}
	hipLaunchKernelGGL(mean2d_kernel, dim3(cdiv(W, 64), cdiv(H, M2_TR * M2_OY)), dim3(64 * M2_TR), lds, st, img, kernel, out, H, W, kr,
	                   alpha2);
	return check_launch("mean2d");
}

// ---- Normalize_forward, adcensus.cu:1284-1308 -------------------------------------------------------
// One thread per pixel: norm = sum_c x^2 (c ascending, fma as nvcc contracts sum += x * x) + 1e-5, out = x / sqrtf(norm).  CT > 0: the pixel's C <= CT
// values stay in registers between the two loops (every byte read once: 0.17 -> 0.11 ms at 2 x 64 x 370 x 1226); CT = 0: any C, values read twice.
template <int CT>
__global__ void __launch_bounds__(256) normalize_kernel(const float *__restrict__ in, float *__restrict__ norm, float *__restrict__ out,
                                                        int C, int64_t HW, int64_t NHW)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= NHW) return;
	const int64_t n = id / HW, p = id % HW;
	const float *src = in + n * C * HW + p;
	float *dst = out + n * C * HW + p;
	float sum = 0.0f;
	if constexpr (CT > 0) {
		float v[CT];
#pragma unroll
		for (int c = 0; c < CT; ++c) v[c] = c < C ? __builtin_nontemporal_load(src + c * HW) : 0.0f;
#pragma unroll
		for (int c = 0; c < CT; ++c)
			if (c < C) sum = fmaf(v[c], v[c], sum);
		const float nv = (float)((double)sum + 1e-5);
		if (norm) norm[id] = nv;
		const float r = sqrtf(nv);
#pragma unroll
		for (int c = 0; c < CT; ++c)
			if (c < C) dst[c * HW] = v[c] / r;
	} else {
		for (int c = 0; c < C; ++c) {
			const float x = src[c * HW];
			sum = fmaf(x, x, sum);
		}
		const float nv = (float)((double)sum + 1e-5);
		if (norm) norm[id] = nv;
		const float r = sqrtf(nv);
		for (int c = 0; c < C; ++c) dst[c * HW] = src[c * HW] / r;
	}
}

int normalize_forward(const float *in, float *norm, float *out, int N, int C, int H, int W, hipStream_t st)
{
	const int64_t HW = (int64_t)H * W;
	const dim3 grid(cdiv(N * HW, 256)), block(256);
	if (C <= 64) hipLaunchKernelGGL(normalize_kernel<64>, grid, block, 0, st, in, norm, out, C, HW, N * HW);
	else if (C <= 128) hipLaunchKernelGGL(normalize_kernel<128>, grid, block, 0, st, in, norm, out, C, HW, N * HW);
	else hipLaunchKernelGGL(normalize_kernel<0>, grid, block, 0, st, in, norm, out, C, HW, N * HW);
	return check_launch("normalize_forward");
}

}  // namespace mc
