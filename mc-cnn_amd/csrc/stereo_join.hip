// Matching-cost volumes from features (StereoJoin) and the ad / census baselines.
//
// StereoJoin_ (adcensus.cu:1455-1477): s(x,d) = -sum_c L[c,y,x] * R[c,y,x-d], accumulated
// with c ascending as sum = fma(-L, R, sum) (nvcc contracts `sum -= a*b`); the value goes to
// volL[d,y,x] and volR[d,y,x-d].  The fma chain order is kept, so results are bit-identical.
#include "mc_common.h"
#include <type_traits>

namespace mc {

// ---- (D,H,W) outputs, the reference's layout (mc_stereo_join: what an unchanged main.lua calls) ---------------------
// A block owns 64 pixels x 32 disparities of one image row: the two feature rows it needs -- L[c][x0 .. x0+63] and
// R[c][x0-d0-31 .. x0-d0+63] -- are staged through LDS 16 channels at a time, a thread keeps the sums of 8 consecutive
// disparities of one pixel (one value of L per channel against 8 of R), and both stores run along x.  Only x-d >= 0 voxels
// are written (the reference leaves the rest to the caller's NaN fill, main.lua:946).  The chain is the reference's:
// sum = fma(-L, R, sum), c ascending.  (Round 6: the one-thread-per-voxel form this replaces re-read every feature value
// 32 times from L2 and took 7.2 ms of the op-by-op route's 10.3 ms per KITTI pair, profiles/r06_ops_route.txt.)
constexpr int JD_PX = 64, JD_D = 32, JD_CK = 16, JD_RW = JD_PX + JD_D;   // 96 staged R columns (95 used)

__global__ void __launch_bounds__(256) join_dhw_kernel(const float *__restrict__ fL, const float *__restrict__ fR,
                                                       float *__restrict__ volL, float *__restrict__ volR, int C, int D, int H, int W)
{
	__shared__ float Ls[JD_CK][JD_PX];
	__shared__ float Rs[JD_CK][JD_RW];
	const int tid = threadIdx.x, xi = tid & 63, dg = tid >> 6;
	const int x0 = blockIdx.x * JD_PX, y = blockIdx.y, d0 = blockIdx.z * JD_D;
	const int64_t HW = (int64_t)H * W;
	const float *__restrict__ rowL = fL + (int64_t)y * W;
	const float *__restrict__ rowR = fR + (int64_t)y * W;
	const int r0 = x0 - d0 - (JD_D - 1);             // image column of Rs[.][0]
	float sum[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) sum[k] = 0.0f;
	for (int c0 = 0; c0 < C; c0 += JD_CK) {
		__syncthreads();
		for (int e = tid; e < JD_CK * JD_PX; e += 256) {
			const int c = e / JD_PX, i = e % JD_PX;
			Ls[c][i] = (c0 + c < C && x0 + i < W) ? rowL[(int64_t)(c0 + c) * HW + x0 + i] : 0.0f;
		}
		for (int e = tid; e < JD_CK * JD_RW; e += 256) {
			const int c = e / JD_RW, j = e % JD_RW;
			const int xr = r0 + j;
			Rs[c][j] = (c0 + c < C && xr >= 0 && xr < W) ? rowR[(int64_t)(c0 + c) * HW + xr] : 0.0f;
		}
		__syncthreads();
		const int nc = min(JD_CK, C - c0);
		for (int c = 0; c < nc; ++c) {
			const float l = -Ls[c][xi];
#pragma unroll
			for (int k = 0; k < 8; ++k) sum[k] = fmaf(l, Rs[c][xi + (JD_D - 1) - (dg * 8 + k)], sum[k]);   // R[x - d], d = d0 + 8 dg + k
		}
	}
	const int x = x0 + xi;
	if (x >= W) return;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int d = d0 + dg * 8 + k;
		if (d < D && x - d >= 0) {
			const int64_t id = (int64_t)d * HW + (int64_t)y * W + x;
			volL[id] = sum[k];
			volR[id - d] = sum[k];
		}
	}
}

int stereo_join_dhw(const float *fL, const float *fR, float *volL, float *volR, int C, int D, int H, int W, hipStream_t st)
{
	hipLaunchKernelGGL(join_dhw_kernel, dim3(cdiv(W, JD_PX), H, cdiv(D, JD_D)), dim3(256), 0, st, fL, fR, volL, volR, C, D, H, W);
	return check_launch("stereo_join");
}

// ---- (H,W,ds) outputs for the fused pipeline: banded L x R^T on the matrix cores ---------------------
// Per image row y the cost band is S[x, x'] = -sum_c L[c,x] R[c,x'] for 0 <= x-x' < D: a dense
// (banded) GEMM with K = C.  v_mfma_f32_32x32x2_f32 evaluates, per output element,
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) (MI355X guide, "FP32-input MFMA": bit-for-bit a k-ordered
// fmaf chain), so chaining the k-steps with c ascending reproduces StereoJoin_'s
// `sum -= L*R` (adcensus.cu:1468-1471) bit for bit -- checked by tests/test_gpu_parity.py.
//
// One wave owns 32 consecutive LEFT pixels x of one image row (M side, operand A = -L, held in VGPRs for
// the whole band) and walks the right-image tiles x' = p0 - 32J .. (N side, operand B = R, prefetched one
// tile ahead so its latency hides under the 32 chained MFMAs of the current tile).  Every 32x32 tile is
// computed ONCE and stored twice:
//   left  volume: a register across lanes 0..31 is pixel x = p0+m, disparities d = m - n + 32J descending
//                 with the lane -> one 128-byte run of the (H,W,ds) volume;
//   right volume: the tile goes through a padded LDS transpose so that a register across lanes is pixel
//                 x' = q0+n with d ascending with the lane -> again one 128-byte run.
// Every (pixel, d < D) voxel of both volumes is written exactly once, NaN where the partner pixel lies
// outside the image (the reference's fill(0/0), main.lua:946): "virtual" tiles p0 >= W exist only to write
// the right volume's NaN triangle.  fix_border is a separate small copy.
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned cb_u4_t __attribute__((ext_vector_type(4)));

template <int KSTEPS>  // KSTEPS = ceil(C/2) rounded up to a supported size
__global__ void __launch_bounds__(256) join_mfma_kernel(const float *__restrict__ fL, const float *__restrict__ fR,
                                                        float *__restrict__ volL, float *__restrict__ volR, int C, int D,
                                                        int ds, int H, int W, int tiles_per_row)
{
	__shared__ float T[4][32 * 33];
	const int lane = threadIdx.x & 63;
	const int wid = threadIdx.x >> 6;
	// XCD-aware mapping: blocks b and b+8k share an XCD (b % 8); give all blocks of one image
	// row to one XCD so the row's features are fetched into one L2 only.
	const int b = blockIdx.x;
	const int xcd = b & 7, k = b >> 3;
	const int blocks_per_row = (tiles_per_row + 3) >> 2;
	const int y = (k / blocks_per_row) * 8 + xcd;
	const int tile = (k % blocks_per_row) * 4 + wid;
	if (y >= H || tile >= tiles_per_row) return;

	const int64_t HW = (int64_t)H * W;
	const float *__restrict__ fA = fL + (int64_t)y * W;
	const float *__restrict__ fB = fR + (int64_t)y * W;
	const int p0 = tile * 32;
	const int nl = lane & 31, kh = lane >> 5;
	const bool real_tile = p0 < W;  // wave-uniform; virtual tiles only fill the right volume's NaN triangle
	float *Tw = T[wid];

	// Features of this image row as raw buffers: (C,H,W) plane stride HW, so channel c of pixel x is at byte
	// ((c*HW) + x)*4 from the row's first pixel.  The lane part (pixel, channel parity) is the 32-bit voffset, the
	// channel-pair part 2*kk*HW*4 a scalar soffset; out-of-range lanes get an offset beyond the buffer -> 0.0.
	const unsigned feat_bytes = (unsigned)(((int64_t)(C - 1) * HW + W) * 4);  // < 2^31 checked by the launcher
	const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)fA, 0, feat_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)fB, 0, feat_bytes, 0x00020000);
	const unsigned pair_bytes = (unsigned)(2 * HW * 4);
	const unsigned OOBF = 0x80000000u;
	// A operand: -L[c = 2kk + kh][p0 + nl], zero outside the image / channel range
	float a[KSTEPS];
	{
		const int px = p0 + nl;
		const unsigned vo = (real_tile && px < W) ? (unsigned)((int64_t)kh * HW + px) * 4u : OOBF;
#pragma unroll
		for (int kk = 0; kk < KSTEPS; ++kk) {
			const int c = 2 * kk + kh;
			const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rA, c < C ? vo : OOBF, kk * pair_bytes, 0));
			a[kk] = -v;  // out-of-range loads return +0.0 -> -0.0: fma(-0.0, b, acc) == acc exactly as fma(0.0, b, acc)
		}
	}
	const float NANV = __builtin_nanf("");
	const int nJ = (D + 30) / 32 + 1;  // tiles J = 0..nJ-1 cover d up to 32*(nJ-1)+31 >= D-1

	auto load_b = [&](float (&bv)[KSTEPS], int J) {
		const int qx = p0 - 32 * J + nl;
		const bool qin = real_tile && qx >= 0 && qx < W;
		const unsigned vo = qin ? (unsigned)((int64_t)kh * HW + qx) * 4u : OOBF;
#pragma unroll
		for (int kk = 0; kk < KSTEPS; ++kk) {
			const int c = 2 * kk + kh;
			bv[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rB, c < C ? vo : OOBF, kk * pair_bytes, 0));
		}
	};

	const int row_bytes = W * ds * 4;  // one image row of a (H,W,ds) volume; < 2^31 checked by the launcher
	const __amdgpu_buffer_rsrc_t rowL = __builtin_amdgcn_make_buffer_rsrc((void *)(volL + (int64_t)y * W * ds), 0, row_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rowR = __builtin_amdgcn_make_buffer_rsrc((void *)(volR + (int64_t)y * W * ds), 0, row_bytes, 0x00020000);

	float bcur[KSTEPS], bnxt[KSTEPS];
	load_b(bcur, 0);
	for (int J = 0; J < nJ; ++J) {
		const int q0 = p0 - 32 * J;  // partner tile base (right-image pixels x')
		if (J + 1 < nJ) load_b(bnxt, J + 1);
		floatx16 acc;
#pragma unroll
		for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
		if (real_tile && q0 + 31 >= 0) {  // wave-uniform: skip tiles entirely left of the image (they store NaN)
#pragma unroll
			for (int kk = 0; kk < KSTEPS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bcur[kk], acc, 0, 0, 0);
		}
		// C/D layout: col n = lane & 31, row m = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
		// Stores are raw-buffer ops on this image row of the volume (32-bit offsets; an offset beyond the row's bytes
		// drops the lane), so no 64-bit address arithmetic or exec-mask branches per element.
		const int qn = q0 + nl;  // right-image pixel of this lane's column
		const bool qok = qn >= 0 && qn < W;
		if (real_tile) {
			// element (m, n): pixel x = p0+m, d = m - nl + 32J  ->  float offset (p0+m)*ds + d
			const int dJ = 32 * J - nl;
			int off = (p0 + 4 * kh) * ds + 4 * kh + dJ;  // m = 4*kh
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int m = (i & 3) + 8 * (i >> 2) + 4 * kh;
				const int d = m + dJ;
				const bool ok = p0 + m < W && d >= 0 && d < D;
				const int mo = (i & 3) + 8 * (i >> 2);  // compile-time part of m
				__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(qok ? acc[i] : NANV), rowL,
				                                      ok ? (unsigned)(off + mo * (ds + 1)) * 4u : 0x80000000u, 0, 0);
			}
		}
		// right volume: transpose through LDS (row stride 33: conflict-free both ways)
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int m = (i & 3) + 8 * (i >> 2) + 4 * kh;
			Tw[m * 33 + nl] = acc[i];
		}
		// (one wave owns Tw: LDS ops of a wave complete in order, no barrier needed)
		{
			const int mL = nl;                 // lane <-> left pixel offset m
			const bool pin = p0 + mL < W;      // the left pixel x = p0 + m exists
			// element (m = mL, n): pixel x' = q0+n, d = mL - n + 32J  ->  float offset (q0+n)*ds + d
			const int offR = (q0 + kh) * ds + mL - kh + 32 * J;   // n = kh
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int n = 2 * i + kh;      // right pixel offset handled by this half-wave in this step
				const int xr = q0 + n;
				const int d = mL - n + 32 * J;
				const bool ok = xr >= 0 && xr < W && d >= 0 && d < D;
				const float v = Tw[mL * 33 + n];
				__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pin ? v : NANV), rowR,
				                                      ok ? (unsigned)(offR + 2 * i * (ds - 1)) * 4u : 0x80000000u, 0, 0);
			}
		}
#pragma unroll
		for (int kk = 0; kk < KSTEPS; ++kk) bcur[kk] = bnxt[kk];
	}
}

// ---- (H,W,ds) outputs, line-aligned stores: every tile computed by the owner of each volume ---------------------
// PMC on join_mfma_kernel: 1.14 GB written and 0.42 GB read for 0.83 GB + 0.12 GB algorithmic -- its 128-byte store
// pieces start wherever d = m - n + 32J falls inside a pixel's 912-byte run, leave L2 as partial 32-byte sectors and
// come back as read-modify-writes.  Here a wave OWNS 32 pixels of ONE volume and walks all partner tiles J itself
// (SIDE 0: left pixels, partners x - d; SIDE 1: right pixels, partners x + d -- the MFMAs are issued twice per pair of
// volumes, 2 x 13 GFLOP is still < 0.25 ms of fp32 MFMA), so the 32 disparities a tile adds to each of its pixels
// extend that pixel's run contiguously.  They are parked in a 64-float LDS ring per pixel; after every tile the
// 128-byte LINES of the volume that have become complete are written, 8 lanes x 16 bytes per line -- full, aligned
// lines only, apart from the first and last line of a pixel's run.
// cache policy of the volume stores: nt (bit 1) -- the 2 V of output stream through L2 once and must not push the image
// row's features (re-read by every wave of the row) out of it
#define MC_JOIN_STORE_AUX 2

template <int KSTEPS, int SIDE, int NT>
__device__ __forceinline__ void join_owner_tiles(const float *__restrict__ fL, const float *__restrict__ fR, float *__restrict__ vol,
                                                 int C, int D, int ds, int H, int W, int y, int tile0, float *__restrict__ rings)
{
	// A wave owns TWO adjacent tiles (64 pixels) of one volume: both multiply against the same partner tile in the same
	// step (with disparity offsets one tile apart), so every partner tile is fetched once per two tile products.
	const int lane = threadIdx.x & 63;
	const int nl = lane & 31, kh = lane >> 5;
	const int64_t HW = (int64_t)H * W;
	const unsigned feat_bytes = (unsigned)(((int64_t)(C - 1) * HW + W) * 4);
	const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void *)(fL + (int64_t)y * W), 0, feat_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void *)(fR + (int64_t)y * W), 0, feat_bytes, 0x00020000);
	const unsigned pair_bytes = (unsigned)(2 * HW * 4);
	const unsigned OOBF = 0x80000000u;
	const float NANV = __builtin_nanf("");
	// operand of the tile starting at pixel p0 (lane part: pixel nl of the tile, channel parity kh); outside the image: 0
	auto load_tile = [&](float (&v)[KSTEPS], const __amdgpu_buffer_rsrc_t &r, int p0, bool negate) {
		const int px = p0 + nl;
		// the channel offset rides in the vector offset, which the buffer range check covers (the scalar offset is not
		// checked): channels c >= C lie beyond feat_bytes and read as 0 without a per-channel select
		const unsigned vo = (px >= 0 && px < W) ? (unsigned)((int64_t)kh * HW + px) * 4u : OOBF;
#pragma unroll
		for (int kk = 0; kk < KSTEPS; ++kk) {
			const float t = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo + (unsigned)kk * pair_bytes, 0, 0));
			v[kk] = negate ? -t : t;
		}
	};
	// (the minus sign of -L*R always goes to the OWN operand, loaded once: (-L)*R and L*(-R) are the same bits, and a
	// prefetched partner value that needs no arithmetic needs no wait until its MFMA)
	auto load_partner = [&](float (&v)[KSTEPS], int T) { load_tile(v, SIDE == 0 ? rB : rA, 32 * T, false); };
	float own[NT][KSTEPS], pa[KSTEPS], pb[KSTEPS];
#pragma unroll
	for (int t = 0; t < NT; ++t) load_tile(own[t], SIDE == 0 ? rA : rB, 32 * (tile0 + t), true);
	const int nJ = (D + 30) / 32 + 1;
	// partner tile of step s: SIDE 0 walks left from tile0+1, SIDE 1 right from tile0; own tile t sees it as its
	// J = (SIDE 0: tile0 + t - T, SIDE 1: T - tile0 - t), active while 0 <= J < nJ
	const int nsteps = nJ + NT - 1;
	auto partner_of = [&](int s) { return SIDE == 0 ? tile0 + NT - 1 - s : tile0 + s; };
	load_partner(pa, partner_of(0));

	// C/D layout of a tile: col n = lane & 31 (right pixel), row m = mc_i + 4 * (lane >> 5), mc_i = (i & 3) + 8 * (i >> 2)
	// (left pixel); d = m - n + 32J in both roles.  Ring slot of element i: (d & 63) in the row of its owner pixel
	// (SIDE 0: m, SIDE 1: n); J only toggles bit 5 of the slot.
	const int wbase = 4 * kh - nl;                       // d = wbase + mc_i + 32J
	const int wrow = SIDE == 0 ? 4 * kh * 64 : nl * 64;  // (+ mc_i * 64 for SIDE 0)

	// line writer: lane -> (pixel fo = 8*pass + lane/8, 16-byte piece fp = lane%8).  Pixel o holds every d <= 32J + c_o
	// after tile J (c_o = o for SIDE 0, 31 - o for SIDE 1); line k of pixel o covers d in [32k - a_o, 32k - a_o + 31],
	// a_o = offset of the pixel's run inside a 128-byte line.  Exactly one line per pixel completes per tile:
	// k = kl_o + J with kl_o = floor((c_o + a_o - 63) / 32) + 1; the last tile also completes the line after it.
	const int fp = lane & 7;
	const int row_floats = W * ds;
	const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void *)(vol + (int64_t)y * row_floats), 0, row_floats * 4, 0x00020000);
	const int row_mis = (int)(((int64_t)y * row_floats) & 31);
	int fd0[NT][4];          // first disparity of this lane's piece of the line that completes at J = 0 (may be negative)
	unsigned fgo[NT][4];     // byte offset of the pixel's run in the image row, or out of range if the pixel is outside the image
#pragma unroll
	for (int t = 0; t < NT; ++t) {
#pragma unroll
		for (int pass = 0; pass < 4; ++pass) {
			const int px = 32 * (tile0 + t) + pass * 8 + (lane >> 3);
			const int fo = pass * 8 + (lane >> 3);
			const int a_o = (row_mis + px * ds) & 31;
			const int c_o = SIDE == 0 ? fo : 31 - fo;
			const int kl = ((c_o + a_o - 63 + 64) >> 5) - 2 + 1;
			fd0[t][pass] = 32 * kl - a_o + 4 * fp;
			fgo[t][pass] = px < W ? (unsigned)(px * ds) * 4u : OOBF;
		}
	}
	auto tile_product = [&](int t, const float (&part)[KSTEPS], int p0, floatx16 &acc) {
#pragma unroll
		for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
		// partner tile entirely outside the image: nothing to multiply, the whole tile is NaN
		const bool any_in = SIDE == 0 ? (p0 + 31 >= 0) : (p0 < W);
		if (any_in) {
#pragma unroll
			for (int kk = 0; kk < KSTEPS; ++kk)
				acc = SIDE == 0 ? __builtin_amdgcn_mfma_f32_32x32x2f32(own[t][kk], part[kk], acc, 0, 0, 0)
				                : __builtin_amdgcn_mfma_f32_32x32x2f32(part[kk], own[t][kk], acc, 0, 0, 0);
		}
	};
	auto tile_store = [&](int t, int J, int p0, const floatx16 &acc) {
		float *__restrict__ ring = rings + t * (32 * 64);
		const int jbit = (J & 1) << 5;
		// interior tile: every d of it lies in [0, D) and every partner pixel inside the image
		const bool interior = J >= 1 && 32 * J + 31 < D && (SIDE == 0 ? p0 >= 0 : p0 + 31 < W);
		if (interior) {
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int mc = (i & 3) + 8 * (i >> 2);
				ring[wrow + (SIDE == 0 ? mc * 64 : 0) + (((wbase + mc) & 63) ^ jbit)] = acc[i];
			}
		} else {
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int mc = (i & 3) + 8 * (i >> 2);
				const int d = wbase + mc + 32 * J;
				const int po = SIDE == 0 ? p0 + nl : p0 + mc + 4 * kh;   // partner pixel of this element
				const bool pin = SIDE == 0 ? po >= 0 : po < W;           // (the other bound cannot be violated for d >= 0)
				if (d >= 0 && d < D) ring[wrow + (SIDE == 0 ? mc * 64 : 0) + (((wbase + mc) & 63) ^ jbit)] = pin ? acc[i] : NANV;
			}
		}
		const bool last = J + 1 == nJ;
#pragma unroll
		for (int pass = 0; pass < 4; ++pass) {
			const int d0 = fd0[t][pass] + 32 * J;
			const int fro = (pass * 8 + (lane >> 3)) * 64;
			// (unsigned)d0 < ds  <=>  the piece holds at least one d of [0, D) of THIS pixel (d0 % 4 == 0, ds = D rounded up to 4)
			const float4 v = *(const float4 *)(ring + fro + (d0 & 63));
			__builtin_amdgcn_raw_buffer_store_b128((cb_u4_t){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
			                                       rrow, ((unsigned)d0 < (unsigned)ds) ? fgo[t][pass] + (unsigned)d0 * 4u : OOBF, 0, MC_JOIN_STORE_AUX);
			if (last) {
				const int d1 = d0 + 32;
				const float4 u = *(const float4 *)(ring + fro + (d1 & 63));
				__builtin_amdgcn_raw_buffer_store_b128((cb_u4_t){__float_as_uint(u.x), __float_as_uint(u.y), __float_as_uint(u.z), __float_as_uint(u.w)},
				                                       rrow, ((unsigned)d1 < (unsigned)ds) ? fgo[t][pass] + (unsigned)d1 * 4u : OOBF, 0, MC_JOIN_STORE_AUX);
			}
		}
	};
	// Product and stores of one own tile after the other.  (Measured and rejected: the products of a step's own tiles first, the two chains
	// interleaved MFMA by MFMA, and the previous tile's line writer + the next partner tile's loads issued between the MFMAs of a chain --
	// profiles/r05_ab_join_order3.txt, profiles/r05_mfma_order.txt; the variants are in the history, commit 5cbe4ae.)
	auto do_step = [&](int s, const float (&part)[KSTEPS]) {
		const int T = partner_of(s);
		floatx16 acc[NT];
#pragma unroll
		for (int t = 0; t < NT; ++t) {
			const int J = SIDE == 0 ? tile0 + t - T : T - tile0 - t;
			if (J >= 0 && J < nJ) { tile_product(t, part, 32 * T, acc[0]); tile_store(t, J, 32 * T, acc[0]); }
		}
	};
	// The prefetch of the next partner tile is issued UNCONDITIONALLY (past the last step it fetches a tile that is never
	// multiplied): vmcnt counts in order, and a conditional batch of 32 loads makes the compiler wait, before every
	// MFMA of the step, as if the batch had not been issued -- i.e. for the loads it has just sent.
	for (int s = 0; s < nsteps; s += 2) {
		load_partner(pb, partner_of(s + 1));
		do_step(s, pa);
		load_partner(pa, partner_of(s + 2));
		if (s + 1 < nsteps) do_step(s + 1, pb);
	}
}

#define MC_JOIN_NT 2
// waves per block: each wave's two rings are 16 KB of LDS, so blocks of four hold a CU at 8 waves (2 per SIMD), blocks of two / one at 10
#define MC_JOIN_WPB 4
template <int KSTEPS, int NT = MC_JOIN_NT>
__global__ void __launch_bounds__(64 * MC_JOIN_WPB) join_owner_kernel(const float *__restrict__ fL, const float *__restrict__ fR,
                                                         float *__restrict__ volL, float *__restrict__ volR, int C, int D, int ds,
                                                         int H, int W, int pairs_per_row)
{
	__shared__ __attribute__((aligned(16))) float rings[MC_JOIN_WPB][NT * 32 * 64];
	const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	// XCD-aware mapping as in join_mfma_kernel: all blocks of one image row on one XCD
	const int b = blockIdx.x;
	const int xcd = b & 7, k = b >> 3;
	const int blocks_per_row = (2 * pairs_per_row + MC_JOIN_WPB - 1) / MC_JOIN_WPB;
	const int y = (k / blocks_per_row) * 8 + xcd;
	const int w = (k % blocks_per_row) * MC_JOIN_WPB + wid;
	if (y >= H || w >= 2 * pairs_per_row) return;
	if (w < pairs_per_row) join_owner_tiles<KSTEPS, 0, NT>(fL, fR, volL, C, D, ds, H, W, y, NT * w, rings[wid]);
	else join_owner_tiles<KSTEPS, 1, NT>(fL, fR, volR, C, D, ds, H, W, y, NT * (w - pairs_per_row), rings[wid]);
}

// fix_border (main.lua:922-927) on (H,W,ds): the n outermost pixels of one side replicate the
// (n+1)-th pixel's whole cost vector.  One wave per (row, border pixel).
__global__ void __launch_bounds__(256) fix_border_hwd_kernel(float *__restrict__ vol, int D, int ds, int H, int W, int n,
                                                             int direction)
{
	const int lane = threadIdx.x & 63;
	const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= (int64_t)H * n) return;
	const int y = (int)(w / n), i = (int)(w % n) + 1;
	const int dst = direction < 0 ? W - i : i - 1;
	const int src = direction < 0 ? W - (n + 1) : n;
	const float *s = vol + ((int64_t)y * W + src) * ds;
	float *t = vol + ((int64_t)y * W + dst) * ds;
	for (int d = lane; d < D; d += 64) t[d] = s[d];
}

int stereo_join_hwd(const float *fL, const float *fR, float *volL, float *volR, int C, int D, int ds, int H, int W, int n,
                    hipStream_t st)
{
	const int rows8 = (H + 7) / 8;
	const int ks = (C + 1) / 2;
	const dim3 block(256);
	if (ds % 4 == 0 && (uintptr_t)volL % 16 == 0 && (uintptr_t)volR % 16 == 0 && ks <= 32) {
		const int tiles_own = ((W + 31) / 32 + MC_JOIN_NT - 1) / MC_JOIN_NT;   // groups of MC_JOIN_NT 32-pixel tiles per image row and volume
		const dim3 grid_o((unsigned)(rows8 * ((2 * tiles_own + MC_JOIN_WPB - 1) / MC_JOIN_WPB) * 8)), block_o(64 * MC_JOIN_WPB);
		if (ks <= 8) hipLaunchKernelGGL((join_owner_kernel<8>), grid_o, block_o, 0, st, fL, fR, volL, volR, C, D, ds, H, W, tiles_own);
		else if (ks <= 16) hipLaunchKernelGGL((join_owner_kernel<16>), grid_o, block_o, 0, st, fL, fR, volL, volR, C, D, ds, H, W, tiles_own);
		else hipLaunchKernelGGL((join_owner_kernel<32>), grid_o, block_o, 0, st, fL, fR, volL, volR, C, D, ds, H, W, tiles_own);
		int rc = check_launch("stereo_join_hwd (owner tiles)");
		if (rc || n <= 0) return rc;
		hipLaunchKernelGGL(fix_border_hwd_kernel, dim3(cdiv((int64_t)H * n * 64, 256)), block, 0, st, volL, D, ds, H, W, n, -1);
		hipLaunchKernelGGL(fix_border_hwd_kernel, dim3(cdiv((int64_t)H * n * 64, 256)), block, 0, st, volR, D, ds, H, W, n, 1);
		return check_launch("fix_border_hwd");
	}
	const int tiles = (W + D - 1 + 31) / 32;  // incl. the virtual tiles right of the image (right volume's NaN triangle)
	const int blocks_per_row = (tiles + 3) / 4;
	const dim3 grid((unsigned)(rows8 * blocks_per_row * 8));
#define MC_JOIN_LAUNCH(KS)                                                                                                  \
	do {                                                                                                                    \
		hipLaunchKernelGGL((join_mfma_kernel<KS>), grid, block, 0, st, fL, fR, volL, volR, C, D, ds, H, W, tiles);             \
	} while (0)
	if (ks <= 8) MC_JOIN_LAUNCH(8);
	else if (ks <= 16) MC_JOIN_LAUNCH(16);
	else if (ks <= 32) MC_JOIN_LAUNCH(32);
	else if (ks <= 56) MC_JOIN_LAUNCH(56);
	else MC_JOIN_LAUNCH(64);
#undef MC_JOIN_LAUNCH
	int rc = check_launch("stereo_join_hwd");
	if (rc || n <= 0) return rc;
	hipLaunchKernelGGL(fix_border_hwd_kernel, dim3(cdiv((int64_t)H * n * 64, 256)), block, 0, st, volL, D, ds, H, W, n, -1);
	hipLaunchKernelGGL(fix_border_hwd_kernel, dim3(cdiv((int64_t)H * n * 64, 256)), block, 0, st, volR, D, ds, H, W, n, 1);
	return check_launch("fix_border_hwd");
}

// ---- census, signature form ---------------------------------------------------------------------
// The 81 comparisons x[q] < x[p] of a pixel's 9x9 window do not depend on the disparity: census_sig_kernel packs them
// once per pixel and channel into 81 bits (3 words; taps outside the image are 0) together with the window's
// in-image mask.  The cost of voxel (d,y,x) is then, per channel,
//     (81 - popcount(V)) + popcount((S0[y,x] ^ S1[y,x+d]) & V),   V = mask[y,x] & mask[y,x+d]
// -- the reference's count of out-of-bounds taps plus differing in-bounds taps (adcensus.cu:128-144): small integers,
// so the float result (sum, then / channels) is bit-identical.
__global__ void __launch_bounds__(256) census_sig_kernel(const float *__restrict__ img, uint32_t *__restrict__ sig,
                                                         uint32_t *__restrict__ mask, int Cimg, int H, int W)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t HW = (int64_t)H * W;
	if (id >= (int64_t)Cimg * HW) return;
	const int c = (int)(id / HW);
	const int64_t pix = id - c * HW;
	const int x = (int)(pix % W), y = (int)(pix / W);
	const float *__restrict__ im = img + c * HW;
	const float p = im[pix];
	uint32_t s[3] = {0, 0, 0}, m[3] = {0, 0, 0};
	int t = 0;
	for (int yy = y - 4; yy <= y + 4; yy++) {
		for (int xx = x - 4; xx <= x + 4; xx++, t++) {
			if (0 <= xx && xx < W && 0 <= yy && yy < H) {
				m[t >> 5] |= 1u << (t & 31);
				if (im[(int64_t)yy * W + xx] < p) s[t >> 5] |= 1u << (t & 31);
			}
		}
	}
	uint32_t *so = sig + id * 3;
	so[0] = s[0]; so[1] = s[1]; so[2] = s[2];
	if (c == 0) {
		uint32_t *mo = mask + pix * 3;
		mo[0] = m[0]; mo[1] = m[1]; mo[2] = m[2];
	}
}

__global__ void __launch_bounds__(256) census_cost_kernel(const uint32_t *__restrict__ sig0, const uint32_t *__restrict__ sig1,
                                                          const uint32_t *__restrict__ mask, float *__restrict__ out, int Cimg, int D,
                                                          int H, int W, int direction)
{
	const int x = blockIdx.x * 256 + threadIdx.x;
	const int y = blockIdx.y;
	const int d = blockIdx.z * direction;
	if (x >= W) return;
	const int64_t HW = (int64_t)H * W;
	const int64_t pix = (int64_t)y * W + x;
	float dist;
	if (0 <= x + d && x + d < W) {
		const uint32_t *m0 = mask + pix * 3, *m1 = mask + (pix + d) * 3;
		const uint32_t v0 = m0[0] & m1[0], v1 = m0[1] & m1[1], v2 = m0[2] & m1[2];
		const int oob = 81 - (__popc(v0) + __popc(v1) + __popc(v2));
		dist = 0;
		for (int c = 0; c < Cimg; c++) {
			const uint32_t *a = sig0 + (c * HW + pix) * 3, *b = sig1 + (c * HW + pix + d) * 3;
			const int diff = __popc((a[0] ^ b[0]) & v0) + __popc((a[1] ^ b[1]) & v1) + __popc((a[2] ^ b[2]) & v2);
			// the reference counts in a float accumulator (dist++); every partial count is an integer <= 81*Cimg, exact
			dist += (float)(oob + diff);
		}
		dist /= (float)Cimg;
	} else {
		dist = __builtin_nanf("");
	}
	out[(int64_t)blockIdx.z * HW + pix] = dist;
}

size_t census_scratch_bytes(int Cimg, int H, int W)
{
	return (((size_t)2 * Cimg + 1) * H * W * 3 * sizeof(uint32_t) + 255) & ~(size_t)255;
}

int census_sig(const float *x0, const float *x1, float *vol, void *scratch, int Cimg, int D, int H, int W, int direction,
               hipStream_t st)
{
	const int64_t HW = (int64_t)H * W;
	uint32_t *s0 = (uint32_t *)scratch, *s1 = s0 + (size_t)Cimg * HW * 3, *mk = s1 + (size_t)Cimg * HW * 3;
	hipLaunchKernelGGL(census_sig_kernel, dim3(cdiv((int64_t)Cimg * HW, 256)), dim3(256), 0, st, x0, s0, mk, Cimg, H, W);
	hipLaunchKernelGGL(census_sig_kernel, dim3(cdiv((int64_t)Cimg * HW, 256)), dim3(256), 0, st, x1, s1, mk, Cimg, H, W);
	hipLaunchKernelGGL(census_cost_kernel, dim3(cdiv(W, 256), H, D), dim3(256), 0, st, s0, s1, mk, vol, Cimg, D, H, W, direction);
	return check_launch("census");
}

// ---- ad, LDS form ---------------------------------------------------------------------------------
// One block = a 16 x 64 pixel tile of one disparity plane: |x0 - x1(x+d)| for the tile and its 4-pixel frame is
// staged once in LDS (NaN where the reference's bounds test fails), then every voxel adds its 81 taps in the
// reference's order (yy outer, xx inner; adcensus.cu:71-84) and divides by the number of in-bounds taps.
__global__ void __launch_bounds__(256) ad_tile_kernel(const float *__restrict__ x0, const float *__restrict__ x1, float *__restrict__ out,
                                                      int D, int H, int W, int direction)
{
	constexpr int TY = 16, TX = 64, R = 4, TS = TX + 2 * R + 1;
	__shared__ float A[(TY + 2 * R) * TS];
	const int bx = blockIdx.x * TX, by = blockIdx.y * TY;
	const int d = blockIdx.z * direction;
	const float NANV = __builtin_nanf("");
	for (int i = threadIdx.x; i < (TY + 2 * R) * (TX + 2 * R); i += 256) {
		const int ty = i / (TX + 2 * R), tx = i - ty * (TX + 2 * R);
		const int yy = by + ty - R, xx = bx + tx - R;
		float v = NANV;
		if (0 <= xx && xx < W && 0 <= xx + d && xx + d < W && 0 <= yy && yy < H) {
			const int64_t ind = (int64_t)yy * W + xx;
			v = fabsf(x0[ind] - x1[ind + d]);
		}
		A[ty * TS + tx] = v;
	}
	__syncthreads();
	const int lx = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int x = bx + lx;
	if (x >= W) return;
	const int64_t HW = (int64_t)H * W;
	for (int r = wv; r < TY; r += 4) {
		const int y = by + r;
		if (y >= H) break;
		float dist;
		if (0 <= x + d && x + d < W) {
			int cnt = 0;
			dist = 0;
			const float *base = A + r * TS + lx;
#pragma unroll
			for (int j = 0; j < 2 * R + 1; ++j) {
#pragma unroll
				for (int k = 0; k < 2 * R + 1; ++k) {
					const float v = base[j * TS + k];
					const bool ok = v == v;
					dist += ok ? v : -0.0f;   // x + -0.0 == x: a skipped tap leaves the accumulator untouched
					cnt += ok ? 1 : 0;
				}
			}
			dist /= (float)cnt;
		} else {
			dist = NANV;
		}
		out[(int64_t)blockIdx.z * HW + (int64_t)y * W + x] = dist;
	}
}

int ad_tiled(const float *x0, const float *x1, float *vol, int D, int H, int W, int direction, hipStream_t st)
{
	hipLaunchKernelGGL(ad_tile_kernel, dim3(cdiv(W, 64), cdiv(H, 16), D), dim3(256), 0, st, x0, x1, vol, D, H, W, direction);
	return check_launch("ad");
}

}  // namespace mc
