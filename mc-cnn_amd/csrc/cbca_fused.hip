// Cross-based cost aggregation (adcensus.cu:343-377): TWO iterations per pass over the volume, driven by a once-per-pair
// classification of every output's support (gfx950).
//
// adcensus.cbca is applied 2 + 16 times per volume on the accurate Middlebury configuration (main.lua:132-135,
// 998-1001, 1033-1039); each application is a pure function of the previous volume, and the supports depend on the two
// images' arms and the disparity only -- they are the same for every iteration of a pair.
//
//   cbca_classify_kernel (once per pair and direction) leaves, per (plane, strip, row), four 64-bit lane masks --
//   need[j]: the output in frame column 4*lane + j has a support other than the minimal 3x3 (and exists and has its
//   partner inside the image) -- and, for every flagged output, one 32-bit DESCRIPTOR in a (D,H,W) array: bit 5*k + t
//   = tap (row y-2+k, column x-2+t) belongs to the support, bits 25..29 = the number of taps, 0 = the support does
//   not fit that 5 x 5 window.  It also counts the flagged and the non-fitting outputs (the density gate below).
//
//   cbca_fused2_kernel reads V_k and writes V_(k+2); V_(k+1) lives only in registers and a wave-private LDS ring, so
//   two iterations cost 2 V of HBM traffic instead of 4 V.  One wave = one disparity plane x a strip of 256 staged
//   columns x RB output rows, walked top to bottom with no block barrier; rows are fetched 6 ahead (a wave keeps 6 KB
//   of reads in flight: with 3 the same data movement measured 20 % slower), the masks arrive by scalar loads, the
//   descriptor of a lane's first flagged output travels with the row prefetch so that vector loads are consumed in
//   issue order.  Per new input row r:
//     stage 1, row r-2: minimal supports out of registers (the lane's four columns of three rows as five adjacent column
//       pairs, outer columns from the neighbour lanes by DPP; nine packed v_pk_add_f32 per output pair in the reference's
//       order; division by 9 as three packed operations inside the range the form is proven for, IEEE divide outside);
//       flagged outputs re-evaluated by the lane that owns them: 5 x 5 window of V_k from the LDS ring with immediate
//       offsets (the row loop is unrolled by the ring size), each tap kept or replaced by -0.0f according to its
//       descriptor bit, added in the reference's order, IEEE divide by the tap count; descriptor 0: the reference's loop
//       over ring rows / global V_k.  The finished row of V_(k+1) goes to registers and to the V_(k+1) ring.
//     stage 2, row r-4: the same from the V_(k+1) registers / ring; descriptor 0: every V_(k+1) value of the support
//       that is not in the ring (or outside the columns this wave owns) is rebuilt from global V_k by the reference's
//       loop -- nested, exact, rare on textured images.
//   V_(k+1) is valid in frame columns 4..251, V_(k+2) is produced for frame columns 8..247 (240 per strip).
//   Every voxel of V_(k+2) is the reference's expression on the reference's V_(k+1) values: bit-identical.
//
//   Density gate: on images where many outputs are flagged (or do not fit the window) the per-row re-evaluation passes
//   dominate and two plain iterations of cbca_strip_kernel are faster; mc_predict enqueues both forms and each kernel
//   looks at the classification counters first -- exactly one of the two does the work (cbca_gate_*).
#include "cbca_common.h"
#include <algorithm>

namespace mc {

constexpr int F2_STEP = 240;   // output columns per strip (frame columns 8 .. 247)
constexpr int F2_HALO = 8;
constexpr int F2_RING = 6;     // rows per LDS ring = unroll factor of the row loop (multiple of 3: the register windows)
constexpr int F2_NMASK = 4;    // lane masks per (plane, strip, row)
constexpr int F2_HDR = 256;    // bytes of counters in front of the masks: [0] flagged outputs, [1] outputs that do not fit

typedef unsigned long long bm_mask;
struct C2Row { cb_f2 A, B, C, D, E; };   // columns (-1,0) (0,1) (1,2) (2,3) (3,4) relative to the lane's first column

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

// per-lane select by a 64-bit lane mask held in SGPRs: bit set -> b, clear -> a (one v_cndmask, no compare)
__device__ __forceinline__ float sel_f(bm_mask m, float a, float b)
{
	float d;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m));
	return d;
}
__device__ __forceinline__ int sel_i(bm_mask m, int a, int b)
{
	int d;
	asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m));
	return d;
}

// s / 9 for two sums at once: q = s*r, e = fma(-9, q, s), q' = fma(e, r, q) with r = RN(1/9).  Equal to the IEEE
// quotient for every float with 2^-95 <= |s| < 2^125 (mc_selftest_div9 walks all 2^32 bit patterns).
__device__ __forceinline__ cb_f2 div9_pk(cb_f2 s)
{
	const float r9 = 0x1.c71c72p-4f;  // RN(1/9)
	const cb_f2 r = cb_f2{r9, r9};
	const cb_f2 q = s * r;
	const cb_f2 e = __builtin_elementwise_fma(cb_f2{-9.0f, -9.0f}, q, s);
	return __builtin_elementwise_fma(e, r, q);
}
__device__ __forceinline__ bool div9_in_range(float s) { return __builtin_fabsf(s) >= 0x1p-95f && __builtin_fabsf(s) < 0x1p125f; }
// lane mask of the lanes whose s lies OUTSIDE that range: two instructions (the sign is shifted out, the biased exponent
// 32 .. 251 is the top byte of the rest: one add moves 32 to zero, one unsigned compare against 220 << 24)
__device__ __forceinline__ unsigned long long div9_bad(float s)
{
	unsigned t;
	unsigned long long m;
	asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t) : "v"(s), "s"(0xE0000000u));   // (bits << 1) - (32 << 24)
	asm("v_cmp_le_u32_e64 %0, %1, %2" : "=s"(m) : "s"(0xDC000000u), "v"(t));       // 220 << 24 <= t
	return m;
}

__global__ void __launch_bounds__(256) div9_selftest_kernel(uint32_t first, uint64_t count, unsigned long long *__restrict__ bad_in,
                                                            unsigned long long *__restrict__ bad_out, uint32_t *__restrict__ example)
{
	// every bit pattern first .. first+count-1: inside the guarded range the packed form must equal IEEE s / 9
	unsigned long long nin = 0, nout = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t bits = first + (uint32_t)i;
		const float s = __uint_as_float(bits);
		const cb_f2 q = div9_pk(cb_f2{s, -s});
		const float want = s / 9.0f;
		const bool same = (__float_as_uint(q.x) == __float_as_uint(want) || (want != want && q.x != q.x)) &&
		                  (__float_as_uint(q.y) == __float_as_uint(-want) || (want != want && q.y != q.y));
		if (!same) {
			if (div9_in_range(s)) { ++nin; *example = bits; }
			else ++nout;
		}
	}
	if (nin) atomicAdd(bad_in, nin);
	if (nout) atomicAdd(bad_out, nout);
}

int div9_selftest(uint32_t first, uint64_t count, unsigned long long *counters, hipStream_t st)
{
	hipLaunchKernelGGL(div9_selftest_kernel, dim3(4096), dim3(256), 0, st, first, count, counters, counters + 1, (uint32_t *)(counters + 2));
	return check_launch("div9_selftest");
}


struct F2Args {
	CbcaArgs c;
	unsigned long long *counters;   // [0] flagged outputs, [1] flagged outputs whose support does not fit the window
	bm_mask *masks;                  // [plane][strip][row][F2_NMASK]
	cb_u32 *desc;                    // (D,H,W) support descriptors of the flagged outputs
	unsigned long long max_flagged, max_unfit;   // the fused kernel stands down above these counts
};

__device__ __forceinline__ void f2_wave_coords(const CbcaArgs &A, int &d, int &cx, int &cy, bool &live)
{
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int dgroups = (A.nd + 3) >> 2;
	const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
	const int region = (kb / dgroups) * 8 + xcd;
	d = A.d0 + (kb % dgroups) * 4 + wv;
	live = region < A.gx * A.gy && d < A.d0 + A.nd;
	cx = region % A.gx;
	cy = region / A.gx;
}

// the reference's loop for voxel (d, q, x) of the INPUT volume, everything from global memory (adcensus.cu:356-373)
static __device__ float cbca_point_global(const uint32_t *__restrict__ p0, const uint32_t *__restrict__ p1,
                                          const float *__restrict__ plane, int W, int sh, int q, int x)
{
	const int g0 = q * W + x;
	const cb_u32 own = bytemin4(p0[g0], p1[g0 + sh]);
	const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
	float sum = 0;
	int cnt = 0;
	for (int qq = q - u; qq <= q + dn; ++qq) {
		const int g = qq * W + x;
		const cb_u32 mm = bytemin4(p0[g], p1[g + sh]);
		const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
		const float *row = plane + g - l;
		const int n = l + rg + 1;
		for (int k = 0; k < n; ++k) sum += row[k];
		cnt += n;
	}
	return sum / (float)cnt;
}

// Rare paths.
struct F2Ctx {
	const uint32_t *p0, *p1;
	const float *plane_in;
	int W, sh, xs0, ra;
};

// stage 1, any support: lengths from global memory, values from the V_k ring rows [lo_row, hi_row] where the run lies
// inside the staged columns, else global V_k
static __device__ __forceinline__ float f2_general1(const F2Ctx &X, const float (*V0)[CS_COLS], int yo, int c, int lo_row, int hi_row)
{
	const int x = X.xs0 + c;
	const int g0 = yo * X.W + x;
	const cb_u32 own = bytemin4(X.p0[g0], X.p1[g0 + X.sh]);
	const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
	float sum = 0;
	int cnt = 0;
	for (int q = yo - u; q <= yo + dn; ++q) {
		const int g = q * X.W + x;
		const cb_u32 mm = bytemin4(X.p0[g], X.p1[g + X.sh]);
		const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
		const int n = l + rg + 1;
		if (q >= lo_row && q <= hi_row && c - l >= 0 && c + rg < CS_COLS) {
			const float *row = &V0[(int)((unsigned)(q - X.ra) % (unsigned)F2_RING)][c - l];
			for (int k = 0; k < n; ++k) sum += row[k];
		} else {
			const float *row = X.plane_in + g - l;
			for (int k = 0; k < n; ++k) sum += row[k];
		}
		cnt += n;
	}
	return sum / (float)cnt;
}

// stage 2, any support: V_(k+1) values inside the final ring rows [lo1, hi1] / owned columns 4..251 from LDS (V_(k+1) row q
// sits in the slot of the iteration that produced it, i.e. of input row q+2), every other one rebuilt from global V_k
static __device__ __forceinline__ float f2_general2(const F2Ctx &X, const float (*V1)[CS_COLS], int yo, int c, int lo1, int hi1)
{
	const int x = X.xs0 + c;
	const int g0 = yo * X.W + x;
	const cb_u32 own = bytemin4(X.p0[g0], X.p1[g0 + X.sh]);
	const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
	float sum = 0;
	int cnt = 0;
	for (int q = yo - u; q <= yo + dn; ++q) {
		const int g = q * X.W + x;
		const cb_u32 mm = bytemin4(X.p0[g], X.p1[g + X.sh]);
		const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
		const bool row_in = q >= lo1 && q <= hi1;
		const int slot = (int)((unsigned)(q + 2 - X.ra) % (unsigned)F2_RING);
		for (int t = -l; t <= rg; ++t) {
			const int cc = c + t;
			float v;
			if (row_in && cc >= 4 && cc <= 251) v = V1[slot][cc];
			else v = cbca_point_global(X.p0, X.p1, X.plane_in, X.W, X.sh, q, x + t);
			sum += v;
		}
		cnt += l + rg + 1;
	}
	return sum / (float)cnt;
}

// ---- classification, once per pair and direction -----------------------------------------------------------------
// One wave per (plane, strip, chunk of rows), rows walked top to bottom with the byte-minimum arm lengths of the last five
// rows in registers (the support of an output is described by the lengths in ITS column on the rows it spans).
__global__ void __launch_bounds__(256) cbca_classify_kernel(const F2Args B)
{
	const CbcaArgs &A = B.c;
	const int lane = threadIdx.x & 63;
	int d, cx, cy;
	bool live;
	f2_wave_coords(A, d, cx, cy, live);
	if (!live) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs0 = cx * F2_STEP - F2_HALO;
	const int xs = xs0 + 4 * lane;
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const cb_u32 OOB = 0x80000000u;
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(B.desc + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const bool has1 = lane >= 1 && lane <= 62;   // lanes that own intermediate (stage-1) columns
	const bool has2 = lane >= 2 && lane <= 61;   // lanes that own output columns (counted once per strip)
	bool ok[4];   // column exists and its shifted partner is inside the image (adcensus.cu:353-354)
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		ok[j] = has1 && x >= 0 && x < W && x + sh >= 0 && x + sh < W;
	}
	bm_mask *__restrict__ out = B.masks + ((size_t)(d * A.gx + cx) * H) * F2_NMASK;
	cb_u32 m[5][4];   // rows r-4 .. r (m[4] = newest)
#pragma unroll
	for (int k = 0; k < 5; ++k)
#pragma unroll
		for (int j = 0; j < 4; ++j) m[k][j] = 0;
	unsigned n_flagged = 0, n_unfit = 0;
	for (int r = y0 - 2; r <= y1 + 1; ++r) {   // row r completes the window of row r-2
#pragma unroll
		for (int k = 0; k < 4; ++k)
#pragma unroll
			for (int j = 0; j < 4; ++j) m[k][j] = m[k + 1][j];
		const bool rok = r >= 0 && r < H;
		const cb_u32 vo = rok ? (cb_u32)((r * W + xs0) * 4 + lane * 16) : OOB;
		const cb_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rp0, vo, CS_PAD * 4, 0);
		const cb_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rp1, vo, (sh + CS_PAD) * 4, 0);
		const cb_u4 mn = bytemin4x4_sdwa(a, b);
		m[4][0] = mn.x; m[4][1] = mn.y; m[4][2] = mn.z; m[4][3] = mn.w;
		const int yo = r - 2;
		if (yo < y0 || yo >= y1) continue;
		bm_mask need_m[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const cb_u32 own = m[2][j];
			// minimal <=> own arms all 1 and the rows above / below have left = right = 1 in this column
			const bool minimal = own == 0x01010101u && (m[1][j] & 0xffffu) == 0x0101u && (m[3][j] & 0xffffu) == 0x0101u;
			const bool need = ok[j] && !minimal;
			const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
			bool fits = u <= 2 && dn <= 2;
			cb_u32 bits = 0;
			int cnt = 0;
#pragma unroll
			for (int k = 0; k < 5; ++k) {
				const int rel = k - 2;
				const bool act = rel < 0 ? u >= -rel : (rel == 0 ? true : dn >= rel);
				const int l = (int)(m[k][j] & 0xff), rg = (int)((m[k][j] >> 8) & 0xff);
				fits = fits && (!act || (l <= 2 && rg <= 2));
				cb_u32 rowbits = 0;
#pragma unroll
				for (int t = 0; t < 5; ++t) {
					const int dx = t - 2;
					const bool in = dx < 0 ? l >= -dx : (dx == 0 ? true : rg >= dx);
					rowbits |= in ? (1u << t) : 0u;
				}
				bits |= act ? (rowbits << (5 * k)) : 0u;
				cnt += act ? l + rg + 1 : 0;
			}
			need_m[j] = __ballot(need);
			if (need) __builtin_amdgcn_raw_buffer_store_b32(fits ? (bits | ((cb_u32)cnt << 25)) : 0u, rd, (cb_u32)((yo * W + xs + j) * 4), 0, 0);
			n_flagged += (need && has2) ? 1u : 0u;
			n_unfit += (need && has2 && !fits) ? 1u : 0u;
		}
		if (lane < F2_NMASK) {
			bm_mask v = need_m[0];
			v = lane == 1 ? need_m[1] : v; v = lane == 2 ? need_m[2] : v; v = lane == 3 ? need_m[3] : v;
			out[(size_t)yo * F2_NMASK + lane] = v;
		}
	}
	// one pair of atomics per wave
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		n_flagged += __shfl_xor(n_flagged, o);
		n_unfit += __shfl_xor(n_unfit, o);
	}
	if (lane == 0) {
		if (n_flagged) atomicAdd(B.counters + 0, (unsigned long long)n_flagged);
		if (n_unfit) atomicAdd(B.counters + 1, (unsigned long long)n_unfit);
	}
}

struct F2Lds {
	float V0[4][F2_RING][CS_COLS];
	float V1[4][F2_RING][CS_COLS];
};

template <bool NT>
__global__ void __launch_bounds__(256) cbca_fused2_kernel(const F2Args B)
{
	const CbcaArgs &A = B.c;
	constexpr int VOL_AUX = NT ? 2 : 0;
	constexpr int RG = F2_RING;
	__shared__ F2Lds S;
	{   // density gate: dense / long-armed images are left to the plain iterations (wave-uniform, scalar loads)
		typedef const __attribute__((address_space(4))) unsigned long long *cnt_ptr;
		const cnt_ptr cn = (cnt_ptr)B.counters;
		if (cn[0] > B.max_flagged || cn[1] > B.max_unfit) return;
	}
	const int lane = threadIdx.x & 63;
	int d, cx, cy;
	bool live;
	f2_wave_coords(A, d, cx, cy, live);
	if (!live) return;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float (*__restrict__ V0)[CS_COLS] = S.V0[wv];
	float (*__restrict__ V1)[CS_COLS] = S.V1[wv];
	const int H = A.H, W = A.W, direction = A.direction;
	const int HWi = H * W;
	const int sh = d * direction;
	const int xs0 = cx * F2_STEP - F2_HALO;             // image column of frame column 0 (wave-uniform)
	const int xs = xs0 + 4 * lane;                      // image column of this lane's first column
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const int ra = y0 - 6;                              // first staged row: V_(k+1) rows from y0-2 on, their windows from y0-4 on,
	                                                    // two more so that the register window of row y0-3 is complete
	const int plane_bytes = HWi * 4;
	const cb_u32 OOB = 0x80000000u;
	const float *__restrict__ plane_in = A.vin + (size_t)d * HWi;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)plane_in, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(B.desc + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	typedef const __attribute__((address_space(4))) bm_mask *mask_ptr;
	const mask_ptr masks = (mask_ptr)(B.masks + ((size_t)(d * A.gx + cx) * H) * F2_NMASK);
	const bm_mask OUT_LANES = 0x3ffffffffffffffcull;   // lanes 2..61 own output columns
	const bool interior = xs0 >= 0 && xs0 + CS_COLS <= W;   // wave-uniform
	const bool full_in = xs >= 0 && xs + 3 < W;
	const bool has2 = lane >= 2 && lane <= 61;
	const bool full_out = has2 && xs + 3 < W;
	const bool any_out = has2 && xs < W;
	// does this strip hold columns whose partner lies outside the image (they are copied through, adcensus.cu:353-354)?
	bool copy_any = false;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		copy_any = copy_any || (x >= 0 && x < W && !(x + sh >= 0 && x + sh < W));
	}
	const bool any_copy = __any(copy_any);   // wave-uniform
	const int lane16 = lane * 16;

	// Row r of the plane -> registers (rows outside the image: zeros), together with the descriptor of this lane's first
	// flagged column of the row whose stage 1 runs when row r is committed (r - 2).  Fetching both at the same distance
	// keeps vector loads consumed in issue order: a wait for the descriptor never drains the rows prefetched behind it.
	auto fetch = [&](cb_u4 &v, cb_u32 &dsc, int r) {
		{
			const int yn = r - 2;
			dsc = 0;
			if (yn >= 0 && yn < H && yn >= y0 - 2) {
				const bm_mask a0 = masks[yn * F2_NMASK + 0], a1 = masks[yn * F2_NMASK + 1], a2 = masks[yn * F2_NMASK + 2], a3 = masks[yn * F2_NMASK + 3];
				const bm_mask an = a0 | a1 | a2 | a3;
				if (an != 0) {
					const int jn = sel_i(a0, sel_i(a1, sel_i(a2, 3, 2), 1), 0);
					const cb_u32 off = sel_i(an, 0, 1) ? (cb_u32)((yn * W + xs0 + 4 * lane + jn) * 4) : OOB;
					dsc = __builtin_amdgcn_raw_buffer_load_b32(rd, off, 0, 0);
				}
			}
		}
		const bool rok = r >= 0 && r < H;
		const int rowoff = (r * W + xs0) * 4;            // scalar
		const cb_u32 vo = rok ? (cb_u32)(rowoff + lane16) : OOB;
		if (interior || full_in) {
			v = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, VOL_AUX);
		} else {  // image edges: per column
			cb_u32 t[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok && xs + k >= 0 && xs + k < W) ? vo + 4u * k : OOB, 0, 0);
			v = cb_u4{t[0], t[1], t[2], t[3]};
		}
	};

	auto make_row = [&](C2Row &nw, float v0, float v1, float v2, float v3) {
		const float l3 = lane_from_below(v3, 0.0f), r0 = lane_from_above(v0, 0.0f);
		nw.A = cb_f2{l3, v0}; nw.B = cb_f2{v0, v1}; nw.C = cb_f2{v1, v2}; nw.D = cb_f2{v2, v3}; nw.E = cb_f2{v3, r0};
	};

	// minimal 3x3 quotients of a lane's four columns; n0..n3: flagged columns (their value is replaced later)
	auto skeleton = [&](const C2Row &up, const C2Row &own, const C2Row &dn_, bm_mask n0, bm_mask n1, bm_mask n2, bm_mask n3,
	                    float &res0, float &res1, float &res2, float &res3) {
		cb_f2 s01 = cb_f2{0.0f, 0.0f}, s23 = cb_f2{0.0f, 0.0f};
		s01 += up.A; s01 += up.B; s01 += up.C;
		s23 += up.C; s23 += up.D; s23 += up.E;
		s01 += own.A; s01 += own.B; s01 += own.C;
		s23 += own.C; s23 += own.D; s23 += own.E;
		s01 += dn_.A; s01 += dn_.B; s01 += dn_.C;
		s23 += dn_.C; s23 += dn_.D; s23 += dn_.E;
		const cb_f2 q01 = div9_pk(s01), q23 = div9_pk(s23);
		res0 = q01.x; res1 = q01.y; res2 = q23.x; res3 = q23.y;
		// a sum outside [2^-95, 2^125) (zero, tiny, huge, inf, nan) in an unflagged column: IEEE divide for the row
		// (columns that do not exist or are copied through count too: conservative, never wrong)
		const bm_mask odd = (div9_bad(s01.x) & ~n0) | (div9_bad(s01.y) & ~n1) | (div9_bad(s23.x) & ~n2) | (div9_bad(s23.y) & ~n3);
		if (odd != 0) {
			res0 = s01.x / 9.0f; res1 = s01.y / 9.0f; res2 = s23.x / 9.0f; res3 = s23.y / 9.0f;
		}
		if (any_copy) {  // adcensus.cu:353-354: columns whose partner lies outside the image are copied through
			const int xp = xs + sh;
			res0 = (xp + 0 >= 0 && xp + 0 < W) ? res0 : own.B.x; res1 = (xp + 1 >= 0 && xp + 1 < W) ? res1 : own.B.y;
			res2 = (xp + 2 >= 0 && xp + 2 < W) ? res2 : own.D.x; res3 = (xp + 3 >= 0 && xp + 3 < W) ? res3 : own.D.y;
		}
	};

	// descriptor-driven window sum: rows yo-2 .. yo+2 of ring VR in slots s0 .. s0+4 (mod RING, static), columns c-2 .. c+2
#define F2_WINDOW(VR, UU, OUTV)                                                                              \
	do {                                                                                                     \
		float sum_;                                                                                          \
		{                                                                                                    \
			const cb_u32 keep0 = (cb_u32)(((int)(desc << 31)) >> 31);                                        \
			const cb_u32 tb0 = __float_as_uint(VR[((UU) + RG - 4) % RG][c - 2]);                              \
			const float m0_ = __uint_as_float((tb0 & keep0) | (0x80000000u & ~keep0));                       \
			asm("v_add_f32 %0, 0, %1" : "=v"(sum_) : "v"(m0_));   /* the reference's accumulator starts at +0.0 */ \
		}                                                                                                    \
		_Pragma("unroll") for (int k = 0; k < 5; ++k) {                                                      \
			_Pragma("unroll") for (int t = 0; t < 5; ++t) {                                                  \
				if (k == 0 && t == 0) continue;                                                              \
				const cb_u32 keep = (cb_u32)(((int)(desc << (31 - (5 * k + t)))) >> 31);                      \
				const cb_u32 tb = __float_as_uint(VR[((UU) + RG - 4 + k) % RG][c + t - 2]);                   \
				sum_ += __uint_as_float((tb & keep) | (0x80000000u & ~keep));                                \
			}                                                                                                \
		}                                                                                                    \
		OUTV = sum_ / (float)(desc >> 25);                                                                   \
	} while (0)

	const F2Ctx X = {A.p0, A.p1, plane_in, W, sh, xs0, ra};
	constexpr int PF = 6;
	cb_u4 st[PF];
	cb_u32 dsc[PF];
	C2Row w0[3], w1[3];
#pragma unroll
	for (int u = 0; u < PF; ++u) fetch(st[u], dsc[u], ra + u);
#pragma unroll
	for (int u = 0; u < 3; ++u) {
		w0[u].A = w0[u].B = w0[u].C = w0[u].D = w0[u].E = cb_f2{0.0f, 0.0f};
		w1[u] = w0[u];
	}
	cb_u32 desc_m1 = 0, desc_m2 = 0;   // descriptors of the rows whose stage 1 ran one / two iterations ago
	const int last = y1 - 1 + 4;
	for (int g = ra; g <= last; g += RG) {
#pragma unroll
		for (int u = 0; u < RG; ++u) {
			const int r = g + u;
			if (r > last) break;
			cb_u4 &s = st[u % PF];
			// ---- commit V_k row r to ring slot u ----
			const float nv0 = __uint_as_float(s.x), nv1 = __uint_as_float(s.y), nv2 = __uint_as_float(s.z), nv3 = __uint_as_float(s.w);
			*(cb_f4 *)&V0[u][4 * lane] = cb_f4{nv0, nv1, nv2, nv3};
			const cb_u32 desc_a = dsc[u % PF];
			fetch(s, dsc[u % PF], r + PF);
			// ---- stage 1, row ya = r-2: V_k rows r-3, r-2, r-1 are w0[u%3], w0[(u+1)%3], w0[(u+2)%3] ----
			const int ya = r - 2;
			float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
			if (ya >= y0 - 2 && ya >= 0 && ya < H) {
				bm_mask n0 = masks[ya * F2_NMASK + 0], n1 = masks[ya * F2_NMASK + 1], n2 = masks[ya * F2_NMASK + 2], n3 = masks[ya * F2_NMASK + 3];
				skeleton(w0[u % 3], w0[(u + 1) % 3], w0[(u + 2) % 3], n0, n1, n2, n3, a0, a1, a2, a3);
				cb_u32 desc = desc_a;
				bool first = true;
				while ((n0 | n1 | n2 | n3) != 0) {
					const bm_mask act = n0 | n1 | n2 | n3;
					const int jsel = sel_i(n0, sel_i(n1, sel_i(n2, 3, 2), 1), 0);
					const int c = 4 * lane + jsel;
					const bool is_act = sel_i(act, 0, 1) != 0;
					if (!first) desc = __builtin_amdgcn_raw_buffer_load_b32(rd, is_act ? (cb_u32)((ya * W + xs0 + c) * 4) : OOB, 0, 0);
					first = false;
					float v = 0.0f;
					if (is_act && desc != 0) F2_WINDOW(V0, u, v);   // V_k rows ya-2 .. ya+2 = r-4 .. r: slots (u-4 .. u) mod RING
					if (__any(is_act && desc == 0)) {
						if (is_act && desc == 0) v = f2_general1(X, V0, ya, c, max(max(ra, 0), r - (RG - 1)), min(H - 1, r));
					}
					a0 = sel_f(n0, a0, v);
					a1 = sel_f(n1 & ~n0, a1, v);
					a2 = sel_f(n2 & ~(n0 | n1), a2, v);
					a3 = sel_f(n3 & ~(n0 | n1 | n2), a3, v);
					n3 = n3 & (n0 | n1 | n2);
					n2 = n2 & (n0 | n1);
					n1 = n1 & n0;
					n0 = 0;
				}
			}
			// V_(k+1) row ya is final: ring slot u (the slot of the iteration that produced it)
			*(cb_f4 *)&V1[u][4 * lane] = cb_f4{a0, a1, a2, a3};
			// ---- stage 2, row ye = r-4: V_(k+1) rows r-5, r-4, r-3 are w1[u%3], w1[(u+1)%3], w1[(u+2)%3] ----
			const int ye = r - 4;
			if (ye >= y0) {
				bm_mask n0 = masks[ye * F2_NMASK + 0] & OUT_LANES, n1 = masks[ye * F2_NMASK + 1] & OUT_LANES;
				bm_mask n2 = masks[ye * F2_NMASK + 2] & OUT_LANES, n3 = masks[ye * F2_NMASK + 3] & OUT_LANES;
				float res0, res1, res2, res3;
				skeleton(w1[u % 3], w1[(u + 1) % 3], w1[(u + 2) % 3], n0, n1, n2, n3, res0, res1, res2, res3);
				cb_u32 desc = desc_m2;
				bool first = true;
				while ((n0 | n1 | n2 | n3) != 0) {
					const bm_mask act = n0 | n1 | n2 | n3;
					const int jsel = sel_i(n0, sel_i(n1, sel_i(n2, 3, 2), 1), 0);
					const int c = 4 * lane + jsel;
					const bool is_act = sel_i(act, 0, 1) != 0;
					if (!first) desc = __builtin_amdgcn_raw_buffer_load_b32(rd, is_act ? (cb_u32)((ye * W + xs0 + c) * 4) : OOB, 0, 0);
					first = false;
					float v = 0.0f;
					// V_(k+1) rows ye-2 .. ye+2 = r-6 .. r-2 were produced by the iterations of input rows r-4 .. r: slots (u-4 .. u)
					if (is_act && desc != 0) F2_WINDOW(V1, u, v);
					if (__any(is_act && desc == 0)) {
						// final V_(k+1) rows resident in the ring: produced from input rows r-5 .. r, i.e. rows r-7 .. r-2, not
						// before the first one this wave computes (y0-2) nor outside the image
						if (is_act && desc == 0) v = f2_general2(X, V1, ye, c, max(max(y0 - 2, 0), r - 2 - (RG - 1)), min(H - 1, r - 2));
					}
					res0 = sel_f(n0, res0, v);
					res1 = sel_f(n1 & ~n0, res1, v);
					res2 = sel_f(n2 & ~(n0 | n1), res2, v);
					res3 = sel_f(n3 & ~(n0 | n1 | n2), res3, v);
					n3 = n3 & (n0 | n1 | n2);
					n2 = n2 & (n0 | n1);
					n1 = n1 & n0;
					n0 = 0;
				}
				const cb_u32 ob = (cb_u32)((ye * W + xs0) * 4 + lane16);
				if (interior) {
					if (has2) __builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res0), __float_as_uint(res1), __float_as_uint(res2), __float_as_uint(res3)}, ro, ob, 0, VOL_AUX);
				} else if (full_out) {
					__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res0), __float_as_uint(res1), __float_as_uint(res2), __float_as_uint(res3)}, ro, ob, 0, VOL_AUX);
				} else if (any_out) {
					const float res[4] = {res0, res1, res2, res3};
#pragma unroll
					for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[j]), ro, xs + j < W ? ob + 4u * j : OOB, 0, 0);
				}
			}
			// ---- rotate: V_k row r and V_(k+1) row r-2 into the register windows (slots of rows r-3 / r-5) ----
			make_row(w0[u % 3], nv0, nv1, nv2, nv3);
			make_row(w1[u % 3], a0, a1, a2, a3);
			desc_m2 = desc_m1;
			desc_m1 = desc_a;
		}
	}
#undef F2_WINDOW
}

// ---- host side ---------------------------------------------------------------------------------------------------
static void f2_geometry(CbcaArgs &A, const CbcaCfg &cfg, int D, int H, int W)
{
	const int nd = cfg.nd > 0 ? cfg.nd : D;
	A.d0 = cfg.nd > 0 ? cfg.d0 : 0;
	A.nd = nd;
	A.gx = (int)cdiv(W, F2_STEP);
	// output rows per strip: 10 halo rows per chunk (6 above, 4 below); 64 unless that leaves fewer than ~16 K waves
	const int64_t gy_min = cdiv((int64_t)16384, (int64_t)A.gx * nd);
	const int rb_auto = (int)std::min<int64_t>(64, std::max<int64_t>(24, cdiv((int64_t)H, gy_min)));
	A.rb = cfg.rb > 0 ? cfg.rb : rb_auto;
	A.gy = (int)cdiv(H, A.rb);
}

static size_t f2_mask_bytes(int D, int H, int W) { return ((size_t)D * cdiv(W, F2_STEP) * H * F2_NMASK * sizeof(bm_mask) + 255) & ~(size_t)255; }
// bytes of the classification of one (pair, direction): counters + lane masks + descriptors
size_t cbca_class_bytes(int D, int H, int W) { return F2_HDR + f2_mask_bytes(D, H, W) + (((size_t)D * H * W * sizeof(cb_u32) + 255) & ~(size_t)255); }

static void f2_fill(F2Args &B, const void *packed, const void *cls, int D, int H, int W, int direction)
{
	CbcaArgs &A = B.c;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = nullptr; A.vout = nullptr;
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.overflow = nullptr;
	B.counters = (unsigned long long *)cls;
	B.masks = (bm_mask *)((char *)cls + F2_HDR);
	B.desc = (cb_u32 *)((char *)cls + F2_HDR + f2_mask_bytes(D, H, W));
	// the gate: two plain iterations are faster when more than ~5 % of the outputs are flagged (the per-row re-evaluation
	// passes dominate) and the nested rebuild must stay rare
	const double total = (double)D * H * W;
	B.max_flagged = (unsigned long long)(total * 0.05);
	B.max_unfit = (unsigned long long)(total * 2e-4);
}

// the gate as the plain strip kernel sees it: run only when the fused kernel stood down
void cbca_gate_args(const void *cls, int D, int H, int W, const unsigned long long **counters, unsigned long long *max_flagged,
                    unsigned long long *max_unfit)
{
	F2Args B;
	f2_fill(B, cls /* unused */, cls, D, H, W, 1);
	*counters = (const unsigned long long *)cls;
	*max_flagged = B.max_flagged;
	*max_unfit = B.max_unfit;
}

// classification of every output of a (pair, direction): once, before the iterations (arms <= 254 required)
int cbca_classify(const void *packed, void *cls, int D, int H, int W, int direction, hipStream_t st)
{
	F2Args B;
	f2_fill(B, packed, cls, D, H, W, direction);
	CbcaCfg cfg;
	cfg.rb = 64;
	f2_geometry(B.c, cfg, D, H, W);
	const hipError_t e = hipMemsetAsync(cls, 0, F2_HDR, st);
	if (e != hipSuccess) {
		set_error("cbca_classify: %s", hipGetErrorString(e));
		return (int)e;
	}
	const int64_t waves = (int64_t)cdiv((int64_t)B.c.gx * B.c.gy, 8) * 8 * cdiv(B.c.nd, 4) * 4;
	hipLaunchKernelGGL(cbca_classify_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, B);
	return check_launch("cbca_classify");
}

// Two cbca iterations in one pass: vin = V_k, vout = V_(k+2).  force: ignore the density gate (tests).
int cbca_fused2(const void *packed, const void *cls, const float *vin, float *vout, int D, int H, int W, int direction,
                hipStream_t st, const CbcaCfg &cfg, bool force)
{
	F2Args B;
	f2_fill(B, packed, cls, D, H, W, direction);
	if (force) B.max_flagged = B.max_unfit = ~0ull;
	B.c.vin = vin; B.c.vout = vout;
	f2_geometry(B.c, cfg, D, H, W);
	const int64_t waves = (int64_t)cdiv((int64_t)B.c.gx * B.c.gy, 8) * 8 * cdiv(B.c.nd, 4) * 4;
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)B.c.nd * H * W * 4 > ((int64_t)768 << 20);
	if (nt) hipLaunchKernelGGL((cbca_fused2_kernel<true>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, B);
	else hipLaunchKernelGGL((cbca_fused2_kernel<false>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, B);
	return check_launch("cbca_fused2");
}

}  // namespace mc
