// Cross-based cost aggregation (adcensus.cu:343-377), one iteration per launch, with the re-evaluation of non-minimal
// supports DEFERRED and batched (gfx950).
//
// Findings this design answers (measured at 1000x1500x256): the data movement of a strip walk takes
// 0.72-0.86 ms per iteration once a wave keeps 6 rows of reads in flight; the minimal-3x3 arithmetic out of registers is
// free next to it; what made every earlier kernel cost ~1.0 ms is the pass that re-evaluates the FEW outputs with a larger
// support -- ~3 of the 248 outputs of a row on textured images, but present on 96 % of the rows, and a wave64 instruction
// costs its 4 cycles whether 3 or 64 lanes work.  Here that pass runs once per ~8 rows on a full wave's worth of entries:
//   * cbca_dq_classify_kernel (once per pair and direction): per (plane, strip, row) four 64-bit lane masks -- the output in
//     frame column 4*lane + j has a support other than the minimal 3x3 (and exists and has its partner inside the image)
//     -- and per flagged output a 32-bit descriptor in a (D,H,W) array: bit 5*k + t = tap (row y-2+k, column x-2+t)
//     belongs to the support, bits 25..29 = tap count, 0 = the support does not fit that window;
//   * per row the iteration kernel commits the row to a 12-row LDS ring, sums the minimal supports out of registers (column
//     pairs, v_pk_add_f32, rows ascending / x ascending as the reference, division by 9 as three packed operations inside
//     the range the form is proven for, IEEE divide outside), stores the row, and appends the row's flagged outputs
//     (scalar masks -> mbcnt ranks) to a wave-private queue in LDS;
//   * every 8 rows (or when the queue would overflow: dense images) lane e takes queue entry e: descriptor from memory, the 5 x 5
//     window from the ring, each tap kept or replaced by -0.0f according to its descriptor bit, added in the reference's
//     order from +0.0, IEEE divide by the tap count, one dword store over the row's provisional value (same wave, later
//     in program order); descriptor 0: the reference's loop over ring rows / global memory.
// No arm lengths, no flags and no block barrier in the iteration kernel; results bit-identical to adcensus.cbca.
#include "cbca_common.h"
#include <algorithm>

namespace mc {

constexpr int DQ_STEP = 248;   // output columns per strip (frame columns 4 .. 251)
constexpr int DQ_HALO = 4;
constexpr int DQ_RING = 12;    // rows per LDS ring (2 x the prefetch depth: the slot of a row is a compile-time offset + 0 or 6)
constexpr int DQ_NMASK = 4;    // lane masks per (plane, strip, row)
constexpr int DQ_QCAP = 512;   // queue entries per wave (16 bit each)
constexpr int DQ_DENSE = 64;   // a row with more flagged outputs than this is re-evaluated in place, lanes = its own columns

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



struct DqArgs {
	CbcaArgs c;
	bm_mask *masks;    // [plane][strip][row][DQ_NMASK]
	cb_u32 *desc;      // (D,H,W) support descriptors of the flagged outputs
};

__device__ __forceinline__ void dq_wave_coords(const CbcaArgs &A, int &d, int &cx, int &cy, bool &live)
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

// ---- classification, once per pair and direction -----------------------------------------------------------------
__global__ void __launch_bounds__(256) cbca_dq_classify_kernel(const DqArgs B)
{
	const CbcaArgs &A = B.c;
	const int lane = threadIdx.x & 63;
	int d, cx, cy;
	bool live;
	dq_wave_coords(A, d, cx, cy, live);
	if (!live) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs0 = cx * DQ_STEP - DQ_HALO;
	const int xs = xs0 + 4 * lane;
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const cb_u32 OOB = 0x80000000u;
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(B.desc + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const bool has_out = lane >= 1 && lane <= 62;
	bool ok[4];   // output exists and its shifted partner is inside the image (adcensus.cu:353-354)
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		ok[j] = has_out && x < W && x + sh >= 0 && x + sh < W;
	}
	bm_mask *__restrict__ out = B.masks + ((size_t)(d * A.gx + cx) * H) * DQ_NMASK;
	cb_u32 m[5][4];   // byte-minimum arm lengths of rows r-4 .. r (m[4] = newest)
#pragma unroll
	for (int k = 0; k < 5; ++k)
#pragma unroll
		for (int j = 0; j < 4; ++j) m[k][j] = 0;
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
		}
		if (lane < DQ_NMASK) {
			bm_mask v = need_m[0];
			v = lane == 1 ? need_m[1] : v; v = lane == 2 ? need_m[2] : v; v = lane == 3 ? need_m[3] : v;
			out[(size_t)yo * DQ_NMASK + lane] = v;
		}
	}
}

struct DqLds {
	float V[4][DQ_RING][CS_COLS];
	unsigned short Q[4][DQ_QCAP];
};

template <bool NT>
__global__ void __launch_bounds__(256) cbca_dq_kernel(const DqArgs B)
{
	const CbcaArgs &A = B.c;
	constexpr int VOL_AUX = NT ? 2 : 0;
	constexpr int PF = 6;                      // rows in flight per wave = rows per group; DQ_RING = 2 * PF
	static_assert(DQ_RING == 2 * PF, "ring slots are derived from the position inside a group of PF rows");
	static_assert(DQ_QCAP >= PF * DQ_DENSE, "a group's sparse rows must fit the queue");
	__shared__ DqLds S;
	const int lane = threadIdx.x & 63;
	int d, cx, cy;
	bool live;
	dq_wave_coords(A, d, cx, cy, live);
	if (!live) return;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float (*__restrict__ V)[CS_COLS] = S.V[wv];
	unsigned short *__restrict__ Q = S.Q[wv];
	const int H = A.H, W = A.W, direction = A.direction;
	const int HWi = H * W;
	const int sh = d * direction;
	const int xs0 = cx * DQ_STEP - DQ_HALO;             // image column of frame column 0 (wave-uniform)
	const int xs = xs0 + 4 * lane;                      // image column of this lane's first column
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const int ra = y0 - 2;                              // first staged row: the window of output row y0 starts two rows above it
	const int plane_bytes = HWi * 4;
	const cb_u32 OOB = 0x80000000u;
	const float *__restrict__ plane_in = A.vin + (size_t)d * HWi;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)plane_in, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)(B.desc + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	typedef const __attribute__((address_space(4))) bm_mask *mask_ptr;
	const mask_ptr masks = (mask_ptr)(B.masks + ((size_t)(d * A.gx + cx) * H) * DQ_NMASK);
	const bool interior = xs0 >= 0 && xs0 + CS_COLS <= W;   // wave-uniform
	const bool full_in = xs >= 0 && xs + 3 < W;
	const bool has_out = lane >= 1 && lane <= 62;
	const bool full_out = has_out && xs + 3 < W;
	const bool any_out = has_out && xs < W;
	bool copy_any = false;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		copy_any = copy_any || (has_out && x < W && !(x + sh >= 0 && x + sh < W));
	}
	const bool any_copy = __any(copy_any);   // wave-uniform: this strip holds outputs that are copied through
	const int lane16 = lane * 16;

	auto fetch = [&](cb_u4 &v, int r) {  // row r of the plane -> registers (rows outside the image: zeros)
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

	cb_u4 st[PF];
	C2Row w[3];
#pragma unroll
	for (int u = 0; u < PF; ++u) fetch(st[u], ra + u);
#pragma unroll
	for (int u = 0; u < 3; ++u) w[u].A = w[u].B = w[u].C = w[u].D = w[u].E = cb_f2{0.0f, 0.0f};
	const int last = y1 + 1;   // the windows of the last output row reach two rows below it
	int gslot = 0;             // ring slot of row g (0 or PF)
	for (int g = ra; g <= last; g += PF, gslot ^= PF) {
		int qn = 0;            // queued outputs of this group's rows g-2 .. g+3 (entry = position in the group << 8 | frame column)
		cb_u32 dense = 0;      // bit u: row g-2+u has more than DQ_DENSE flagged outputs and is re-evaluated in place
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const int r = g + u;
			if (r > last) break;
			cb_u4 &s = st[u];
			// ---- commit row r to the ring ----
			const float nv0 = __uint_as_float(s.x), nv1 = __uint_as_float(s.y), nv2 = __uint_as_float(s.z), nv3 = __uint_as_float(s.w);
			*(cb_f4 *)&V[gslot + u][4 * lane] = cb_f4{nv0, nv1, nv2, nv3};
			fetch(s, r + PF);
			// row r into the register window (it takes the slot of row r-3)
			{
				const float l3 = lane_from_below(nv3, 0.0f), r0 = lane_from_above(nv0, 0.0f);
				C2Row &nw = w[u % 3];
				nw.A = cb_f2{l3, nv0}; nw.B = cb_f2{nv0, nv1}; nw.C = cb_f2{nv1, nv2}; nw.D = cb_f2{nv2, nv3}; nw.E = cb_f2{nv3, r0};
			}
			// ---- output row yb = r-1: rows r-2, r-1, r are w[(u+1)%3], w[(u+2)%3], w[u%3] ----
			const int yb = r - 1;
			if (yb >= y0 && yb < y1) {
				const bm_mask n0 = masks[yb * DQ_NMASK + 0], n1 = masks[yb * DQ_NMASK + 1], n2 = masks[yb * DQ_NMASK + 2], n3 = masks[yb * DQ_NMASK + 3];
				const C2Row &up = w[(u + 1) % 3], &own = w[(u + 2) % 3], &dn_ = w[u % 3];
				cb_f2 s01 = cb_f2{0.0f, 0.0f}, s23 = cb_f2{0.0f, 0.0f};
				s01 += up.A; s01 += up.B; s01 += up.C;
				s23 += up.C; s23 += up.D; s23 += up.E;
				s01 += own.A; s01 += own.B; s01 += own.C;
				s23 += own.C; s23 += own.D; s23 += own.E;
				s01 += dn_.A; s01 += dn_.B; s01 += dn_.C;
				s23 += dn_.C; s23 += dn_.D; s23 += dn_.E;
				const cb_f2 q01 = div9_pk(s01), q23 = div9_pk(s23);
				float res0 = q01.x, res1 = q01.y, res2 = q23.x, res3 = q23.y;
				// a sum outside [2^-95, 2^125) (zero, tiny, huge, inf, nan) in an unflagged column: IEEE divide for the row
				// (columns that do not exist or are copied through count too: conservative, never wrong)
				const bm_mask odd = (div9_bad(s01.x) & ~n0) | (div9_bad(s01.y) & ~n1) | (div9_bad(s23.x) & ~n2) | (div9_bad(s23.y) & ~n3);
				if (odd != 0) {
					res0 = s01.x / 9.0f; res1 = s01.y / 9.0f; res2 = s23.x / 9.0f; res3 = s23.y / 9.0f;
				}
				if (any_copy) {  // adcensus.cu:353-354: outputs whose partner lies outside the image are copied through
					const int xp = xs + sh;
					res0 = (xp + 0 >= 0 && xp + 0 < W) ? res0 : own.B.x; res1 = (xp + 1 >= 0 && xp + 1 < W) ? res1 : own.B.y;
					res2 = (xp + 2 >= 0 && xp + 2 < W) ? res2 : own.D.x; res3 = (xp + 3 >= 0 && xp + 3 < W) ? res3 : own.D.y;
				}
				// the row leaves now; its flagged columns carry provisional values until the group is drained
				const cb_u32 ob = (cb_u32)((yb * W + xs0) * 4 + lane16);
				if (interior) {
					if (has_out) __builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res0), __float_as_uint(res1), __float_as_uint(res2), __float_as_uint(res3)}, ro, ob, 0, VOL_AUX);
				} else if (full_out) {
					__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res0), __float_as_uint(res1), __float_as_uint(res2), __float_as_uint(res3)}, ro, ob, 0, VOL_AUX);
				} else if (any_out) {
					const float res[4] = {res0, res1, res2, res3};
#pragma unroll
					for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[j]), ro, xs + j < W ? ob + 4u * j : OOB, 0, VOL_AUX);
				}
			}
			// ---- queue the flagged outputs of row yq = r-2: its 5 x 5 windows are complete now ----
			const int yq = r - 2;
			if (yq >= y0 && yq < y1) {
				const bm_mask nm[4] = {masks[yq * DQ_NMASK + 0], masks[yq * DQ_NMASK + 1], masks[yq * DQ_NMASK + 2], masks[yq * DQ_NMASK + 3]};
				if ((nm[0] | nm[1] | nm[2] | nm[3]) != 0) {
					const int nnew = __builtin_popcountll(nm[0]) + __builtin_popcountll(nm[1]) + __builtin_popcountll(nm[2]) + __builtin_popcountll(nm[3]);
					if (nnew > DQ_DENSE) {
						dense |= 1u << u;
					} else {
						int base = qn;
#pragma unroll
						for (int j = 0; j < 4; ++j) {
							if (nm[j] != 0) {
								const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((cb_u32)(nm[j] >> 32), __builtin_amdgcn_mbcnt_lo((cb_u32)nm[j], 0));
								if (sel_i(nm[j], 0, 1)) Q[pos] = (unsigned short)((u << 8) | (4 * lane + j));
								base += __builtin_popcountll(nm[j]);
							}
						}
						qn = base;
					}
				}
			}
		}
		// ---- re-evaluate the flagged outputs of rows g-2 .. g+3 (one code site per group) ----
		// work items: the queue in chunks of 64 entries, then the dense rows, one pass per column j with the lanes of
		// the row's own mask
		if (qn == 0 && dense == 0) continue;
		const int nchunks = (qn + 63) >> 6;
		const int rnew = min(g + PF - 1, last);     // newest committed row
		int it = 0, dj = 0;
		cb_u32 dm = dense;
		while (true) {
			bool active;
			int uu, c;
			if (it < nchunks) {
				const int e = it * 64 + lane;
				active = e < qn;
				const int ent = active ? (int)Q[e] : 0;
				uu = ent >> 8; c = ent & 255;
				++it;
			} else if (dm != 0) {
				const int u0 = __builtin_ctz(dm);
				const bm_mask m = masks[(g - 2 + u0) * DQ_NMASK + dj];
				active = sel_i(m, 0, 1) != 0;
				uu = u0; c = 4 * lane + dj;
				if (++dj == 4) { dj = 0; dm &= dm - 1; }
			} else {
				break;
			}
			const int row = g - 2 + uu;
			const cb_u32 go = (cb_u32)((row * W + xs0 + c) * 4);
			const cb_u32 desc = __builtin_amdgcn_raw_buffer_load_b32(rd, active ? go : OOB, 0, 0);
			float v = 0.0f;
			if (active && desc != 0) {
				// slot of row - 2: rows g .. g+5 sit in slots gslot .. gslot+5, the six rows before them in the other half
				int sl = gslot + uu + (DQ_RING - 4);
				sl = sl >= DQ_RING ? sl - DQ_RING : sl;
				float sum;
				{
					const cb_u32 keep0 = (cb_u32)(((int)(desc << 31)) >> 31);
					const cb_u32 tb0 = __float_as_uint(V[sl][c - 2]);
					const float m0 = __uint_as_float((tb0 & keep0) | (0x80000000u & ~keep0));
					asm("v_add_f32 %0, 0, %1" : "=v"(sum) : "v"(m0));   // the reference's accumulator starts at +0.0
				}
#pragma unroll
				for (int k = 0; k < 5; ++k) {
					const float *rowp = &V[sl][c - 2];
#pragma unroll
					for (int t = 0; t < 5; ++t) {
						if (k == 0 && t == 0) continue;
						const cb_u32 kp = (cb_u32)(((int)(desc << (31 - (5 * k + t)))) >> 31);   // all ones if the tap is in
						const cb_u32 tb = __float_as_uint(rowp[t]);
						sum += __uint_as_float((tb & kp) | (0x80000000u & ~kp));
					}
					sl = sl + 1 == DQ_RING ? 0 : sl + 1;
				}
				v = sum / (float)(desc >> 25);
			}
			if (__any(active && desc == 0)) {   // descriptor 0: the support does not fit the window -> the reference's loop
				if (active && desc == 0) {
					const int lo_row = max(max(ra, 0), rnew - (DQ_RING - 1)), hi_row = min(H - 1, rnew);
					const int x = xs0 + c;
					const int g0 = row * W + x;
					const cb_u32 own = bytemin4(A.p0[g0], A.p1[g0 + sh]);
					const int ua = (int)((own >> 16) & 0xff), da = (int)(own >> 24);
					float sum = 0;
					int cnt = 0;
					for (int q = row - ua; q <= row + da; ++q) {
						const int gq = q * W + x;
						const cb_u32 mm = bytemin4(A.p0[gq], A.p1[gq + sh]);
						const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
						const int n = l + rg + 1;
						if (q >= lo_row && q <= hi_row && c - l >= 0 && c + rg < CS_COLS) {
							const int t = q - ra;
							const float *rp = &V[t - DQ_RING * ((t * 43691) >> 19)][c - l];   // t % 12
							for (int k = 0; k < n; ++k) sum += rp[k];
						} else {
							const float *rp = plane_in + gq - l;
							for (int k = 0; k < n; ++k) sum += rp[k];
						}
						cnt += n;
					}
					v = sum / (float)cnt;
				}
			}
			if (active) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, go, 0, VOL_AUX);
		}
	}
}

static void dq_geometry(CbcaArgs &A, const CbcaCfg &cfg, int D, int H, int W)
{
	const int nd = cfg.nd > 0 ? cfg.nd : D;
	A.d0 = cfg.nd > 0 ? cfg.d0 : 0;
	A.nd = nd;
	A.gx = (int)cdiv(W, DQ_STEP);
	// output rows per strip: 4 halo rows per chunk; 40 unless that leaves fewer than ~16 K waves
	const int64_t gy_min = cdiv((int64_t)16384, (int64_t)A.gx * nd);
	const int rb_auto = (int)std::min<int64_t>(40, std::max<int64_t>(16, cdiv((int64_t)H, gy_min)));
	A.rb = cfg.rb > 0 ? cfg.rb : rb_auto;
	A.gy = (int)cdiv(H, A.rb);
}

static size_t dq_mask_bytes(int D, int H, int W) { return ((size_t)D * cdiv(W, DQ_STEP) * H * DQ_NMASK * sizeof(bm_mask) + 255) & ~(size_t)255; }
// bytes of the classification of one (pair, direction): lane masks + descriptors
size_t cbca_class_bytes(int D, int H, int W) { return dq_mask_bytes(D, H, W) + (((size_t)D * H * W * sizeof(cb_u32) + 255) & ~(size_t)255); }

static void dq_fill(DqArgs &B, const void *packed, const void *cls, int D, int H, int W, int direction)
{
	CbcaArgs &A = B.c;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = nullptr; A.vout = nullptr;
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.overflow = nullptr;
	B.masks = (bm_mask *)cls;
	B.desc = (cb_u32 *)((char *)cls + dq_mask_bytes(D, H, W));
}

// classification of every output of a (pair, direction): once, before the iterations (arms <= 254 required)
int cbca_classify(const void *packed, void *cls, int D, int H, int W, int direction, hipStream_t st)
{
	DqArgs B;
	dq_fill(B, packed, cls, D, H, W, direction);
	CbcaCfg cfg;
	cfg.rb = 64;
	dq_geometry(B.c, cfg, D, H, W);
	const int64_t waves = (int64_t)cdiv((int64_t)B.c.gx * B.c.gy, 8) * 8 * cdiv(B.c.nd, 4) * 4;
	hipLaunchKernelGGL(cbca_dq_classify_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, B);
	return check_launch("cbca_classify");
}

// one cbca iteration driven by the classification: vin -> vout
int cbca_dq(const void *packed, const void *cls, const float *vin, float *vout, int D, int H, int W, int direction,
            hipStream_t st, const CbcaCfg &cfg)
{
	DqArgs B;
	dq_fill(B, packed, cls, D, H, W, direction);
	B.c.vin = vin; B.c.vout = vout;
	dq_geometry(B.c, cfg, D, H, W);
	const int64_t waves = (int64_t)cdiv((int64_t)B.c.gx * B.c.gy, 8) * 8 * cdiv(B.c.nd, 4) * 4;
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)B.c.nd * H * W * 4 > ((int64_t)768 << 20);
	if (nt) hipLaunchKernelGGL((cbca_dq_kernel<true>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, B);
	else hipLaunchKernelGGL((cbca_dq_kernel<false>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, B);
	return check_launch("cbca_dq");
}

}  // namespace mc
