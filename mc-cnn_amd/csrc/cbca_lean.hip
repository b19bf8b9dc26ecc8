// Cross-based cost aggregation (adcensus.cu:343-377) on pairs whose supports are nearly all the minimal 3 x 3 -- the regime
// SURVEY 8(d) specifies for the 1000 x 1500 x 256 case (Gaussian texture: 98.7 % of the outputs) -- as a bandwidth problem.
//
// Whether an output's support is the minimal 3 x 3 depends on the pair's arms and the plane only, not on the volume, and
// mc_predict aggregates a pair 2 + 16 times per direction (main.lua:998-1001, 1033-1039).  So the work is split once per
// pair and direction:
//   cbca_classify_kernel  (arms only)  LISTS every output with a partner whose support is NOT the minimal 3 x 3, with the
//                                      support's shape (16 bytes per entry), per wave of the lean kernel, in the pair's plan area;
//   cbca_lean_kernel      (per pass)   computes the minimal 3 x 3 mean for EVERY output -- nine additions in the reference's
//                                      order and the division by 9, out of registers, neighbours' columns through DPP: no arm
//                                      lengths, no LDS, no tests; the plane is read once and written once -- and re-runs the
//                                      reference's loop for the wave's listed outputs (one lane per entry, values through L1 /
//                                      L2: the wave and its neighbours are reading those rows) and overwrites them.
// Outputs without a partner are copied through by the lean kernel (adcensus.cu:353-354).  The strip kernel (cbca.hip), which
// does all of this per pass, remains what adcensus.cbca runs on its own (no state between calls) and the fallback when the
// list does not fit.
//
// Second half of the file: cbca_lean2x_kernel runs TWO consecutive passes per launch (the first one's rows stay in LDS) out of
// per-wave RECORDS that cbca_classify2x_kernel writes -- what mc_predict runs on such pairs; the single-pass kernels above are
// what the passes were in the first half of round 4 and stay behind the test hook (mc_cbca_ws_cfg forms 8 / 9).
#include "cbca_common.h"
#include <algorithm>

namespace mc {

namespace {

struct LeanArgs {
	const uint32_t *p0, *p1;     // packed arm lengths (H,W)
	const float *vin;
	float *vout;
	uint32_t *hdr;               // list header (LH_*); behind it the wave table and the slots
	uint32_t capd;               // slots per plane (the slots are allotted per plane)
	uint32_t wtab_words;         // word offset of the wave table: per wave of the lean kernel {slot + 1 of its first entry, its entries}
	float cost_limit;            // two-pass geometry: values recomputed per voxel, on average over a plane, beyond which the list is declared unusable
	uint32_t slots_words;        // word offset of the slots (16 bytes each)
	uint32_t cap;                // slots the list can hold
	int D, H, W, direction;
	int rb, gx, gy;              // rows per wave, strips per row, row chunks
	cb_u32 gx_rcp;               // ceil(2^32 / gx)
	int order, gyb;              // wave order: 0 linear over the volume, 1 one band of gyb row chunks per XCD (blockIdx & 7), each swept linearly
	int pitch, xoff;             // a wave's 256 columns start at strip * pitch + xoff (single pass: 256, 0; two passes in one launch: 252, -2)
	int wpb;                     // waves per block of the launch (4; cbca_lean2x_kernel: L2X_WPB)
	cb_u32 rbcode;               // header word LH_RB of a list of this geometry: rb, + 0x100 for the two-pass geometry
	const uint32_t *flags;       // cbca_pack's flag words
	int route;
};

// does this launch run?  cbca_gate (the pair's route) and list_valid (the list is this problem's, and complete) with ONE wait: the
// eight flag words and the eight header words are requested together (the short-lived waves of cbca_lean_kernel cannot afford a
// chain of dependent scalar loads before their rows are requested)
__device__ __forceinline__ bool lean_runs(const LeanArgs &A)
{
	static_assert(CS_FLAGS == 8 && LH_RB < 8, "two 32-byte scalar loads");
	const cb_u4 f0 = *(const cb_u4 *)A.flags, f1 = *(const cb_u4 *)(A.flags + 4);
	const cb_u4 h0 = *(const cb_u4 *)A.hdr, h1 = *(const cb_u4 *)(A.hdr + 4);
	(void)f1;
	const cb_u32 fl[4] = {f0.x, f0.y, f0.z, f0.w};   // CF_SATURATED, CF_ARM_GT4, CF_ARM_GT13, CF_ROUTE
	bool gate;
	switch (A.route) {
	case CR_ARMS_LE4: gate = !fl[CF_ARM_GT4]; break;
	case CR_ARMS_LE13: gate = !fl[CF_ARM_GT13]; break;
	case CR_NOT_DIRECT: gate = fl[CF_ROUTE] != CR_DIRECT; break;
	case CR_STRIP_OR_TILE13: gate = fl[CF_ROUTE] == CR_STRIP || fl[CF_ROUTE] == CR_TILE13; break;
	default: gate = fl[CF_ROUTE] == (cb_u32)A.route; break;
	}
	// (LH_COUNT, LH_OVERFLOW, LH_D, LH_H | LH_W, LH_DIR, LH_MAGIC, LH_RB)
	const bool valid = (h1.z == LH_MAGIC_VALUE) & (h0.z == (cb_u32)A.D) & (h0.w == (cb_u32)A.H) & (h1.x == (cb_u32)A.W) &
	                   (h1.y == (cb_u32)(A.direction + 1)) & (h1.w == A.rbcode) & (h0.y == 0u);
	return gate & valid;
}

// wave -> (plane, row chunk, strip of 256 columns); strips fastest, so that the waves of a block are neighbours in a row
// (a strip's edge columns are its neighbours' lines: L1 / L2 hits).  w = the wave's number in the canonical order
// (plane, chunk, strip): its word of the wave table.  order 1: the blocks of an XCD (blockIdx & 7: observed placement, used for
// speed only) sweep ONE band of gyb row chunks of every plane, top to bottom, plane after plane -- so that each L2 streams a
// contiguous range and holds the rows that vertically adjacent waves share.
// (grid: x = the blocks of one plane, y = the plane -- planes are dispatched one after the other, no 64-bit division per wave)
__device__ __forceinline__ bool lean_wave(const LeanArgs &A, int wv, long long &w, int &d, int &y0, int &y1, int &x0)
{
	d = (int)blockIdx.y;
	const cb_u32 lw = (A.order == 1 ? (blockIdx.x >> 3) : blockIdx.x) * (cb_u32)A.wpb + (cb_u32)wv;
	const cb_u32 t = A.gx == 1 ? lw : __umulhi(lw, A.gx_rcp);   // lw / gx (exact for lw < 2^16: gx_rcp = ceil(2^32 / gx), gx >= 2)
	const int strip = (int)(lw - t * (cb_u32)A.gx);
	int chunk = (int)t;
	if (A.order == 1) {
		if (chunk >= A.gyb) return false;
		chunk += (int)(blockIdx.x & 7) * A.gyb;
	}
	if (chunk >= A.gy) return false;
	w = ((long long)d * A.gy + chunk) * A.gx + strip;
	y0 = chunk * A.rb;
	y1 = min(A.H, y0 + A.rb);
	x0 = strip * A.pitch + A.xoff;
	return true;
}

// byte j (0 .. 11) of the entry's words 1 .. 3
__device__ __forceinline__ cb_u32 entry_byte(const cb_u4 &e, int j)
{
	const cb_u32 wsel = j < 4 ? e.y : (j < 8 ? e.z : e.w);
	return (wsel >> ((j & 3) * 8)) & 0xffu;
}

// The wave's listed outputs: the reference's loop (rows ascending, x ascending, one accumulator, adcensus.cu:356-373), one lane
// per entry.  An entry holds the voxel index and its support's shape, so that no arm length is looked up in a pass:
//   byte 0 = 0xe0 | up | down << 2      at most four rows of at most four values each (99.8 % of a texture's entries): bytes 1 .. 4 =
//                                       left | right << 4 per row; the four runs are requested at once, one 16-byte load each
//   byte 0 = up | down << 4 (< 0xe0)    at most 11 rows, arms up to 13 / 15: bytes 1 .. 11 = left | right << 4 per row
//   byte 0 = 0xff                       anything else: the arm lengths are looked up
// Runs are read four values at a time (any 4-byte alignment); the values behind a run's end add -0.0f (x + -0.0f == x).
// the small class in two halves, so that cbca_lean_kernel can request an entry's runs before and sum them after its own rows' work:
// `on` = this lane has a small-class entry (else nothing is requested and the sum is void)
__device__ __forceinline__ void small_request(const LeanArgs &A, const __amdgpu_buffer_rsrc_t &rv, bool on, const cb_u4 &e, cb_u32 rem, cb_u4 (&v)[4], int (&nn)[4])
{
	const cb_u32 b0 = e.y & 0xffu;
	const int u = (int)(b0 & 3u), rows = on ? u + (int)((b0 >> 2) & 3u) + 1 : 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const cb_u32 lr = entry_byte(e, 1 + k);
		nn[k] = k < rows ? (int)(lr & 15u) + (int)(lr >> 4) + 1 : 0;
		v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, k < rows ? (cb_u32)((int)rem + (k - u) * A.W - (int)(lr & 15u)) * 4u : 0x80000000u, 0, 0);
	}
}
__device__ __forceinline__ float small_sum(const cb_u4 (&v)[4], const int (&nn)[4])
{
	float sum = 0;
	int cnt = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		sum += 0 < nn[k] ? __uint_as_float(v[k].x) : -0.0f;
		sum += 1 < nn[k] ? __uint_as_float(v[k].y) : -0.0f;
		sum += 2 < nn[k] ? __uint_as_float(v[k].z) : -0.0f;
		sum += 3 < nn[k] ? __uint_as_float(v[k].w) : -0.0f;
		cnt += nn[k];
	}
	return sum / (float)cnt;
}

// value of one listed output (entry e of plane d; rem = y * W + x): the reference's loop out of the entry's shape
__device__ __forceinline__ float list_entry_value(const LeanArgs &A, const __amdgpu_buffer_rsrc_t &rv, int d, const cb_u4 &e, cb_u32 rem)
{
	const int W = A.W;
	const int sh = d * A.direction;
	const cb_u32 OOB = 0x80000000u;
	const cb_u32 b0 = e.y & 0xffu;
	float sum = 0;
	int cnt = 0;
	if ((b0 & 0xf0u) == 0xe0u) {
		const int u = (int)(b0 & 3u), rows = u + (int)((b0 >> 2) & 3u) + 1;
		cb_u4 v[4];
		int nn[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const cb_u32 lr = entry_byte(e, 1 + k);
			nn[k] = k < rows ? (int)(lr & 15u) + (int)(lr >> 4) + 1 : 0;
			v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, k < rows ? (cb_u32)((int)rem + (k - u) * W - (int)(lr & 15u)) * 4u : OOB, 0, 0);
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			sum += 0 < nn[k] ? __uint_as_float(v[k].x) : -0.0f;
			sum += 1 < nn[k] ? __uint_as_float(v[k].y) : -0.0f;
			sum += 2 < nn[k] ? __uint_as_float(v[k].z) : -0.0f;
			sum += 3 < nn[k] ? __uint_as_float(v[k].w) : -0.0f;
			cnt += nn[k];
		}
	} else if (b0 != 0xffu) {
		const int u = (int)(b0 & 15u), rows = u + (int)(b0 >> 4) + 1;
		for (int k = 0; k < rows; ++k) {
			const cb_u32 lr = entry_byte(e, 1 + k);
			const int l = (int)(lr & 15u), nk = l + (int)(lr >> 4) + 1;
			const int start = (int)rem + (k - u) * W - l;
			for (int c = 0; c < nk; c += 4) {
				const cb_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rv, (cb_u32)(start + c) * 4u, 0, 0);
				sum += __uint_as_float(v.x);
				sum += c + 1 < nk ? __uint_as_float(v.y) : -0.0f;
				sum += c + 2 < nk ? __uint_as_float(v.z) : -0.0f;
				sum += c + 3 < nk ? __uint_as_float(v.w) : -0.0f;
			}
			cnt += nk;
		}
	} else {
		const uint32_t mm = bytemin4(A.p0[rem], A.p1[(int)rem + sh]);
		const int u = (int)((mm >> 16) & 0xffu), dn = (int)(mm >> 24);
		for (int q = -u; q <= dn; ++q) {
			const int g = (int)rem + q * W;
			const uint32_t m = bytemin4(A.p0[g], A.p1[g + sh]);
			const int l = (int)(m & 0xffu), nk = l + (int)((m >> 8) & 0xffu) + 1;
			for (int k = 0; k < nk; ++k) sum += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, (cb_u32)(g - l + k) * 4u, 0, 0));
			cnt += nk;
		}
	}
	return sum / (float)cnt;
}

__device__ __forceinline__ void list_phase(const LeanArgs &A, long long w, int d, int lane)
{
	const int HWi = A.H * A.W;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const uint32_t *__restrict__ slots = A.hdr + A.slots_words;
	const cb_u32 first = (cb_u32)__builtin_amdgcn_readfirstlane((int)A.hdr[A.wtab_words + 2 * w]);   // slot + 1 of the wave's first entry
	cb_u32 n = (cb_u32)__builtin_amdgcn_readfirstlane((int)A.hdr[A.wtab_words + 2 * w + 1]);
	if (first == 0 || first > A.cap) return;
	n = min(n, A.cap - (first - 1));
	for (cb_u32 i = (cb_u32)lane; i < n; i += 64) {
		const cb_u4 e = *(const cb_u4 *)(slots + (size_t)(first - 1 + i) * 4);
		const cb_u32 rem = e.x - (cb_u32)d * (cb_u32)HWi;   // y * W + x
		if (rem >= (cb_u32)HWi) continue;   // (not this plane's: never written by cbca_classify_kernel)
		A.vout[(size_t)d * HWi + rem] = list_entry_value(A, rv, d, e, rem);
	}
}

}  // namespace

// ---- once per pair and direction: the outputs the lean kernel gets wrong ---------------------------------------------------
// An output (d, y, x) with a partner has the minimal support iff its combined arms (per-arm minimum of the two images,
// cbca.hip "Packed arm lengths") are all 1 and the rows above and below have left = right = 1 in its column -- the test of
// the strip kernel.  Everything else with a partner is listed.  The waves are the lean kernel's (same plane, rows, strip).
// Three launches and no atomic (one shared counter for 400 K short-lived waves measured 4 ms, one per plane 2 ms):
//   PASS 0  every wave COUNTS its listed outputs -> its word of the wave table;
//   scan    one block per plane turns the counts of the plane's waves into slot numbers (the slots are allotted per plane, capd each);
//   PASS 1  every wave WRITES its entries -- voxel index and the support's shape, looked up here, once per pair, instead of in
//           every pass -- into its own run of slots.
template <int PASS>
__global__ void __launch_bounds__(256) cbca_classify_kernel(const LeanArgs A)
{
	__shared__ cb_u32 bufs[PASS ? 4 : 1][PASS ? 512 : 1];
	if (!cbca_gate(A.flags, A.route)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	cb_u32 *__restrict__ buf = bufs[PASS ? wv : 0];
	if (PASS == 1 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // (the overflow word is the scan's)
		A.hdr[LH_D] = (uint32_t)A.D; A.hdr[LH_H] = (uint32_t)A.H; A.hdr[LH_W] = (uint32_t)A.W;
		A.hdr[LH_DIR] = (uint32_t)(A.direction + 1); A.hdr[LH_RB] = A.rbcode; A.hdr[LH_MAGIC] = LH_MAGIC_VALUE;
	}
	long long w;
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, w, d, y0, y1, xb)) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs = xb + 4 * lane;
	const cb_u32 OOB = 0x80000000u;
	uint32_t *__restrict__ slots = A.hdr + A.slots_words;
	cb_u32 first = 0, total = 0;   // PASS 1: slot + 1 of the wave's first entry, its entries (0: none -- or the plane's list does not fit)
	if (PASS == 1) {
		first = (cb_u32)__builtin_amdgcn_readfirstlane((int)A.hdr[A.wtab_words + 2 * w]);
		total = (cb_u32)__builtin_amdgcn_readfirstlane((int)A.hdr[A.wtab_words + 2 * w + 1]);
		if (first == 0 || first - 1 + total > ((cb_u32)d + 1u) * A.capd) return;
	}
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	cb_u32 want = 0;   // bit j: output column exists and has a partner (adcensus.cu:353)
#pragma unroll
	for (int j = 0; j < 4; ++j)
		if (xs + j < W && xs + j + sh >= 0 && xs + j + sh < W) want |= 1u << j;
	// combined lengths of row r for this lane's four columns (rows outside the image: 0 = "not the unit arm")
	auto fetch = [&](int r, cb_u32 (&m)[4]) {
		const bool rok = r >= 0 && r < H;
		const int base = r * W + xs;
		// the padded scratch makes any start readable; columns outside the image / the shifted range are never looked at (want)
		const cb_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
		const cb_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
		m[0] = bytemin4(a.x, b.x); m[1] = bytemin4(a.y, b.y); m[2] = bytemin4(a.z, b.z); m[3] = bytemin4(a.w, b.w);
	};
	int cnt = 0;         // PASS 0: the count; PASS 1: entries waiting in LDS
	cb_u32 written = 0;  // PASS 1: entries already in their slots
	auto flush = [&]() {
		for (int i = lane; i < cnt; i += 64) {
			const cb_u32 idx = buf[i];
			const int rem = (int)(idx - (cb_u32)d * (cb_u32)HWi);   // y * W + x
			const uint32_t mm = bytemin4(A.p0[rem], A.p1[rem + sh]);
			const int u = (int)((mm >> 16) & 0xffu), dn = (int)(mm >> 24);
			const int rows = u + dn + 1;
			bool fits = u <= 13 && dn <= 13 && rows <= 11;
			bool small = rows <= 4;   // at most four rows of at most four values
			cb_u32 ew[3] = {0u, 0u, 0u};
			for (int k = 0; k < rows && fits; ++k) {
				const int g = rem + (k - u) * W;
				const uint32_t m = bytemin4(A.p0[g], A.p1[g + sh]);
				const cb_u32 l = m & 0xffu, r = (m >> 8) & 0xffu;
				fits = l <= 15u && r <= 15u;
				small = small && l + r <= 3u;
				const int j = 1 + k;
				const cb_u32 byte = (l | (r << 4)) << ((j & 3) * 8);
				if (j < 4) ew[0] |= byte; else if (j < 8) ew[1] |= byte; else ew[2] |= byte;
			}
			if (!fits) { ew[0] = 0xffu; ew[1] = ew[2] = 0u; }
			else ew[0] |= small ? (0xe0u | (cb_u32)u | ((cb_u32)dn << 2)) : (cb_u32)(u | (dn << 4));
			if (written + (cb_u32)i < total) *(cb_u4 *)(slots + (size_t)(first - 1 + written + i) * 4) = cb_u4{idx, ew[0], ew[1], ew[2]};
		}
		written += (cb_u32)cnt;
		cnt = 0;
	};
	cb_u32 ma[4], mb[4], mc_[4], md[4];   // rows y - 1, y, y + 1 and, on its way, y + 2
	fetch(y0 - 1, ma);
	fetch(y0, mb);
	fetch(y0 + 1, mc_);
	for (int y = y0; y < y1; ++y) {
		fetch(y + 2 <= y1 ? y + 2 : -1, md);
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const bool minimal = mb[j] == 0x01010101u && (ma[j] & 0xffffu) == 0x0101u && (mc_[j] & 0xffffu) == 0x0101u;
			const bool listed = ((want >> j) & 1u) && !minimal;
			const unsigned long long bal = __ballot(listed);
			if (PASS == 1) {
				const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((cb_u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((cb_u32)bal, 0));
				if (listed) buf[pos] = (cb_u32)d * (cb_u32)HWi + (cb_u32)(y * W + xs + j);
			}
			cnt += __builtin_popcountll(bal);
		}
		if (PASS == 1 && cnt >= 256) flush();   // (a row adds at most 256 entries: 512 always hold them)
#pragma unroll
		for (int j = 0; j < 4; ++j) { ma[j] = mb[j]; mb[j] = mc_[j]; mc_[j] = md[j]; }
	}
	if (PASS == 0) {
		if (lane == 0) A.hdr[A.wtab_words + 2 * w + 1] = (cb_u32)cnt;
	} else if (cnt) {
		flush();
	}
}

// counts of a plane's waves -> slot numbers: wave table word 0 = slot + 1 of the wave's first entry (0: none).  One block per plane,
// the plane's npl waves are consecutive in the table; a plane whose entries exceed capd raises the overflow word (the list is then
// not used: the strip kernel runs the passes).
__global__ void __launch_bounds__(256) cbca_list_scan_kernel(const LeanArgs A, int npl)
{
	__shared__ cb_u32 wsum[4];
	if (!cbca_gate(A.flags, A.route)) return;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int d = blockIdx.x;
	uint32_t *__restrict__ tab = A.hdr + A.wtab_words + 2 * (size_t)d * npl;
	cb_u32 running = 0;
	for (int i0 = 0; i0 < npl; i0 += 256) {
		const int i = i0 + tid;
		const cb_u32 c = i < npl ? tab[2 * i + 1] : 0u;
		cb_u32 incl = c;   // inclusive scan inside the wave, then across the block's four waves
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const cb_u32 t = (cb_u32)__shfl_up((int)incl, o);
			if (lane >= o) incl += t;
		}
		if (lane == 63) wsum[wv] = incl;
		__syncthreads();
		cb_u32 before = 0, blk = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) { if (k < wv) before += wsum[k]; blk += wsum[k]; }
		const cb_u32 ex = running + before + incl - c;
		if (i < npl) tab[2 * i] = (c && ex + c <= A.capd) ? (cb_u32)d * A.capd + ex + 1u : 0u;
		running += blk;
		__syncthreads();
	}
	if (tid == 0 && running > A.capd) A.hdr[LH_OVERFLOW] = 1u;
}

// ---- per pass: the minimal 3 x 3 mean for every output, then the wave's listed outputs ------------------------------------------
// A wave owns R output rows x 256 columns of one plane: a lane holds the four columns 4 lane .. 4 lane + 3 of the wave's R + 2 rows
// (one aligned 16-byte load per row, all requested at once); the columns to the left and right come from the neighbouring lanes
// through DPP wave shifts, the strip's two outer columns from one extra 4-byte load in lanes 0 and 63 (lines the neighbouring strips
// read anyway).  Nine additions per output in the reference's order -- rows ascending, x ascending, accumulator starting at +0.0
// (adcensus.cu:356-373) -- and the division by 9; outputs without a partner are copied through.  What is wrong afterwards -- outputs
// with another support, among them every output on the image border, where the rows outside the image read as zeros -- is exactly
// what cbca_classify_kernel listed for this wave: their runs are requested with the rows and summed after them.
//
// SHORT-LIVED waves, dispatched in address order: what a plain copy reaches on this chip depends on how the waves in flight lie in
// memory -- 5.0 - 5.5 TB/s at 2 x 2 GB for a grid-stride copy of 16 K blocks, 6.3 / 6.7 TB/s (plain / non-temporal) for ONE 16-byte
// element per thread (profiles/r04_bw_sizes.txt).  The first form of this kernel (round 4, in the history: 6 K waves, each walking
// 125 rows of a region of its own through a register ring) was of the first kind and took 0.68 ms per pass alone, 0.85 with its
// listed outputs at the end (rows long gone from L2); this one 0.54 / 0.60 (profiles/r04_cbca_lean.txt, scripts/microbench/bw_lean.hip).
// order 1 (one band of rows per XCD, so that the two rows a wave shares with each vertical neighbour come out of that XCD's L2)
// measured equal to plain address order in the kernel (ahead in the microbenchmark); loads must NOT be non-temporal (the shared rows).
// s / 9 in three operations: q = s r, e = fma(-9, q, s), q' = fma(e, r, q) with r = RN(1 / 9) -- equal to the IEEE quotient for EVERY
// float with 2^-95 <= |s| < 2^125 (tests/test_div9.py walks all 2^32 bit patterns); outside that range the IEEE divide.
__device__ __forceinline__ float div9(float s)
{
	const float r = 0x1.c71c72p-4f;
	const float q = s * r;
	const float e = __builtin_fmaf(-9.0f, q, s);
	return __builtin_fmaf(e, r, q);
}
__device__ __forceinline__ bool div9_ok(float s) { const float a = __builtin_fabsf(s); return a >= 0x1p-95f && a < 0x1p125f; }

// POL: bit 0 non-temporal row loads, bit 1 non-temporal stores (measured: the two rows a wave shares with each vertical neighbour
// must stay in L2 -- plain loads; scripts/microbench/bw_lean.hip)
template <int R, int POL, bool INLINE_LIST>
__global__ void __launch_bounds__(256) cbca_lean_kernel(const LeanArgs A)
{
	constexpr int LAUX = (POL & 1) ? 2 : 0, SAUX = (POL & 2) ? 2 : 0;
	if (!lean_runs(A)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	long long w;
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, w, d, y0, y1, xb)) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs = xb + 4 * lane;
	const cb_u32 OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const bool lane_in = xs < W;
	const int ecol = lane == 0 ? xs - 1 : (lane == 63 ? xs + 4 : -1);
	const bool eok = ecol >= 0 && ecol < W;
	cb_u32 inr = 0;   // bit j: the output has a partner (adcensus.cu:353)
#pragma unroll
	for (int j = 0; j < 4; ++j)
		if (xs + j + sh >= 0 && xs + j + sh < W) inr |= 1u << j;
	// rows y0 - 1 .. y0 + R, all requested before the first is used (loads as in cbca_lean_kernel: no branch, rows outside the image: zeros)
	cb_u4 v[R + 2];
	cb_u32 e[R + 2];
#pragma unroll
	for (int k = 0; k < R + 2; ++k) {
		const int r = y0 - 1 + k;
		const bool rok = (unsigned)r < (unsigned)H;
		v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, (rok & lane_in) ? (cb_u32)(r * W + xs) * 4u : OOB, 0, LAUX);
		e[k] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok & eok) ? (cb_u32)(r * W + ecol) * 4u : OOB, 0, 0);
	}
	// the wave's listed outputs: its word of the wave table says where they are and how many -- requested now, under the rows
	const uint32_t *__restrict__ slots = A.hdr + A.slots_words;
	cb_u32 seg = 0, nent = 0;   // slot + 1 of the wave's first entry, its entries
	if (INLINE_LIST) {
		seg = (cb_u32)__builtin_amdgcn_readfirstlane((int)A.hdr[A.wtab_words + 2 * w]);
		nent = (cb_u32)__builtin_amdgcn_readfirstlane((int)A.hdr[A.wtab_words + 2 * w + 1]);
	}
	bool quick = INLINE_LIST && seg != 0 && seg - 1 + nent <= A.cap && nent <= 64u;   // at most a wave's worth of entries: one lane each
	cb_u4 ent = cb_u4{0u, 0u, 0u, 0u};
	if (quick && (cb_u32)lane < nent) ent = *(const cb_u4 *)(slots + (size_t)(seg - 1 + lane) * 4);
	// ... and the runs of the small-class entries (at most four rows of at most four values: nearly all of a texture's entries) behind
	// the rows: they are lines this wave and its neighbours are fetching anyway, and they arrive while the rows are summed
	const cb_u32 erem = ent.x - (cb_u32)d * (cb_u32)HWi;
	const bool ehas = quick && (cb_u32)lane < nent && erem < (cb_u32)HWi;
	const bool esmall = ehas && (ent.y & 0xf0u) == 0xe0u;
	if (__any(ehas && !esmall)) quick = false;   // (an entry of another class: the wave's entries go through list_phase, after its rows)
	cb_u4 ev[4];
	int enn[4];
	if (INLINE_LIST) small_request(A, rv, quick && esmall, ent, erem, ev, enn);

	// columns xs - 1 / xs + 4 of every row: the neighbouring lanes' last / first column (lane 0 / 63: the strip's outer columns)
	float nl[R + 2], nr[R + 2];
#pragma unroll
	for (int k = 0; k < R + 2; ++k) {
		nl[k] = lane_from_below(__uint_as_float(v[k].w), __uint_as_float(e[k]));
		nr[k] = lane_from_above(__uint_as_float(v[k].x), __uint_as_float(e[k]));
	}
#pragma unroll
	for (int k = 0; k < R; ++k) {
		const int yo = y0 + k;
		const float ra[6] = {nl[k], __uint_as_float(v[k].x), __uint_as_float(v[k].y), __uint_as_float(v[k].z), __uint_as_float(v[k].w), nr[k]};
		const float rb[6] = {nl[k + 1], __uint_as_float(v[k + 1].x), __uint_as_float(v[k + 1].y), __uint_as_float(v[k + 1].z), __uint_as_float(v[k + 1].w), nr[k + 1]};
		const float rc[6] = {nl[k + 2], __uint_as_float(v[k + 2].x), __uint_as_float(v[k + 2].y), __uint_as_float(v[k + 2].z), __uint_as_float(v[k + 2].w), nr[k + 2]};
		float sum[4], res[4];
		bool fast = true;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			float t = 0;
			t += ra[j]; t += ra[j + 1]; t += ra[j + 2];
			t += rb[j]; t += rb[j + 1]; t += rb[j + 2];
			t += rc[j]; t += rc[j + 1]; t += rc[j + 2];
			sum[j] = t;
			fast = fast && div9_ok(t);
			res[j] = div9(t);
		}
		if (__any(!fast)) {   // (zeros, denormals, huge values, infinities, NaNs somewhere in the wave's row)
#pragma unroll
			for (int j = 0; j < 4; ++j) res[j] = sum[j] / 9.0f;
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) res[j] = ((inr >> j) & 1u) ? res[j] : rb[j + 1];
		const bool myrow = yo < y1;
		const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, myrow ? (yo + 1) * W * 4 : 0, 0x00020000);
		__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])},
		                                       rrow, (myrow & lane_in) ? (cb_u32)(yo * W + xs) * 4u : OOB, 0, SAUX);
		__builtin_amdgcn_sched_barrier(0);   // (a row at a time: otherwise every row's sums are hoisted and the registers of all of them are live at once)
	}
	if (INLINE_LIST) {
		// the listed outputs: their values out of L1 / L2 (this wave and its neighbours have just read those rows), their stores after
		// the wave's own stores have completed (the same addresses, written by other lanes)
		if (quick) {
			const float val = small_sum(ev, enn);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (ehas) A.vout[(size_t)d * HWi + erem] = val;
		} else if (seg != 0) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			list_phase(A, w, d, lane);
		}
	}
}

// ... the listed outputs in a launch of their own (one wave per wave of the lean kernel): the form the inline one is measured against
__global__ void __launch_bounds__(256) cbca_list_kernel(const LeanArgs A)
{
	if (!lean_runs(A)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	long long w;
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, w, d, y0, y1, xb)) return;
	list_phase(A, w, d, lane);
}


// ---- two passes in one launch ---------------------------------------------------------------------------------------------------
// mc_predict aggregates a volume 2 + 16 times in a row (main.lua:998-1001, 1033-1039), and on a texture each pass is a 3 x 3 stencil
// that moves the whole volume through HBM.  cbca_lean2x_kernel runs TWO consecutive passes per launch: a wave reads R + 4 rows of 256
// columns of the plane, computes the R + 2 rows of the FIRST pass it needs (kept on chip: an LDS tile of its own, no barrier -- the LDS
// serves a wave's instructions in order), then its R x 252 outputs of the SECOND pass -- the plane is read 1.5 x and written once per
// two passes instead of read 2.5 x and written twice.  Both passes do, per output, exactly what cbca_lean_kernel does (the additions of
// the reference in its order, adcensus.cu:356-373, the division by 9, copy-through without a partner).
//
// The listed outputs (supports that are not the minimal 3 x 3) of the wave's R + 2 first-pass rows, columns 1 .. 254 of its 256, are
// written once per pair and direction by cbca_classify2x_kernel into the wave's RECORD -- a fixed 4 KB of the plan area: 16 bytes
// {entries, cost, -, -} and 255 slots of 16 bytes (voxel, shape; the entry format of the single-pass list; 256 more for the waves of the
// image's first and last rows, all of whose outputs are listed) -- so that the first 64
// entries are requested together with the rows, without a table lookup in front (a short-lived wave's time is its chain of dependent
// memory round trips, not its instructions: the first version, with the single-pass kernel's wave table -> slots -> values chain, a
// chain of arm lookups per value outside the tile and the listed outputs stored behind the rows' stores, took 1.08 ms per launch with
// 85 as with 58 vector instructions per row).  One lane per entry redoes the reference's loop:
//   first pass   out of the input plane (as in cbca_lean_kernel), the value replaces the tile's;
//   second pass  (entries among the wave's own R x 252 outputs) over the first pass's values: out of the tile where the support lies
//                inside it; a value outside (supports that reach over the tile's edge: arms >= 2 next to it) is recomputed from the
//                input plane on the spot (first_pass_value) -- the same additions in the same order.  The results replace the tile's
//                rows before they are stored (up to 64 entries; beyond, they are stored behind the rows' stores).
// A wave with more than 255 entries, or a plane whose second-pass entries would recompute more than cost_limit values per voxel
// (cbca_list_cost_kernel), makes the list unusable: the pair is not the texture this path is for, and the passes go one per launch.
// Columns: a wave's 256 columns start at 252 strip - 2; lanes 0 and 63 hold two outer columns each whose values are incomplete
// (no neighbour) and store only their two inner ones.
constexpr int L2X_REC_WORDS = 1024;   // a wave's record: 4 words of head + 255 slots of 4 words
constexpr int L2X_SLOTS = 255;
// The waves of an image's first and last row chunk hold a whole image row of entries (a border output's support is never the minimal one):
// they get a second record of 256 slots, behind all the first ones -- per plane one row of them for the top chunks, two for the bottom ones.
struct L2xRecords {
	uint32_t *rec, *xrec;   // the wave's record; its second one (border chunks) or null
	cb_u32 cap;             // entries the wave can hold
};
__device__ __forceinline__ L2xRecords l2x_records(const LeanArgs &A, long long w, int d, int y0, int y1, int xb)
{
	L2xRecords r;
	r.rec = A.hdr + LH_WORDS + (size_t)w * L2X_REC_WORDS;
	const int border = y0 == 0 ? 0 : (y1 == A.H ? 1 : (y1 == A.H - 1 ? 2 : -1));   // (2: the chunk above a last chunk of one row -- the image's last row is its tile's)
	const int strip = (xb - A.xoff) / A.pitch;
	const size_t firsts = (size_t)A.gx * A.gy * A.D;
	r.xrec = border >= 0 ? A.hdr + LH_WORDS + (firsts + ((size_t)d * 3 + border) * A.gx + strip) * L2X_REC_WORDS : nullptr;
	r.cap = border >= 0 ? (cb_u32)(L2X_SLOTS + 256) : (cb_u32)L2X_SLOTS;
	return r;
}
__device__ __forceinline__ uint32_t *l2x_slot(const L2xRecords &r, cb_u32 i)
{
	return i < (cb_u32)L2X_SLOTS ? r.rec + 4 + 4 * (size_t)i : r.xrec + 4 * (size_t)(i - (cb_u32)L2X_SLOTS);
}

// first-pass value of pixel (yy, xx) of this plane, from the input plane.  The common case on a texture -- the minimal 3 x 3 support -- is
// requested together with the arm lengths that decide it (one round trip); anything else walks the support with the reference's loop.
__device__ __forceinline__ float first_pass_value(const LeanArgs &A, const __amdgpu_buffer_rsrc_t &rv, int sh, int yy, int xx)
{
	const int W = A.W, H = A.H;
	const int g = yy * W + xx;
	const cb_u32 OOB = 0x80000000u;
	if (xx + sh < 0 || xx + sh >= W) return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, (cb_u32)g * 4u, 0, 0));   // adcensus.cu:353-354
	const bool inner = yy >= 1 && yy + 1 < H && xx >= 1 && xx + 1 < W;
	const int gu = inner ? g - W : g, gd = inner ? g + W : g;
	const uint32_t mm = bytemin4(A.p0[g], A.p1[g + sh]);
	const uint32_t mu = bytemin4(A.p0[gu], A.p1[gu + sh]), md = bytemin4(A.p0[gd], A.p1[gd + sh]);
	float r0[3], r1[3], r2[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		r0[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, inner ? (cb_u32)(g - W - 1 + k) * 4u : OOB, 0, 0));
		r1[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, inner ? (cb_u32)(g - 1 + k) * 4u : OOB, 0, 0));
		r2[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, inner ? (cb_u32)(g + W - 1 + k) * 4u : OOB, 0, 0));
	}
	if (inner && mm == 0x01010101u && (mu & 0xffffu) == 0x0101u && (md & 0xffffu) == 0x0101u) {
		float t = 0;
		t += r0[0]; t += r0[1]; t += r0[2];
		t += r1[0]; t += r1[1]; t += r1[2];
		t += r2[0]; t += r2[1]; t += r2[2];
		return t / 9.0f;
	}
	const int u = (int)((mm >> 16) & 0xffu), dn = (int)(mm >> 24);
	float sum = 0;
	int cnt = 0;
	for (int q = -u; q <= dn; ++q) {
		const int gq = g + q * W;
		const uint32_t m = bytemin4(A.p0[gq], A.p1[gq + sh]);
		const int l = (int)(m & 0xffu), nk = l + (int)((m >> 8) & 0xffu) + 1;
		for (int k = 0; k < nk; ++k) sum += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, (cb_u32)(gq - l + k) * 4u, 0, 0));
		cnt += nk;
	}
	return sum / (float)cnt;
}

// second-pass value of the listed output (y, x) (entry e, rem = y * W + x): the reference's loop over the first pass's values
template <int R>
__device__ __forceinline__ float second_pass_entry(const LeanArgs &A, const __amdgpu_buffer_rsrc_t &rv, const float *__restrict__ T, int d, const cb_u4 &e,
                                                   cb_u32 rem, int y, int x, int y0, int xb)
{
	const int W = A.W;
	const int sh = d * A.direction;
	const cb_u32 b0 = e.y & 0xffu;
	const bool lookup = b0 == 0xffu, small = (b0 & 0xf0u) == 0xe0u;
	int u, rows;
	if (lookup) {
		const uint32_t mm = bytemin4(A.p0[rem], A.p1[(int)rem + sh]);
		u = (int)((mm >> 16) & 0xffu);
		rows = u + (int)(mm >> 24) + 1;
	} else if (small) {
		u = (int)(b0 & 3u);
		rows = u + (int)((b0 >> 2) & 3u) + 1;
	} else {
		u = (int)(b0 & 15u);
		rows = u + (int)(b0 >> 4) + 1;
	}
	float sum = 0;
	int cnt = 0;
	for (int k = 0; k < rows; ++k) {
		const int yy = y + k - u;
		int l, nk;
		if (lookup) {
			const int g = (int)rem + (k - u) * W;
			const uint32_t m = bytemin4(A.p0[g], A.p1[g + sh]);
			l = (int)(m & 0xffu);
			nk = l + (int)((m >> 8) & 0xffu) + 1;
		} else {
			const cb_u32 lr = entry_byte(e, 1 + k);
			l = (int)(lr & 15u);
			nk = l + (int)(lr >> 4) + 1;
		}
		const int ry = yy - (y0 - 1);
		const bool row_in = (unsigned)ry < (unsigned)(R + 2);
		for (int c = 0; c < nk; ++c) {
			const int xx = x - l + c;
			const int cx = xx - xb;
			float t;
			if (row_in && (unsigned)(cx - 1) < 254u) t = T[ry * 256 + cx];
			else t = first_pass_value(A, rv, sh, yy, xx);
			sum += t;
		}
		cnt += nk;
	}
	return sum / (float)cnt;
}

// a row of the lane's four columns with the neighbouring lanes' columns beside them (columns xs - 1 .. xs + 4): two DPP moves per row,
// shared by the three output rows the row is an operand of
struct Row6 { float c[6]; };
__device__ __forceinline__ Row6 with_neighbours(const cb_f4 &v)
{
	Row6 r;
	r.c[0] = lane_from_below(v.w, 0.0f); r.c[1] = v.x; r.c[2] = v.y; r.c[3] = v.z; r.c[4] = v.w; r.c[5] = lane_from_above(v.x, 0.0f);
	return r;
}

// one row of a pass: the minimal 3 x 3 mean of the lane's four columns out of rows a, b, c, copy-through where the output has no partner
// (allin: every output of the wave has one -- wave-uniform).  The reference's sum starts at +0.0 (adcensus.cu:356); 0 + x is x except
// for x = -0.0, and the sum of the nine then differs only if it is a zero -- which div9_ok sends to the exact path below, together with
// everything else the three-operation division does not cover: there the chain is redone from +0.0.  So the common path adds eight times.
__device__ __forceinline__ cb_f4 lean_row(const Row6 &a, const Row6 &b, const Row6 &c, cb_u32 inr, bool allin)
{
	float res[4];
	bool fast = true;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		float t = a.c[j] + a.c[j + 1];
		t += a.c[j + 2];
		t += b.c[j]; t += b.c[j + 1]; t += b.c[j + 2];
		t += c.c[j]; t += c.c[j + 1]; t += c.c[j + 2];
		fast = fast && div9_ok(t);
		res[j] = div9(t);
	}
	if (__any(!fast)) {   // (zeros, denormals, huge values, infinities, NaNs somewhere in the wave's row)
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			float t = 0;
			t += a.c[j]; t += a.c[j + 1]; t += a.c[j + 2];
			t += b.c[j]; t += b.c[j + 1]; t += b.c[j + 2];
			t += c.c[j]; t += c.c[j + 1]; t += c.c[j + 2];
			res[j] = t / 9.0f;
		}
	}
	if (!allin) {
#pragma unroll
		for (int j = 0; j < 4; ++j) res[j] = ((inr >> j) & 1u) ? res[j] : b.c[j + 1];
	}
	return cb_f4{res[0], res[1], res[2], res[3]};
}

// first-pass values of the n <= 4 pixels (yy, xx0 .. xx0 + n - 1) of this plane, from the input plane: a support row of a second-pass entry
// that lies outside the wave's tile.  The common case on a texture -- all of them with the minimal 3 x 3 support -- out of ONE round trip:
// three rows of six values and the arm lengths that decide it are requested together; anything else pixel by pixel (first_pass_value).
__device__ __forceinline__ void first_pass_row(const LeanArgs &A, const __amdgpu_buffer_rsrc_t &rv, const __amdgpu_buffer_rsrc_t &rp0,
                                               const __amdgpu_buffer_rsrc_t &rp1, int sh, int yy, int xx0, int n, float (&out)[4])
{
	const int W = A.W, H = A.H;
	const cb_u32 OOB = 0x80000000u;
	const int g0 = yy * W + xx0;
	const bool vec = yy >= 1 && yy + 1 < H && xx0 >= 1 && xx0 + n < W && xx0 + sh >= 0 && xx0 + n - 1 + sh < W;   // inside the image with a ring around, partners for all
	cb_u4 a[3], b[3], lo[3];
	cb_u2 hi[3];
#pragma unroll
	for (int q = 0; q < 3; ++q) {
		const int g = g0 + (q - 1) * W;
		a[q] = __builtin_amdgcn_raw_buffer_load_b128(rp0, vec ? (cb_u32)(g + CS_PAD) * 4u : OOB, 0, 0);
		b[q] = __builtin_amdgcn_raw_buffer_load_b128(rp1, vec ? (cb_u32)(g + sh + CS_PAD) * 4u : OOB, 0, 0);
		lo[q] = __builtin_amdgcn_raw_buffer_load_b128(rv, vec ? (cb_u32)(g - 1) * 4u : OOB, 0, 0);
		hi[q] = __builtin_amdgcn_raw_buffer_load_b64(rv, vec ? (cb_u32)(g + 3) * 4u : OOB, 0, 0);
	}
	const cb_u32 au[4] = {a[0].x, a[0].y, a[0].z, a[0].w}, bu[4] = {b[0].x, b[0].y, b[0].z, b[0].w};
	const cb_u32 am[4] = {a[1].x, a[1].y, a[1].z, a[1].w}, bm[4] = {b[1].x, b[1].y, b[1].z, b[1].w};
	const cb_u32 ad[4] = {a[2].x, a[2].y, a[2].z, a[2].w}, bd[4] = {b[2].x, b[2].y, b[2].z, b[2].w};
	bool allmin = vec;
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const bool minimal = bytemin4(am[c], bm[c]) == 0x01010101u && (bytemin4(au[c], bu[c]) & 0xffffu) == 0x0101u && (bytemin4(ad[c], bd[c]) & 0xffffu) == 0x0101u;
		allmin = allmin && (c >= n || minimal);
	}
	if (allmin) {
		float r[3][6];
#pragma unroll
		for (int q = 0; q < 3; ++q) {
			r[q][0] = __uint_as_float(lo[q].x); r[q][1] = __uint_as_float(lo[q].y); r[q][2] = __uint_as_float(lo[q].z); r[q][3] = __uint_as_float(lo[q].w);
			r[q][4] = __uint_as_float(hi[q].x); r[q][5] = __uint_as_float(hi[q].y);
		}
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			float t = 0;
			t += r[0][c]; t += r[0][c + 1]; t += r[0][c + 2];
			t += r[1][c]; t += r[1][c + 1]; t += r[1][c + 2];
			t += r[2][c]; t += r[2][c + 1]; t += r[2][c + 2];
			out[c] = t / 9.0f;
		}
	} else {
		for (int c = 0; c < n; ++c) {
			const float t = first_pass_value(A, rv, sh, yy, xx0 + c);
#pragma unroll
			for (int j = 0; j < 4; ++j) out[j] = j == c ? t : out[j];
		}
	}
}

// second-pass value of a small-class entry (at most four rows of at most four values: nearly all of a texture's entries): the rows inside
// the tile are read at once, a row outside it is recomputed as a row (first_pass_row), then the reference's additions in their order
template <int R>
__device__ __forceinline__ float second_pass_small(const LeanArgs &A, const __amdgpu_buffer_rsrc_t &rv, const __amdgpu_buffer_rsrc_t &rp0,
                                                   const __amdgpu_buffer_rsrc_t &rp1, const float *__restrict__ T, int sh, bool on, const cb_u4 &e,
                                                   int y, int x, int y0, int xb)
{
	const cb_u32 b0 = e.y & 0xffu;
	const int u = (int)(b0 & 3u), rows = on ? u + (int)((b0 >> 2) & 3u) + 1 : 0;
	float t[4][4];
	int nn[4];
	// first_pass_row once per support-row INDEX at which any lane's row lies outside the tile (a wave of a texture: two to three of the four)
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const cb_u32 lr = entry_byte(e, 1 + k);
		const int l = (int)(lr & 15u);
		nn[k] = k < rows ? l + (int)(lr >> 4) + 1 : 0;
		const int yy = y + k - u, ry = yy - (y0 - 1), cx0 = x - l - xb;
		const bool inside = (unsigned)ry < (unsigned)(R + 2) && cx0 >= 1 && cx0 + nn[k] - 1 <= 254;
		const int base = inside ? ry * 256 + cx0 : 0;
#pragma unroll
		for (int c = 0; c < 4; ++c) t[k][c] = T[c < nn[k] ? base + c : 0];
		const bool need = nn[k] > 0 && !inside;
		if (__any(need)) {
			if (need) first_pass_row(A, rv, rp0, rp1, sh, yy, x - l, nn[k], t[k]);
		}
	}
	float sum = 0;
	int cnt = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
#pragma unroll
		for (int c = 0; c < 4; ++c) sum += c < nn[k] ? t[k][c] : -0.0f;
		cnt += nn[k];
	}
	return sum / (float)cnt;
}

#define MC_LEAN2X_WPB 1
constexpr int L2X_WPB = MC_LEAN2X_WPB;   // waves per block: 10 KB of LDS per wave (R = 8); blocks of two waves fill a CU's 160 KB in finer steps than blocks of four

// a listed output of the wave's tile, from its record entry (word 0 = first-pass row | column << 8)
struct L2xEntry { int ry, cx, y, x; cb_u32 rem; bool first, own; };
template <int R>
__device__ __forceinline__ L2xEntry l2x_entry(const cb_u4 &e, bool has, int y0, int y1, int xb, int W)
{
	L2xEntry t;
	t.ry = (int)(e.x & 0xffu); t.cx = (int)(e.x >> 8);
	t.y = y0 - 1 + t.ry; t.x = xb + t.cx;
	t.rem = (cb_u32)(t.y * W + t.x);
	t.first = has && t.ry < R + 2 && (unsigned)t.cx < 256u;                                         // (written by cbca_classify2x_kernel: inside the image)
	t.own = t.first && t.y >= y0 && t.y < y1 && (unsigned)(t.cx - 2) < 252u;                        // ... also one of the wave's own outputs
	return t;
}

template <int R, int WPB>
__global__ void __launch_bounds__(64 * WPB) cbca_lean2x_kernel(const LeanArgs A)
{
	__shared__ __attribute__((aligned(16))) float tiles[WPB][(R + 2) * 256];
	if (!lean_runs(A)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	long long w;
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, w, d, y0, y1, xb)) return;
	float *__restrict__ T = tiles[wv];
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs = xb + 4 * lane;
	const cb_u32 OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	// the wave's record: its first 64 entries are requested before the rows (loads return in order: they are here first), the count beside them
	const L2xRecords rr = l2x_records(A, w, d, y0, y1, xb);
	const uint32_t *__restrict__ rec = rr.rec;
	const cb_u4 ent = *(const cb_u4 *)(rec + 4 + 4 * lane);
	const cb_u32 nent = min((cb_u32)__builtin_amdgcn_readfirstlane((int)rec[0]), rr.cap);
	cb_u32 inr = 0;   // bit j: the output has a partner (adcensus.cu:353)
#pragma unroll
	for (int j = 0; j < 4; ++j)
		if (xs + j + sh >= 0 && xs + j + sh < W) inr |= 1u << j;
	// input rows y0 - 2 .. y0 + R + 1, all requested before the first is used; rows outside the image: zeros.  The first strip's lane 0
	// starts two pixels before its row: the previous row's last two -- never an operand of an output that is not listed -- except where
	// that offset lies before the plane (row 0; row 1 of an image one pixel wide): those rows' two real columns come from loads of their own.
	cb_u4 v[R + 4];
#pragma unroll
	for (int k = 0; k < R + 4; ++k) {
		const int r = y0 - 2 + k;
		const bool ok = (unsigned)r < (unsigned)H && xs < W && r * W + xs >= 0;
		v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, ok ? (cb_u32)(r * W + xs) * 4u : OOB, 0, 0);
	}
	const bool edge0 = y0 == 0 && xs < 0 && 0 < H, edge1 = y0 == 0 && xs < 0 && W + xs < 0 && 1 < H;
	const cb_u2 fx0 = __builtin_amdgcn_raw_buffer_load_b64(rv, edge0 ? 0u : OOB, 0, 0);
	const cb_u2 fx1 = __builtin_amdgcn_raw_buffer_load_b64(rv, edge1 ? (cb_u32)W * 4u : OOB, 0, 0);
	// the entries' places; the runs of the small-class ones (at most four rows of at most four values: nearly all of a texture's
	// entries) are requested now, behind the rows: lines this wave and its neighbours are fetching anyway
	const L2xEntry E = l2x_entry<R>(ent, (cb_u32)lane < nent, y0, y1, xb, W);
	const bool esmall = E.first && (ent.y & 0xf0u) == 0xe0u;
	cb_u4 ev[4];
	int enn[4];
	small_request(A, rv, esmall, ent, E.rem, ev, enn);
	if (edge0) { v[2].z = fx0.x; v[2].w = fx0.y; }
	if (edge1) { v[3].z = fx1.x; v[3].w = fx1.y; }
	const bool allin = __all(inr == 15u);

	// first pass: rows y0 - 1 .. y0 + R into the tile
	{
		Row6 a = with_neighbours(__builtin_bit_cast(cb_f4, v[0])), b = with_neighbours(__builtin_bit_cast(cb_f4, v[1]));
#pragma unroll
		for (int k = 0; k < R + 2; ++k) {
			const Row6 c = with_neighbours(__builtin_bit_cast(cb_f4, v[k + 2]));
			*(cb_f4 *)(T + k * 256 + 4 * lane) = lean_row(a, b, c, inr, allin);
			a = b; b = c;
			__builtin_amdgcn_sched_barrier(0);   // (a row at a time: otherwise every row's sums are hoisted and the registers of all of them are live at once)
		}
	}
	// ... its listed outputs, out of the input plane
	if (E.first) T[E.ry * 256 + E.cx] = esmall ? small_sum(ev, enn) : list_entry_value(A, rv, d, ent, E.rem);
	for (cb_u32 i = 64u + (cb_u32)lane; i < nent; i += 64) {
		const cb_u4 e = *(const cb_u4 *)l2x_slot(rr, i);
		const L2xEntry F = l2x_entry<R>(e, true, y0, y1, xb, W);
		if (F.first) T[F.ry * 256 + F.cx] = list_entry_value(A, rv, d, e, F.rem);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

	// second pass.  Its listed outputs first, over the first pass's values in the tile -- up to 64 entries (one per lane: nearly every
	// wave of a texture): their values replace the rows' in the tile, whose first-pass values nobody needs any more by then, and the rows
	// are stored from there.  More entries: the rows are stored first and the entries' values behind them, once those stores have
	// completed (the same addresses, written by other lanes).
	const bool through_tile = nent <= 64u;
	float val2 = 0;
	if (through_tile && nent) {
		const bool own_small = E.own && esmall;
		if (__any(own_small)) val2 = second_pass_small<R>(A, rv, rp0, rp1, T, sh, own_small, ent, E.y, E.x, y0, xb);
		if (E.own && !esmall) val2 = second_pass_entry<R>(A, rv, T, d, ent, E.rem, E.y, E.x, y0, xb);
	}
	cb_f4 res[R];
	{
		cb_f4 m[R + 2];
#pragma unroll
		for (int k = 0; k < R + 2; ++k) m[k] = *(const cb_f4 *)(T + k * 256 + 4 * lane);
		Row6 ma = with_neighbours(m[0]), mb = with_neighbours(m[1]);
#pragma unroll
		for (int k = 0; k < R; ++k) {
			const Row6 mc = with_neighbours(m[k + 2]);
			res[k] = lean_row(ma, mb, mc, inr, allin);
			ma = mb; mb = mc;
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	if (through_tile && nent) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();   // (every lane's reads of first-pass values are done)
#pragma unroll
		for (int k = 0; k < R; ++k) *(cb_f4 *)(T + k * 256 + 4 * lane) = res[k];
		if (E.own) T[(E.y - y0) * 256 + E.cx] = val2;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
		for (int k = 0; k < R; ++k) res[k] = *(const cb_f4 *)(T + k * 256 + 4 * lane);
	}
	const bool edge = lane == 0 || lane == 63;
	const int xe = lane == 0 ? xs + 2 : xs;   // first of an edge lane's two stored columns
#pragma unroll
	for (int k = 0; k < R; ++k) {
		const int yo = y0 + k;
		const bool myrow = yo < y1;
		const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, myrow ? (yo + 1) * W * 4 : 0, 0x00020000);
		__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[k].x), __float_as_uint(res[k].y), __float_as_uint(res[k].z), __float_as_uint(res[k].w)},
		                                       rrow, (myrow && !edge && xs < W) ? (cb_u32)(yo * W + xs) * 4u : OOB, 0, 0);
		const cb_u2 two = lane == 0 ? cb_u2{__float_as_uint(res[k].z), __float_as_uint(res[k].w)} : cb_u2{__float_as_uint(res[k].x), __float_as_uint(res[k].y)};
		__builtin_amdgcn_raw_buffer_store_b64(two, rrow, (myrow && edge && xe >= 0 && xe < W) ? (cb_u32)(yo * W + xe) * 4u : OOB, 0, 0);
	}
	if (!through_tile) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		for (cb_u32 i = (cb_u32)lane; i < nent; i += 64) {
			const cb_u4 e = *(const cb_u4 *)l2x_slot(rr, i);
			const L2xEntry F = l2x_entry<R>(e, true, y0, y1, xb, W);
			if (F.own) A.vout[(size_t)d * HWi + F.rem] = second_pass_entry<R>(A, rv, T, d, e, F.rem, F.y, F.x, y0, xb);
		}
	}
}

// once per pair and direction: every wave of cbca_lean2x_kernel lists, in its record, the outputs of its R + 2 first-pass rows (columns
// 1 .. 254 of its 256) whose support is not the minimal 3 x 3 -- the test of cbca_classify_kernel -- with the support's shape, and what
// its second-pass entries (those among its own outputs) cost beyond their own supports: a first-pass value outside the wave's tile is
// recomputed at the price of its own support, estimated as (values outside the tile) x (values of the entry's support).
__global__ void __launch_bounds__(256) cbca_classify2x_kernel(const LeanArgs A)
{
	__shared__ cb_u32 bufs[4][512];
	if (!cbca_gate(A.flags, A.route)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	cb_u32 *__restrict__ buf = bufs[wv];
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {   // (the overflow word: zeroed before the launch, raised below and by cbca_list_cost_kernel)
		A.hdr[LH_D] = (uint32_t)A.D; A.hdr[LH_H] = (uint32_t)A.H; A.hdr[LH_W] = (uint32_t)A.W;
		A.hdr[LH_DIR] = (uint32_t)(A.direction + 1); A.hdr[LH_RB] = A.rbcode; A.hdr[LH_MAGIC] = LH_MAGIC_VALUE;
	}
	long long w;
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, w, d, y0, y1, xb)) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs = xb + 4 * lane;
	const cb_u32 OOB = 0x80000000u;
	const L2xRecords rr = l2x_records(A, w, d, y0, y1, xb);
	uint32_t *__restrict__ rec = rr.rec;
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	cb_u32 want = 0;   // bit j: the column exists, has a partner (adcensus.cu:353) and is one of the tile's columns 1 .. 254
#pragma unroll
	for (int j = 0; j < 4; ++j)
		if (xs + j >= 0 && xs + j < W && xs + j + sh >= 0 && xs + j + sh < W && (unsigned)(4 * lane + j - 1) < 254u) want |= 1u << j;
	// combined lengths of row r for this lane's four columns (rows outside the image: 0 = "not the unit arm")
	auto fetch = [&](int r, cb_u32 (&m)[4]) {
		const bool rok = r >= 0 && r < H;
		const int base = r * W + xs;
		const cb_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
		const cb_u4 b = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
		m[0] = bytemin4(a.x, b.x); m[1] = bytemin4(a.y, b.y); m[2] = bytemin4(a.z, b.z); m[3] = bytemin4(a.w, b.w);
	};
	int cnt = 0;         // entries waiting in LDS
	cb_u32 written = 0;  // entries already in the record (or dropped: more than the record holds)
	float cost = 0;
	auto flush = [&]() {
		for (int i = lane; i < cnt; i += 64) {
			const cb_u32 idx = buf[i];
			const int rem = (int)(idx - (cb_u32)d * (cb_u32)HWi);   // y * W + x
			const uint32_t mm = bytemin4(A.p0[rem], A.p1[rem + sh]);
			const int u = (int)((mm >> 16) & 0xffu), dn = (int)(mm >> 24);
			const int rows = u + dn + 1;
			bool fits = u <= 13 && dn <= 13 && rows <= 11;
			bool small = rows <= 4;   // at most four rows of at most four values
			cb_u32 ew[3] = {0u, 0u, 0u};
			const int ye = rem / W, xe = rem - ye * W;
			const bool own = ye >= y0 && ye < y1 && (unsigned)(xe - xb - 2) < 252u;
			float taps = 0, outside = 0;
			for (int k = 0; k < rows && (fits || own); ++k) {
				const int g = rem + (k - u) * W;
				const uint32_t m = bytemin4(A.p0[g], A.p1[g + sh]);
				const cb_u32 l = m & 0xffu, r = (m >> 8) & 0xffu;
				if (own) {
					const int ry = ye + k - u - (y0 - 1);
					int in = 0;
					if ((unsigned)ry < (unsigned)(A.rb + 2)) in = max(0, min(xe + (int)r, xb + 254) - max(xe - (int)l, xb + 1) + 1);
					taps += (float)(l + r + 1u);
					outside += (float)((int)(l + r + 1u) - in);
				}
				if (fits) {
					fits = l <= 15u && r <= 15u;
					small = small && l + r <= 3u;
					const int j = 1 + k;
					const cb_u32 byte = (l | (r << 4)) << ((j & 3) * 8);
					if (j < 4) ew[0] |= byte; else if (j < 8) ew[1] |= byte; else ew[2] |= byte;
				}
			}
			cost += outside * taps;
			if (!fits) { ew[0] = 0xffu; ew[1] = ew[2] = 0u; }
			else ew[0] |= small ? (0xe0u | (cb_u32)u | ((cb_u32)dn << 2)) : (cb_u32)(u | (dn << 4));
			// (word 0: the output's place in the wave's tile -- first-pass row | column << 8 -- instead of its voxel index: no division in the passes)
			if (written + (cb_u32)i < rr.cap) *(cb_u4 *)l2x_slot(rr, written + (cb_u32)i) = cb_u4{(cb_u32)(ye - (y0 - 1)) | ((cb_u32)(xe - xb) << 8), ew[0], ew[1], ew[2]};
		}
		written += (cb_u32)cnt;
		cnt = 0;
	};
	cb_u32 ma[4], mb[4], mc_[4], md[4];   // rows y - 1, y, y + 1 and, on its way, y + 2
	const int ya = y0 - 1, yb = y1 + 1;   // the tile's first-pass rows (those inside the image)
	fetch(ya - 1, ma);
	fetch(ya, mb);
	fetch(ya + 1, mc_);
	for (int y = ya; y < yb; ++y) {
		fetch(y + 2 <= yb ? y + 2 : -1, md);
		const bool rowin = (unsigned)y < (unsigned)H;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const bool minimal = mb[j] == 0x01010101u && (ma[j] & 0xffffu) == 0x0101u && (mc_[j] & 0xffffu) == 0x0101u;
			const bool listed = rowin && ((want >> j) & 1u) && !minimal;
			const unsigned long long bal = __ballot(listed);
			const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((cb_u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((cb_u32)bal, 0));
			if (listed) buf[pos] = (cb_u32)d * (cb_u32)HWi + (cb_u32)(y * W + xs + j);
			cnt += __builtin_popcountll(bal);
		}
		if (cnt >= 256) flush();   // (a row adds at most 256 entries: 512 always hold them)
#pragma unroll
		for (int j = 0; j < 4; ++j) { ma[j] = mb[j]; mb[j] = mc_[j]; mc_[j] = md[j]; }
	}
	if (cnt) flush();
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) cost += __shfl_xor(cost, o);
	if (lane == 0) {
		*(cb_u4 *)rec = cb_u4{written, __float_as_uint(cost), 0u, 0u};
		if (written > rr.cap) A.hdr[LH_OVERFLOW] = 1u;
	}
}

// ... after it: one block per plane adds the plane's waves' costs; more than cost_limit values recomputed per voxel on average means the
// pair has regions of large supports next to its texture -- the list is declared unusable (the overflow word)
#define MC_LEAN2X_COST 2.0f
__global__ void __launch_bounds__(256) cbca_list_cost_kernel(const LeanArgs A, int npl)
{
	__shared__ float wsum[4];
	if (!cbca_gate(A.flags, A.route)) return;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t *__restrict__ recs = A.hdr + LH_WORDS + (size_t)blockIdx.x * npl * L2X_REC_WORDS;
	float t = 0;
	for (int i = tid; i < npl; i += 256) t += __uint_as_float(recs[(size_t)i * L2X_REC_WORDS + 1]);
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
	if (lane == 0) wsum[wv] = t;
	__syncthreads();
	if (tid == 0 && !(wsum[0] + wsum[1] + wsum[2] + wsum[3] <= A.cost_limit * (float)A.H * (float)A.W)) A.hdr[LH_OVERFLOW] = 1u;
}

// before a classification: the head of the plan area back to "no list, no plan" -- if this pair is classified at all
__global__ void __launch_bounds__(64) cbca_list_reset_kernel(const LeanArgs A)
{
	if (!cbca_gate(A.flags, A.route)) return;
	if (threadIdx.x < LH_WORDS) A.hdr[threadIdx.x] = 0u;
}

// rows per wave / launch variant mc_predict uses (cfg.lean_rb = 0 / cfg.lean_variant < 0): measured at 1000 x 1500 x 256, one box
// (profiles/r04_cbca_lean.txt): 8 rows 0.602 / 0.620 ms (address order / a band per XCD), 4 rows 0.601 / 0.601, 2 rows 0.652 / 0.635;
// the classification costs 0.54 / 0.88 / 1.3 ms per direction
#define MC_LEAN_RB_DEFAULT 8
#define MC_LEAN_VARIANT_DEFAULT 0
static int lean_rows(int rb) { return (rb == 2 || rb == 4 || rb == 8) ? rb : MC_LEAN_RB_DEFAULT; }
// ... of the two-pass kernel (cfg.lean_rb; measured at 1000 x 1500 x 256: DESIGN section 7)
#define MC_LEAN2X_RB_DEFAULT 8
static int lean2x_rows(int rb) { return (rb == 4 || rb == 6 || rb == 8 || rb == 10 || rb == 12) ? rb : MC_LEAN2X_RB_DEFAULT; }

static LeanArgs lean_args(const void *packed, void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
                          int route, int rb, int cap_limit = 0, int order = 0, bool two_pass = false)
{
	LeanArgs A;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = vin; A.vout = vout;
	A.hdr = (uint32_t *)plan;
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.pitch = two_pass ? 252 : 256; A.xoff = two_pass ? -2 : 0;
	A.wpb = 4;
	A.gx = (int)cdiv(W, A.pitch);
	A.rb = two_pass ? lean2x_rows(rb) : lean_rows(rb);
	A.rbcode = (cb_u32)A.rb | (two_pass ? 0x100u : 0u);
	A.gy = (int)cdiv(H, A.rb);
	A.order = order;
	A.gyb = (int)cdiv(A.gy, 8);
	A.gx_rcp = (cb_u32)((((uint64_t)1 << 32) + A.gx - 1) / A.gx);
	const int64_t waves = (int64_t)A.gx * A.gy * D;
	// [header LH_WORDS | wave table: 2 words per wave, canonical order (plane, chunk, strip) | slots: 16 bytes each, capd per plane]
	// (two-pass geometry: [header LH_WORDS | a record of L2X_REC_WORDS per wave]; capd = 0 where the records do not fit)
	A.wtab_words = LH_WORDS;
	A.slots_words = (uint32_t)((A.wtab_words + 2 * waves + 3) / 4 * 4);
	const int64_t room = (int64_t)plan_bytes / 4 - A.slots_words;
	int64_t cap = std::max<int64_t>(0, std::min<int64_t>(room / 4, 0x3ffffff0));
	if (cap_limit > 0) cap = std::min<int64_t>(cap, cap_limit);   // (test hook: a list that does not fit)
	A.capd = (uint32_t)(cap / std::max(1, D));
	if (two_pass) A.capd = ((int64_t)plan_bytes / 4 >= LH_WORDS + (waves + (int64_t)A.gx * 3 * D) * L2X_REC_WORDS) ? (uint32_t)L2X_SLOTS : 0u;
	A.cap = A.capd * (uint32_t)D;
	A.flags = cs.flag;
	A.route = route;
	A.cost_limit = MC_LEAN2X_COST;
	return A;
}

// rows per wave of the lean / classify kernels for a problem (rb > 0: forced): the list is valid for this value only
int cbca_lean_rows(int D, int H, int W, int rb, bool two_pass)
{
	(void)D; (void)H; (void)W;
	return two_pass ? (lean2x_rows(rb) | 0x100) : lean_rows(rb);   // (the value of the list header's LH_RB word)
}

// blocks of 4 waves: x over one plane's (chunk, strip) pairs -- order 1: eight bands of gyb chunks, band = blockIdx.x & 7 --, y = plane
static dim3 lean_grid(const LeanArgs &A)
{
	const unsigned per_plane = A.order == 1 ? 8u * cdiv((int64_t)A.gx * A.gyb, A.wpb) : cdiv((int64_t)A.gx * A.gy, A.wpb);
	return dim3(per_plane, (unsigned)A.D);
}

bool cbca_lean_fits(int D, int H, int W, size_t plan_bytes, bool two_pass, int rb)
{
	if ((int64_t)D * H * W >= ((int64_t)1 << 32) || D > 65535) return false;   // (32-bit voxel indices in the entries; the plane is blockIdx.y)
	const LeanArgs A = lean_args(nullptr, nullptr, plan_bytes, nullptr, nullptr, D, H, W, -1, 0, rb, 0, 0, two_pass);
	return A.capd >= 2 && (int64_t)A.gx * A.gy < 65536 && (int64_t)A.gx * A.gy * D < ((int64_t)1 << 30);
}

// bytes of the plan area the two-pass records take at the product's rows per wave (small images: more than the tile kernel's plan)
size_t cbca_lean2x_bytes(int D, int H, int W)
{
	const int64_t waves = (int64_t)cdiv(W, 252) * cdiv(H, lean2x_rows(0)) * D, seconds = (int64_t)cdiv(W, 252) * 3 * D;
	return (size_t)(LH_WORDS + (waves + seconds) * L2X_REC_WORDS) * 4;
}

// once per pair and direction (before the first pass): the list of outputs whose support is not the minimal 3 x 3
int cbca_classify(const void *packed, void *plan, size_t plan_bytes, int D, int H, int W, int direction, int route, int rb, int cap_limit,
                  hipStream_t st, bool two_pass, float cost_limit)
{
	LeanArgs A = lean_args(packed, plan, plan_bytes, nullptr, nullptr, D, H, W, direction, route, rb, cap_limit, 0, two_pass);
	A.cost_limit = cost_limit > 0 ? cost_limit : MC_LEAN2X_COST;
	// the head of the area -- the list's magic and overflow word, and the head of a tile kernel's plan of an earlier pair -- is cleared ON THE
	// DEVICE under the launch condition of the classification itself: a pair whose route is the tile kernel's keeps the plan its first pass
	// wrote there (an unconditional memset between two aggregation stages invalidated it)
	hipLaunchKernelGGL(cbca_list_reset_kernel, dim3(1), dim3(64), 0, st, A);
	if (two_pass) {   // (every wave writes its own record: one launch, + the per-plane cost check)
		if (A.capd == 0) { set_error("cbca_classify: the plan area does not hold the two-pass records"); return MC_EINVAL; }
		hipLaunchKernelGGL(cbca_classify2x_kernel, lean_grid(A), dim3(256), 0, st, A);
		hipLaunchKernelGGL(cbca_list_cost_kernel, dim3((unsigned)D), dim3(256), 0, st, A, A.gx * A.gy);
		return check_launch("cbca_classify (two-pass records)");
	}
	hipLaunchKernelGGL(cbca_classify_kernel<0>, lean_grid(A), dim3(256), 0, st, A);
	hipLaunchKernelGGL(cbca_list_scan_kernel, dim3((unsigned)D), dim3(256), 0, st, A, A.gx * A.gy);
	hipLaunchKernelGGL(cbca_classify_kernel<1>, lean_grid(A), dim3(256), 0, st, A);
	return check_launch("cbca_classify");
}

// one aggregation pass: the lean kernel over every output + the listed outputs.  cfg.lean_rb: rows per wave (2 / 4 / 8; anything else:
// the default); cfg.lean_variant: bit 2 the listed outputs in a launch of their own, bit 4 one band of rows per XCD, bits 5 / 6
// non-temporal loads / stores
int cbca_lean(const void *packed, const void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
              int route, hipStream_t st, const CbcaCfg &cfg)
{
	const int variant = cfg.lean_variant >= 0 ? cfg.lean_variant : MC_LEAN_VARIANT_DEFAULT;
	const int order = (variant & 16) ? 1 : 0;
	const int rb = lean_rows(cfg.lean_rb);
	const LeanArgs A = lean_args(packed, (void *)plan, plan_bytes, vin, vout, D, H, W, direction, route, rb, cfg.nd, order);
	const dim3 blocks = lean_grid(A);
	const bool own_launch = (variant & 4) != 0;
#define MC_LEAN2_GO(P, POL) do { \
		if (own_launch) hipLaunchKernelGGL((cbca_lean_kernel<P, POL, false>), blocks, dim3(256), 0, st, A); \
		else hipLaunchKernelGGL((cbca_lean_kernel<P, POL, true>), blocks, dim3(256), 0, st, A); } while (0)
#define MC_LEAN2_POL(P) do { \
		if (pol == 3) MC_LEAN2_GO(P, 3); else if (pol == 2) MC_LEAN2_GO(P, 2); else if (pol == 1) MC_LEAN2_GO(P, 1); else MC_LEAN2_GO(P, 0); } while (0)
	const int pol = (variant >> 5) & 3;   // bit 5: non-temporal row loads, bit 6: non-temporal stores (cbca_lean_kernel)
	{
		if (rb == 2) MC_LEAN2_POL(2);
		else if (rb == 8) MC_LEAN2_POL(8);
		else MC_LEAN2_POL(4);
	}
#undef MC_LEAN2_POL
#undef MC_LEAN2_GO
	int rc = check_launch("cbca_lean");
	if (rc || !own_launch) return rc;
	hipLaunchKernelGGL(cbca_list_kernel, blocks, dim3(256), 0, st, A);
	return check_launch("cbca_list");
}

// two aggregation passes in one launch (cbca_lean2x_kernel) out of the list cbca_classify(..., two_pass = true) wrote: vout = the volume
// after the second pass.  Stands down (and writes nothing) unless the pair's route is `route` and the list is this problem's.
int cbca_lean2x(const void *packed, const void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
                int route, hipStream_t st, const CbcaCfg &cfg)
{
	LeanArgs A = lean_args(packed, (void *)plan, plan_bytes, vin, vout, D, H, W, direction, route, cfg.lean_rb, cfg.nd, 0, true);
	A.wpb = L2X_WPB;
	const dim3 blocks = lean_grid(A);
	if (A.rb == 4) hipLaunchKernelGGL((cbca_lean2x_kernel<4, L2X_WPB>), blocks, dim3(64 * L2X_WPB), 0, st, A);
	else if (A.rb == 6) hipLaunchKernelGGL((cbca_lean2x_kernel<6, L2X_WPB>), blocks, dim3(64 * L2X_WPB), 0, st, A);
	else if (A.rb == 10) hipLaunchKernelGGL((cbca_lean2x_kernel<10, L2X_WPB>), blocks, dim3(64 * L2X_WPB), 0, st, A);
	else if (A.rb == 12) hipLaunchKernelGGL((cbca_lean2x_kernel<12, L2X_WPB>), blocks, dim3(64 * L2X_WPB), 0, st, A);
	else hipLaunchKernelGGL((cbca_lean2x_kernel<8, L2X_WPB>), blocks, dim3(64 * L2X_WPB), 0, st, A);
	return check_launch("cbca_lean2x");
}

}  // namespace mc
