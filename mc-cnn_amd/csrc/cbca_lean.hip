// Cross-based cost aggregation (adcensus.cu:343-377) on pairs whose supports are nearly all the minimal 3 x 3 -- the regime
// SURVEY 8(d) specifies for the 1000 x 1500 x 256 case (Gaussian texture: 98.7 % of the outputs) -- as a bandwidth problem.
//
// Whether an output's support is the minimal 3 x 3 depends on the pair's arms and the plane only, not on the volume, and
// mc_predict aggregates a pair 2 + 16 times per direction (main.lua:998-1001, 1033-1039).  So the work is split once per
// pair and direction:
//   cbca_classify_kernel  (arms only)  LISTS every output with a partner whose support is NOT the minimal 3 x 3 (4 bytes
//                                      per entry: the voxel index) in the pair's plan area;
//   cbca_lean_kernel      (per pass)   computes the minimal 3 x 3 mean for EVERY output -- nine additions in the reference's
//                                      order and an IEEE divide, out of a three-row register window, neighbours' columns
//                                      through DPP: no arm lengths, no LDS, no tests; the plane is read once and written once;
//   cbca_list_kernel      (per pass)   re-runs the reference's loop for the listed outputs (one thread per entry, reads
//                                      through L2) and overwrites them.
// Outputs without a partner are copied through by the lean kernel (adcensus.cu:353-354).  The strip kernel (cbca.hip), which
// does all of this per pass, remains what adcensus.cbca runs on its own (no state between calls) and the fallback when the
// list does not fit.
#include "cbca_common.h"
#include <algorithm>

namespace mc {

namespace {

struct LeanArgs {
	const uint32_t *p0, *p1;     // packed arm lengths (H,W)
	const float *vin;
	float *vout;
	uint32_t *hdr;               // list header (LH_*), the entries follow it
	uint32_t cap;                // entries the list can hold
	int D, H, W, direction;
	int rb, gx, gy;              // rows per wave, strips per row, row chunks
	const uint32_t *flags;       // cbca_pack's flag words
	int route;
};

// wave -> (plane, row chunk, strip of 256 columns); strips fastest, so that the waves of a block are neighbours in a row
// (a strip's edge columns are its neighbours' lines: L1 / L2 hits)
__device__ __forceinline__ bool lean_wave(const LeanArgs &A, int wv, int &d, int &y0, int &y1, int &x0)
{
	const long long w = (long long)blockIdx.x * 4 + wv;
	const int strip = (int)(w % A.gx);
	const long long t = w / A.gx;
	const int chunk = (int)(t % A.gy);
	d = (int)(t / A.gy);
	y0 = chunk * A.rb;
	y1 = min(A.H, y0 + A.rb);
	x0 = strip * 256;
	return d < A.D;
}

}  // namespace

// ---- once per pair and direction: the outputs the lean kernel gets wrong ---------------------------------------------------
// An output (d, y, x) with a partner has the minimal support iff its combined arms (per-arm minimum of the two images,
// cbca.hip "Packed arm lengths") are all 1 and the rows above and below have left = right = 1 in its column -- the test of
// the strip kernel.  Everything else with a partner is listed.  Entries are collected per wave in LDS and appended 256 and
// more at a time (one atomic per flush).
__global__ void __launch_bounds__(256) cbca_classify_kernel(const LeanArgs A)
{
	__shared__ cb_u32 bufs[4][512];
	if (!cbca_gate(A.flags, A.route)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	cb_u32 *__restrict__ buf = bufs[wv];
	if (blockIdx.x == 0 && threadIdx.x == 0) {   // (the count and the overflow word were zeroed by the host's memset)
		A.hdr[LH_D] = (uint32_t)A.D; A.hdr[LH_H] = (uint32_t)A.H; A.hdr[LH_W] = (uint32_t)A.W;
		A.hdr[LH_DIR] = (uint32_t)(A.direction + 1); A.hdr[LH_MAGIC] = LH_MAGIC_VALUE;
	}
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, d, y0, y1, xb)) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs = xb + 4 * lane;
	const cb_u32 OOB = 0x80000000u;
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
	int cnt = 0;
	auto flush = [&]() {
		cb_u32 base = 0;
		if (lane == 0) base = atomicAdd(A.hdr + LH_COUNT, (cb_u32)cnt);
		base = (cb_u32)__builtin_amdgcn_readfirstlane((int)base);
		for (int i = lane; i < cnt; i += 64) {
			if (base + (cb_u32)i < A.cap) A.hdr[LH_WORDS + base + i] = buf[i];
			else A.hdr[LH_OVERFLOW] = 1u;   // (the list does not fit: the passes fall back to the strip kernel)
		}
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
			const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((cb_u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((cb_u32)bal, 0));
			if (listed) buf[pos] = (cb_u32)d * (cb_u32)HWi + (cb_u32)(y * W + xs + j);
			cnt += __builtin_popcountll(bal);
		}
		if (cnt >= 256) flush();   // (a row adds at most 256 entries: 512 always hold them)
#pragma unroll
		for (int j = 0; j < 4; ++j) { ma[j] = mb[j]; mb[j] = mc_[j]; mc_[j] = md[j]; }
	}
	if (cnt) flush();
}

// ---- per pass: the minimal 3 x 3 mean for every output -------------------------------------------------------------------
// One wave owns a plane x 256 columns x rb rows and walks them top to bottom.  A lane holds the four columns 4 lane .. 4 lane + 3
// of three consecutive rows (one aligned 16-byte load per row, PF rows in flight); the columns to the left and right come from
// the neighbouring lanes through DPP wave shifts, the strip's two outer columns from one extra 4-byte load in lanes 0 and 63
// (lines the neighbouring strips read anyway).  Nine additions per output in the reference's order -- rows ascending, x
// ascending, accumulator starting at +0.0 (adcensus.cu:356-373) -- and an IEEE divide by 9; outputs without a partner are
// copied through.  What is wrong afterwards -- outputs with another support, among them every output on the image border,
// where the window reads zeros -- is exactly what cbca_classify_kernel listed.
template <int PF, bool NT>
__global__ void __launch_bounds__(256) cbca_lean_kernel(const LeanArgs A)
{
	static_assert(PF % 3 == 0, "the neighbour columns of the three-row window rotate by renaming");
	constexpr int AUX = NT ? 2 : 0;   // volumes far beyond the 256 MB Infinity Cache are streamed (cbca_strip_kernel)
	if (!cbca_gate(A.flags, A.route) || !list_valid(A.hdr, A.D, A.H, A.W, A.direction)) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int d, y0, y1, xb;
	if (!lean_wave(A, wv, d, y0, y1, xb)) return;
	const int H = A.H, W = A.W;
	const int HWi = H * W;
	const int sh = d * A.direction;
	const int xs = xb + 4 * lane;
	const cb_u32 OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const bool lane_in = xs < W;                                    // (lanes of the last strip beyond the image: nothing to read or write)
	const int ecol = lane == 0 ? xs - 1 : (lane == 63 ? xs + 4 : -1);   // the strip's outer columns
	const bool eok = ecol >= 0 && ecol < W;
	cb_u32 inr = 0;   // bit j: the output has a partner (adcensus.cu:353)
#pragma unroll
	for (int j = 0; j < 4; ++j)
		if (xs + j + sh >= 0 && xs + j + sh < W) inr |= 1u << j;

	struct Stage { cb_u4 v; cb_u32 e; };
	// No branch around the loads (hipcc's wait counts stay exact: cbca_tile.hip): one 16-byte load wherever the unit starts.  Where W is
	// not a multiple of 4 the row's last unit ends in the next row's first columns (or, behind the plane, in nothing): those words are
	// "columns >= W", which only the border output x = W - 1 would use -- and that one is listed.  Rows outside the image: zeros.
	auto fetch = [&](Stage &st, int r) {
		const bool rok = (unsigned)r < (unsigned)H;
		st.v = __builtin_amdgcn_raw_buffer_load_b128(rv, (rok & lane_in) ? (cb_u32)(r * W + xs) * 4u : OOB, 0, AUX);
		st.e = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok & eok) ? (cb_u32)(r * W + ecol) * 4u : OOB, 0, 0);
	};
	// Row r lives in ONE register set from its request to the last output row that uses it: PF rows in flight + the three-row
	// window = U sets, the row loop unrolled U times, so that a set is requested at one place of the loop body and nowhere else
	// (with fewer sets, or a second request site before the loop, hipcc places copies at the back edge that wait for every load in
	// flight).  The loop therefore starts PF rows early on zeroed sets: those rows request rows ra .. ra + PF - 1 and store nothing.
	constexpr int U = PF + 3;
	Stage st[U];
	float nl[3], nr[3];   // the window rows' columns xs - 1 and xs + 4
#pragma unroll
	for (int u = 0; u < U; ++u) { st[u].v = cb_u4{0u, 0u, 0u, 0u}; st[u].e = 0u; }
	// (no branch anywhere in the row loop -- not around the divide of the outputs with a partner, not around the store of the
	// staged rows that complete no output row of this wave: hipcc's wait counts then count the rows in flight exactly)
	auto output = [&](int yo, const Stage &a, const Stage &b, const Stage &c, int ia, int ib, int ic) {
		const float ra_[6] = {nl[ia], __uint_as_float(a.v.x), __uint_as_float(a.v.y), __uint_as_float(a.v.z), __uint_as_float(a.v.w), nr[ia]};
		const float rb_[6] = {nl[ib], __uint_as_float(b.v.x), __uint_as_float(b.v.y), __uint_as_float(b.v.z), __uint_as_float(b.v.w), nr[ib]};
		const float rc_[6] = {nl[ic], __uint_as_float(c.v.x), __uint_as_float(c.v.y), __uint_as_float(c.v.z), __uint_as_float(c.v.w), nr[ic]};
		float res[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			float sum = 0;
			sum += ra_[j]; sum += ra_[j + 1]; sum += ra_[j + 2];
			sum += rb_[j]; sum += rb_[j + 1]; sum += rb_[j + 2];
			sum += rc_[j]; sum += rc_[j + 1]; sum += rc_[j + 2];
			float q = sum / 9.0f;
			asm volatile("" : "+v"(q));   // (computed for every lane: otherwise the divide becomes a conditional block)
			res[j] = ((inr >> j) & 1u) ? q : rb_[j + 1];
		}
		const bool mine = (yo >= y0) & (yo < y1) & lane_in;
		// stored through a descriptor that ends with the row: the words of a last unit beyond the image are dropped by the range check
		const __amdgpu_buffer_rsrc_t rrow = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, (yo + 1) * W * 4, 0x00020000);
		__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])},
		                                       rrow, mine ? (cb_u32)(yo * W + xs) * 4u : OOB, 0, AUX);
	};
	const int ra = y0 - 1;   // first staged row; rows ra .. y1 are staged, row r completes the window of output row r - 1
	for (int g = ra - PF; g <= y1; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int r = g + u;
			const float e = __uint_as_float(st[u].e);
			nl[u % 3] = lane_from_below(__uint_as_float(st[u].v.w), e);   // lane 0: the strip's left outer column
			nr[u % 3] = lane_from_above(__uint_as_float(st[u].v.x), e);   // lane 63: its right outer column
			output(r - 1, st[(u + U - 2) % U], st[(u + U - 1) % U], st[u], (u + 1) % 3, (u + 2) % 3, u % 3);
			fetch(st[(u + PF) % U], r + PF <= y1 ? r + PF : -1);   // (the set of row r - 3)
			__builtin_amdgcn_sched_barrier(0);   // (rows stay in program order: hoisted additions of later rows would wait for their loads early)
		}
	}
}

// ---- per pass: the listed outputs, one thread per entry: the reference's loop (rows ascending, x ascending, one accumulator) ----
__global__ void __launch_bounds__(256) cbca_list_kernel(const LeanArgs A)
{
	if (!cbca_gate(A.flags, A.route) || !list_valid(A.hdr, A.D, A.H, A.W, A.direction)) return;
	const uint32_t n = min(A.hdr[LH_COUNT], A.cap);
	const uint32_t W = (uint32_t)A.W, HW = (uint32_t)A.H * (uint32_t)A.W;
	const uint32_t *__restrict__ ent = A.hdr + LH_WORDS;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const uint32_t idx = ent[i];
		const uint32_t d = idx / HW, rem = idx - d * HW;
		const int y = (int)(rem / W), x = (int)(rem - (uint32_t)y * W);
		const int sh = (int)d * A.direction;
		const float *__restrict__ vd = A.vin + (size_t)d * HW;
		const uint32_t mm = bytemin4(A.p0[y * (int)W + x], A.p1[y * (int)W + x + sh]);
		const int u = (int)((mm >> 16) & 0xffu), dn = (int)(mm >> 24);
		float sum = 0;
		int cnt = 0;
		for (int q = y - u; q <= y + dn; ++q) {
			const int g = q * (int)W + x;
			const uint32_t m = bytemin4(A.p0[g], A.p1[g + sh]);
			const int l = (int)(m & 0xffu), nn = l + (int)((m >> 8) & 0xffu) + 1;
			const float *__restrict__ row = vd + g - l;
			for (int k = 0; k < nn; ++k) sum += row[k];
			cnt += nn;
		}
		A.vout[idx] = sum / (float)cnt;
	}
}

static LeanArgs lean_args(const void *packed, void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
                          int route, int rb)
{
	LeanArgs A;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = vin; A.vout = vout;
	A.hdr = (uint32_t *)plan;
	A.cap = (uint32_t)std::min<size_t>((plan_bytes - LH_WORDS * 4) / 4, 0xfffffff0u);
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.gx = (int)cdiv(W, 256);
	// rows per wave: 2 / rb of the rows are read twice; at least ~12 K waves
	const int64_t gy_min = cdiv((int64_t)12288, (int64_t)A.gx * D);
	A.rb = rb > 0 ? rb : (int)std::min<int64_t>(128, std::max<int64_t>(16, cdiv((int64_t)H, gy_min)));
	A.gy = (int)cdiv(H, A.rb);
	A.flags = cs.flag;
	A.route = route;
	return A;
}

bool cbca_lean_fits(int D, int H, int W, size_t plan_bytes)
{
	return (int64_t)D * H * W < ((int64_t)1 << 32) && plan_bytes > (size_t)LH_WORDS * 4;
}

// once per pair and direction (before the first pass): the list of outputs whose support is not the minimal 3 x 3
int cbca_classify(const void *packed, void *plan, size_t plan_bytes, int D, int H, int W, int direction, int route, hipStream_t st)
{
	const LeanArgs A = lean_args(packed, plan, plan_bytes, nullptr, nullptr, D, H, W, direction, route, 0);
	const hipError_t e = hipMemsetAsync(plan, 0, LH_WORDS * 4, st);
	if (e != hipSuccess) {
		set_error("cbca_classify: %s", hipGetErrorString(e));
		return (int)e;
	}
	const int64_t waves = (int64_t)A.gx * A.gy * D;
	hipLaunchKernelGGL(cbca_classify_kernel, dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	return check_launch("cbca_classify");
}

// one aggregation pass: the lean kernel over every output, then the listed outputs
int cbca_lean(const void *packed, const void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
              int route, hipStream_t st, const CbcaCfg &cfg)
{
	const LeanArgs A = lean_args(packed, (void *)plan, plan_bytes, vin, vout, D, H, W, direction, route, cfg.rb);
	const int64_t waves = (int64_t)A.gx * A.gy * D;
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)D * H * W * 4 > ((int64_t)768 << 20);
	const unsigned blocks = (unsigned)cdiv(waves, 4);
	if (cfg.variant == 1) {
		if (nt) hipLaunchKernelGGL((cbca_lean_kernel<3, true>), dim3(blocks), dim3(256), 0, st, A);
		else hipLaunchKernelGGL((cbca_lean_kernel<3, false>), dim3(blocks), dim3(256), 0, st, A);
	} else if (cfg.variant == 2) {
		if (nt) hipLaunchKernelGGL((cbca_lean_kernel<9, true>), dim3(blocks), dim3(256), 0, st, A);
		else hipLaunchKernelGGL((cbca_lean_kernel<9, false>), dim3(blocks), dim3(256), 0, st, A);
	} else {
		if (nt) hipLaunchKernelGGL((cbca_lean_kernel<6, true>), dim3(blocks), dim3(256), 0, st, A);
		else hipLaunchKernelGGL((cbca_lean_kernel<6, false>), dim3(blocks), dim3(256), 0, st, A);
	}
	int rc = check_launch("cbca_lean");
	if (rc) return rc;
	hipLaunchKernelGGL(cbca_list_kernel, dim3(2048), dim3(256), 0, st, A);
	return check_launch("cbca_list");
}

}  // namespace mc
