// Cross-based cost aggregation, TWO iterations per pass over the volume (gfx950).
//
// adcensus.cbca (adcensus.cu:343-377) is applied 2 + 16 times per volume on the accurate Middlebury configuration
// (main.lua:132-135, 998-1001, 1033-1039) and each application is a pure function of the previous volume: two
// consecutive applications are fused into one launch that reads V_k and writes V_(k+2), so the intermediate V_(k+1)
// never reaches HBM -- 2 V of traffic for 4 V of algorithmic bytes.  Every voxel of V_(k+2) is the reference's
// expression evaluated on the reference's V_(k+1) values (same supports, same order of additions, IEEE divide or its
// proven three-operation equivalent for the count 9), so results stay bit-identical.
//
// Decomposition as cbca_strip2_kernel: one wave = one disparity plane x a strip of 256 staged columns x RB output rows,
// walked top to bottom with no block barrier.  Per new input row r the wave
//   1. commits V_k row r (registers: column-pair window; LDS: value ring + byte-minimum arm-length ring),
//   2. evaluates the minimal 3x3 support of stage 1 for row r-1 out of registers -> V_(k+1) ring (LDS),
//   3. re-evaluates the stage-1 outputs of row r-2 that have a larger support (compacted; window form over ring rows
//      r-4..r, else the reference's loop out of the ring / global V_k) and patches the V_(k+1) ring: row r-2 is final,
//   4. evaluates the minimal support of stage 2 for row r-3 out of the (final) V_(k+1) rows r-4..r-2 in registers,
//   5. re-evaluates the stage-2 outputs of row r-4 with a larger support from the V_(k+1) ring rows r-6..r-2; a support
//      that leaves those rows or the columns this wave owns is rebuilt point by point from global V_k (nested
//      reference loops: rare on textured images, see the density gate in cbca_fused_pairs), and stores row r-4.
// V_(k+1) is valid in frame columns 4..251, V_(k+2) is produced for frame columns 8..247 (240 per strip).
#include "cbca_common.h"
#include <algorithm>

namespace mc {

constexpr int F2_STEP = 240;   // output columns per strip (frame columns 8 .. 247)
constexpr int F2_HALO = 8;
constexpr int F2_RING = 8;     // rows per LDS ring (values of V_k, values of V_(k+1), minimum arm lengths)
constexpr int F2_WR = 2;       // column radius of the window form

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

template <bool NT, int ABL>
__global__ void __launch_bounds__(128) cbca_fused2_kernel(const CbcaArgs A)
{
	constexpr int PF = 3;                      // rows in flight = rows of the register windows: the loop is unrolled by 3
	constexpr int VOL_AUX = NT ? 2 : 0;
	constexpr int RM = F2_RING - 1;
	__shared__ float V0ring[2][F2_RING * CS_COLS];
	__shared__ float V1ring[2][F2_RING * CS_COLS];
	__shared__ cb_u32 Mring[2][F2_RING * CS_COLS];
	__shared__ float Rrow[2][CS_COLS];
	__shared__ unsigned short Clist[2][CS_COLS];
	if (A.overflow && *A.overflow) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float *__restrict__ V0 = V0ring[wv];
	float *__restrict__ V1 = V1ring[wv];
	cb_u32 *__restrict__ M = Mring[wv];
	float *__restrict__ R = Rrow[wv];
	unsigned short *__restrict__ CL = Clist[wv];
	const int H = A.H, W = A.W, direction = A.direction;
	const int HWi = H * W;
	// wave -> (region, d): two consecutive planes per block, the blocks of an XCD walk all plane pairs of a region
	const int dgroups = (A.nd + 1) >> 1;
	const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
	const int region = (kb / dgroups) * 8 + xcd;
	const int d = A.d0 + (kb % dgroups) * 2 + wv;
	if (region >= A.gx * A.gy || d >= A.d0 + A.nd) return;
	const int cx = region % A.gx, cy = region / A.gx;
	const int sh = d * direction;
	const int xs = cx * F2_STEP - F2_HALO + 4 * lane;   // image column of this lane's first column
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const int ra = y0 - 4;                              // first staged row
	const int plane_bytes = HWi * 4;
	const cb_u32 OOB = 0x80000000u;
	const float *__restrict__ plane_in = A.vin + (size_t)d * HWi;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)plane_in, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	const bool full_in = xs >= 0 && xs + 3 < W;
	const bool has1 = lane >= 1 && lane <= 62;          // lanes that own valid stage-1 (intermediate) columns
	const bool has2 = lane >= 2 && lane <= 61;          // lanes that own output columns
	const bool full_out = has2 && xs + 3 < W;
	const bool any_out = has2 && xs < W;
	// per column: exists / its shifted partner is inside the image (adcensus.cu:353-354) -- lane masks
	bool valid1[4], valid2[4], inr[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		valid1[j] = has1 && x >= 0 && x < W;
		valid2[j] = has2 && x < W;
		inr[j] = x + sh >= 0 && x + sh < W;
	}

	struct Stage { cb_u4 v, a, b; };
	auto fetch = [&](Stage &st, int r) {  // row r of the plane -> registers (rows outside the image: zeros)
		const bool rok = r >= 0 && r < H;
		const int base = r * W + xs;
		if (full_in) {
			st.v = __builtin_amdgcn_raw_buffer_load_b128(rv, rok ? (cb_u32)base * 4u : OOB, 0, VOL_AUX);
		} else {  // image edges: per column
			cb_u32 t[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok && xs + k >= 0 && xs + k < W) ? (cb_u32)(base + k) * 4u : OOB, 0, 0);
			st.v = cb_u4{t[0], t[1], t[2], t[3]};
		}
		st.a = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
		st.b = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
	};

	auto make_row = [&](C2Row &w, float v0, float v1, float v2, float v3) {
		const float l3 = lane_from_below(v3, 0.0f), r0 = lane_from_above(v0, 0.0f);
		w.A = cb_f2{l3, v0}; w.B = cb_f2{v0, v1}; w.C = cb_f2{v1, v2}; w.D = cb_f2{v2, v3}; w.E = cb_f2{v3, r0};
	};

	// minimal 3x3 sums of a lane's four columns (two packed chains in the reference's order), divided by 9; columns whose
	// partner lies outside the image are copied through
	auto skeleton = [&](const C2Row &up, const C2Row &own, const C2Row &dn_, const bool (&use)[4], float (&res)[4]) {
		cb_f2 s01 = cb_f2{0.0f, 0.0f}, s23 = cb_f2{0.0f, 0.0f};
		s01 += up.A; s01 += up.B; s01 += up.C;
		s23 += up.C; s23 += up.D; s23 += up.E;
		s01 += own.A; s01 += own.B; s01 += own.C;
		s23 += own.C; s23 += own.D; s23 += own.E;
		s01 += dn_.A; s01 += dn_.B; s01 += dn_.C;
		s23 += dn_.C; s23 += dn_.D; s23 += dn_.E;
		const cb_f2 q01 = div9_pk(s01), q23 = div9_pk(s23);
		res[0] = q01.x; res[1] = q01.y; res[2] = q23.x; res[3] = q23.y;
		const float sums[4] = {s01.x, s01.y, s23.x, s23.y};
		const float ownv[4] = {own.B.x, own.B.y, own.D.x, own.D.y};
		bool odd = false;
#pragma unroll
		for (int j = 0; j < 4; ++j) odd = odd || (use[j] && !div9_in_range(sums[j]));
		if (__any(odd)) {  // a sum outside the range the packed form is proven for (zero, tiny, huge, inf, nan): IEEE divide
#pragma unroll
			for (int j = 0; j < 4; ++j) res[j] = sums[j] / 9.0f;
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) res[j] = inr[j] ? res[j] : ownv[j];   // adcensus.cu:353-354: copied through
	};

	auto compact = [&](const bool (&nj)[4]) -> int {
		int n = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint64_t bal = __ballot(nj[j]);
			const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((cb_u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((cb_u32)bal, 0));
			if (nj[j]) CL[pos] = (unsigned short)(4 * lane + j);
			n += __builtin_popcountll(bal);
		}
		return n;
	};

	// Window form shared by both stages: the support of the output in frame column c of row yo lies inside rows
	// yo-2 .. yo+2 of the given value ring and columns c-WR .. c+WR: every tap of the window is read in one batch, the taps
	// outside the support add -0.0f (x + -0.0f == x exactly), rows ascending and x ascending as in the reference.
	// Returns false when the support does not fit (nothing is written then).
	auto window = [&](const float *__restrict__ VR, int yo, int c, int lo_row, int hi_row, int clo, int chi, float &out) -> bool {
		cb_u32 mm[5];
		float tv[5][2 * F2_WR + 1];
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const int ro_ = ((yo + k - 2) & RM) * CS_COLS + c;
			mm[k] = M[ro_];
#pragma unroll
			for (int t = 0; t < 2 * F2_WR + 1; ++t) tv[k][t] = VR[ro_ + t - F2_WR];
		}
		const int u = (int)((mm[2] >> 16) & 0xff), dn = (int)(mm[2] >> 24);
		bool ok = u <= 2 && dn <= 2 && yo - u >= lo_row && yo + dn <= hi_row;
		float sum = 0;
		int cnt = 0;
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const int rel = k - 2;
			const bool act = rel >= -u && rel <= dn;
			const int l = (int)(mm[k] & 0xff), rg = (int)((mm[k] >> 8) & 0xff);
			ok = ok && (!act || (l <= F2_WR && rg <= F2_WR && c - l >= clo && c + rg <= chi));
			const int la = act ? l : -1, rga = act ? rg : -1;
#pragma unroll
			for (int t = 0; t < 2 * F2_WR + 1; ++t) {
				const int dx = t - F2_WR;
				const bool in = dx < 0 ? la >= -dx : (dx == 0 ? act : rga >= dx);
				sum += in ? tv[k][t] : -0.0f;
			}
			cnt += act ? l + rg + 1 : 0;
		}
		out = sum / (float)cnt;
		return ok;
	};

	// stage 1, any support: rows inside the rings from LDS, the rest from global V_k (always available)
	auto general1 = [&](int yo, int c, int lo_row, int hi_row) -> float {
		const int x = cx * F2_STEP - F2_HALO + c;
		const cb_u32 own = M[(yo & RM) * CS_COLS + c];
		const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
		float sum = 0;
		int cnt = 0;
		for (int q = yo - u; q <= yo + dn; ++q) {
			const bool row_in = q >= lo_row && q <= hi_row;
			const int rowo = (q & RM) * CS_COLS;
			cb_u32 mm;
			if (row_in) mm = M[rowo + c];
			else {
				const int g = q * W + x;
				mm = bytemin4(A.p0[g], A.p1[g + sh]);
			}
			const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
			const int n = l + rg + 1;
			if (row_in && c - l >= 0 && c + rg < CS_COLS) {
				const float *row = V0 + rowo + c - l;
				for (int k = 0; k < n; ++k) sum += row[k];
			} else {
				const float *row = plane_in + q * W + x - l;
				for (int k = 0; k < n; ++k) sum += row[k];
			}
			cnt += n;
		}
		return sum / (float)cnt;
	};

	// stage 2, any support: V_(k+1) values inside the final ring rows / owned columns from LDS, every other one rebuilt
	// from global V_k by the reference's loop (cbca_point_global) -- nested, exact, meant to be rare
	auto general2 = [&](int yo, int c, int lo1, int hi1, int lom, int him) -> float {
		const int x = cx * F2_STEP - F2_HALO + c;
		const cb_u32 own = M[(yo & RM) * CS_COLS + c];
		const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
		float sum = 0;
		int cnt = 0;
		for (int q = yo - u; q <= yo + dn; ++q) {
			const int rowo = (q & RM) * CS_COLS;
			cb_u32 mm;
			if (q >= lom && q <= him) mm = M[rowo + c];
			else {
				const int g = q * W + x;
				mm = bytemin4(A.p0[g], A.p1[g + sh]);
			}
			const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
			const bool row_in = q >= lo1 && q <= hi1;
			for (int t = -l; t <= rg; ++t) {
				const int cc = c + t;
				float v;
				if (row_in && cc >= 4 && cc <= 251) v = V1[rowo + cc];
				else v = cbca_point_global(A.p0, A.p1, plane_in, W, sh, q, x + t);
				sum += v;
			}
			cnt += l + rg + 1;
		}
		return sum / (float)cnt;
	};

	// lane masks "this column's support is not the minimal 3x3" for the rows in flight.  accA / accB accumulate
	// lr(r-1) | all(r) | lr(r+1) as rows arrive; need1..need3 delay the finished masks until the later stages use them.
	bool accA[4] = {true, true, true, true}, accB[4] = {true, true, true, true};
	bool needB[4] = {true, true, true, true};   // row r-2 (stage-1 patch pass)
	bool needC[4] = {true, true, true, true};   // row r-3 (stage-2 skeleton)
	bool needE[4] = {true, true, true, true};   // row r-4 (stage-2 patch pass)
	float held[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // stage-2 skeleton results of row r-4 (computed one iteration earlier)

	Stage st[PF];
	C2Row w0[PF], w1[PF];
#pragma unroll
	for (int u = 0; u < PF; ++u) {
		fetch(st[u], ra + u);
		w0[u].A = w0[u].B = w0[u].C = w0[u].D = w0[u].E = cb_f2{0.0f, 0.0f};
		w1[u] = w0[u];
	}
	const int last = y1 - 1 + 4;
	for (int g = ra; g <= last; g += PF) {
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const int r = g + u;
			// ---- 1. commit V_k row r ----
			bool needA[4];   // row r-1
			{
				const int o = (r & RM) * CS_COLS + 4 * lane;
				*(cb_f4 *)(V0 + o) = cb_f4{__uint_as_float(st[u].v.x), __uint_as_float(st[u].v.y), __uint_as_float(st[u].v.z), __uint_as_float(st[u].v.w)};
				const cb_u4 m = bytemin4x4_sdwa(st[u].a, st[u].b);
				*(cb_u4 *)(M + o) = m;
				make_row(w0[u], __uint_as_float(st[u].v.x), __uint_as_float(st[u].v.y), __uint_as_float(st[u].v.z), __uint_as_float(st[u].v.w));
				const cb_u32 mj[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					const bool all_ne = mj[j] != 0x01010101u;
					const bool lr_ne = (mj[j] & 0xffffu) != 0x0101u;
					needA[j] = accA[j] || lr_ne;     // row r-1 complete: lr(r-2) | all(r-1) | lr(r)
					accA[j] = accB[j] || all_ne;
					accB[j] = lr_ne;
				}
			}
			fetch(st[u], r + PF);
			// ---- 2. stage 1, minimal supports of row r-1 -> V_(k+1) ring ----
			{
				float res[4];
				bool use[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) use[j] = !needA[j] && inr[j] && valid1[j];
				skeleton(w0[(u + 1) % PF], w0[(u + 2) % PF], w0[u], use, res);
				*(cb_f4 *)(V1 + ((r - 1) & RM) * CS_COLS + 4 * lane) = cb_f4{res[0], res[1], res[2], res[3]};
			}
			// ---- 3. stage 1, larger supports of row r-2: patch the V_(k+1) ring ----
			{
				const int yb = r - 2;
				bool nj[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) nj[j] = needB[j] && inr[j] && valid1[j];
				if (!(ABL & 1) && yb >= 0 && yb < H && yb >= ra + 1 && __any(nj[0] || nj[1] || nj[2] || nj[3])) {
					const int n = compact(nj);
					const int lo_row = max(max(ra, 0), r - RM), hi_row = min(H - 1, r);
					for (int e0 = 0; e0 < n; e0 += 64) {
						const int e = e0 + lane;
						if (e < n) {
							const int c = (int)CL[e];
							float v;
							if (!window(V0, yb, c, lo_row, hi_row, 0, CS_COLS - 1, v)) v = general1(yb, c, lo_row, hi_row);
							V1[(yb & RM) * CS_COLS + c] = v;
						}
					}
				}
				// row r-2 of V_(k+1) is final: the lane's own columns and the two neighbour columns into registers
				const cb_f4 f = *(const cb_f4 *)(V1 + (yb & RM) * CS_COLS + 4 * lane);
				make_row(w1[(u + 1) % PF], f.x, f.y, f.z, f.w);
			}
			// ---- 4. stage 2, minimal supports of row r-3 (registers) ----
			float res2[4];
			{
				bool use[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) use[j] = !needC[j] && inr[j] && valid2[j];
				skeleton(w1[(u + 2) % PF], w1[u], w1[(u + 1) % PF], use, res2);   // rows r-4, r-3, r-2
			}
			// ---- 5. stage 2, larger supports of row r-4, then store it ----
			{
				const int ye = r - 4;
				if (ye >= y0 && ye < y1) {
					float res[4] = {held[0], held[1], held[2], held[3]};
					bool nj[4];
#pragma unroll
					for (int j = 0; j < 4; ++j) nj[j] = needE[j] && inr[j] && valid2[j];
					if (!(ABL & 2) && __any(nj[0] || nj[1] || nj[2] || nj[3])) {
						const int n = compact(nj);
						*(cb_f4 *)(R + 4 * lane) = cb_f4{res[0], res[1], res[2], res[3]};
						// final V_(k+1) rows in the ring: the first one computed is ra+1, the newest final one is r-2
						const int lo1 = max(max(ra + 1, 0), r - 1 - RM), hi1 = min(H - 1, r - 2);
						const int lom = max(max(ra, 0), r - RM), him = min(H - 1, r);
						for (int e0 = 0; e0 < n; e0 += 64) {
							const int e = e0 + lane;
							if (e < n) {
								const int c = (int)CL[e];
								float v;
								if (!window(V1, ye, c, lo1, hi1, 4, 251, v)) v = general2(ye, c, lo1, hi1, lom, him);
								R[c] = v;
							}
						}
						const cb_f4 rr = *(const cb_f4 *)(R + 4 * lane);
						res[0] = rr.x; res[1] = rr.y; res[2] = rr.z; res[3] = rr.w;
					}
					const int ob = ye * W + xs;
					if (full_out) {
						__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])},
						                                       ro, (cb_u32)ob * 4u, 0, VOL_AUX);
					} else if (any_out) {
#pragma unroll
						for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[j]), ro, xs + j < W ? (cb_u32)(ob + j) * 4u : OOB, 0, 0);
					}
				}
			}
			// ---- rotate the delay lines ----
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				held[j] = res2[j];
				needE[j] = needC[j];
				needC[j] = needB[j];
				needB[j] = needA[j];
			}
		}
	}
}

// Two cbca iterations in one pass: vin = V_k, vout = V_(k+2); same scratch (packed arm lengths) as cbca_strips.
int cbca_fused2(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int max_arm, hipStream_t st,
                const CbcaCfg &cfg)
{
	const int d0 = cfg.nd > 0 ? cfg.d0 : 0, nd = cfg.nd > 0 ? cfg.nd : D;
	CbcaArgs A;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = vin; A.vout = vout;
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.d0 = d0; A.nd = nd;
	A.overflow = max_arm < 0 ? cs.flag : nullptr;
	A.gx = (int)cdiv(W, F2_STEP);
	// rows per strip: 8 halo rows per chunk (4 above, 4 below) -> 64 rows unless that leaves fewer than ~16 K waves
	const int64_t gy_min = cdiv((int64_t)16384, (int64_t)A.gx * nd);
	const int rb_auto = (int)std::min<int64_t>(64, std::max<int64_t>(24, cdiv((int64_t)H, gy_min)));
	A.rb = cfg.rb > 0 ? cfg.rb : rb_auto;
	A.gy = (int)cdiv(H, A.rb);
	const int64_t waves = (int64_t)cdiv((int64_t)A.gx * A.gy, 8) * 8 * cdiv(nd, 2) * 2;
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)nd * H * W * 4 > ((int64_t)768 << 20);
	const dim3 grid((unsigned)cdiv(waves, 2));
	if (nt) hipLaunchKernelGGL((cbca_fused2_kernel<true, 0>), grid, dim3(128), 0, st, A);
	else hipLaunchKernelGGL((cbca_fused2_kernel<false, 0>), grid, dim3(128), 0, st, A);
	return check_launch("cbca_fused2");
}

}  // namespace mc
