// Cross-based cost aggregation (adcensus.cu:343-377), one iteration per launch: instruction-lean strip kernel (gfx950).
//
// Same decomposition as cbca_strip_kernel (one wave = one disparity plane x a strip of 256 staged columns x RB output
// rows walked top to bottom, four consecutive planes per block, no block barrier) and the same arithmetic per output
// (rows ascending, x ascending, one fp32 accumulator, IEEE divide or its proven equivalent for the count 9).  What
// differs is how few instructions a row costs -- the strip kernels were issue-bound, not bandwidth-bound:
//   * minimal 3x3 supports (the common case on textured images) are summed out of REGISTERS: a lane keeps its four
//     columns of the last three rows as the five adjacent column pairs, two outputs share one v_pk_add_f32 chain;
//   * "support is not minimal" is a lane mask in SGPRs (v_cmp on the byte-minimum arm lengths, scalar logic);
//   * a non-minimal output is re-evaluated by the lane that OWNS it -- no compaction list, no result row in LDS: each
//     lane picks its first flagged column, reads the 5 x 5 window of values and 5 length words around it from the
//     wave's LDS rings with immediate offsets (the row loop is unrolled by the ring size, so ring slots are static),
//     adds the taps inside the support in the reference's order (the others add -0.0f) and keeps the result in its
//     own register; a second pass runs only if some lane owns two flagged columns of the row;
//   * a support that does not fit rows y-2..y+2 / columns x-2..x+2 (rare: ~1e-5 on textured images) takes the
//     reference's loop over the ring rows or global memory.
#include "cbca_common.h"
#include <algorithm>

namespace mc {

constexpr int L3_STEP = 248;   // output columns per strip (frame columns 4 .. 251)
constexpr int L3_HALO = 4;
constexpr int L3_RING = 6;     // rows per LDS ring = unroll factor of the row loop (multiple of 3: the register window)
constexpr int L3_LAG = 2;      // output row = newest committed row - 2 (the window form reaches two rows down)

struct L3Lds {
	float V[4][L3_RING][CS_COLS];
	cb_u32 M[4][L3_RING][CS_COLS];
};

template <bool NT>
__global__ void __launch_bounds__(256) cbca_lean_kernel(const CbcaArgs A)
{
	constexpr int VOL_AUX = NT ? 2 : 0;
	__shared__ L3Lds S;
	if (A.overflow && *A.overflow) return;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	float (*__restrict__ V)[CS_COLS] = S.V[wv];
	cb_u32 (*__restrict__ M)[CS_COLS] = S.M[wv];
	const int H = A.H, W = A.W, direction = A.direction;
	const int HWi = H * W;
	const int dgroups = (A.nd + 3) >> 2;
	const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
	const int region = (kb / dgroups) * 8 + xcd;
	const int d = A.d0 + (kb % dgroups) * 4 + wv;
	if (region >= A.gx * A.gy || d >= A.d0 + A.nd) return;
	const int cx = region % A.gx, cy = region / A.gx;
	const int sh = d * direction;
	const int xs0 = cx * L3_STEP - L3_HALO;             // image column of frame column 0 (wave-uniform)
	const int xs = xs0 + 4 * lane;                      // image column of this lane's first column
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const int ra = y0 - L3_LAG - 2;                     // first staged row: the window of output row y0 starts two rows above it
	const int plane_bytes = HWi * 4;
	const cb_u32 OOB = 0x80000000u;
	const float *__restrict__ plane_in = A.vin + (size_t)d * HWi;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)plane_in, 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	// strips whose 256 staged columns all lie inside the image take dwordx4 loads / stores without per-column tests
	const bool interior = xs0 >= 0 && xs0 + CS_COLS <= W;   // wave-uniform
	const bool full_in = xs >= 0 && xs + 3 < W;
	const bool has_out = lane >= 1 && lane <= 62;
	const bool full_out = has_out && xs + 3 < W;
	const bool any_out = has_out && xs < W;
	bool valid[4], inr[4];
	bool all_inr = true;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		valid[j] = has_out && x < W;
		inr[j] = x + sh >= 0 && x + sh < W;
		all_inr = all_inr && (inr[j] || !valid[j]);
	}
	const bool wave_all_inr = !__any(!all_inr);         // wave-uniform: no output of this strip is copied through
	const int lane16 = lane * 16;
	const int soff_a = CS_PAD * 4, soff_b = (sh + CS_PAD) * 4;

	struct Stage { cb_u4 v, a, b; };
	auto fetch = [&](Stage &st, int r) {  // row r of the plane -> registers (rows outside the image: zeros)
		const bool rok = r >= 0 && r < H;
		const int rowoff = (r * W + xs0) * 4;            // scalar
		const cb_u32 vo = rok ? (cb_u32)(rowoff + lane16) : OOB;
		if (interior) {
			st.v = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, VOL_AUX);
		} else if (full_in) {
			st.v = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, VOL_AUX);
		} else {  // image edges: per column
			cb_u32 t[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok && xs + k >= 0 && xs + k < W) ? vo + 4u * k : OOB, 0, 0);
			st.v = cb_u4{t[0], t[1], t[2], t[3]};
		}
		st.a = __builtin_amdgcn_raw_buffer_load_b128(rp0, vo, soff_a, 0);
		st.b = __builtin_amdgcn_raw_buffer_load_b128(rp1, vo, soff_b, 0);
	};

	// the reference's loop for the output in frame column c of row yo (any support): ring rows from LDS, the rest global
	auto general = [&](int yo, int c, int lo_row, int hi_row) -> float {
		const int x = xs0 + c;
		const int g0 = yo * W + x;
		const cb_u32 own = bytemin4(A.p0[g0], A.p1[g0 + sh]);
		const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
		float sum = 0;
		int cnt = 0;
		for (int q = yo - u; q <= yo + dn; ++q) {
			const bool row_in = q >= lo_row && q <= hi_row;
			const int slot = (int)((unsigned)(q - ra) % (unsigned)L3_RING);
			cb_u32 mm;
			if (row_in) mm = M[slot][c];
			else {
				const int g = q * W + x;
				mm = bytemin4(A.p0[g], A.p1[g + sh]);
			}
			const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
			const int n = l + rg + 1;
			if (row_in && c - l >= 0 && c + rg < CS_COLS) {
				const float *row = &V[slot][c - l];
				for (int k = 0; k < n; ++k) sum += row[k];
			} else {
				const float *row = plane_in + q * W + x - l;
				for (int k = 0; k < n; ++k) sum += row[k];
			}
			cnt += n;
		}
		return sum / (float)cnt;
	};

	// lane masks "this column's support is not the minimal 3x3": accA / accB accumulate lr(r-1) | all(r) | lr(r+1) as
	// rows arrive, needB holds the finished masks of the row that is output next
	bool accA[4] = {true, true, true, true}, accB[4] = {true, true, true, true};
	bool needB[4] = {true, true, true, true};

	constexpr int PF = 3;
	Stage st[PF];
	C2Row w[3];
#pragma unroll
	for (int u = 0; u < PF; ++u) {
		fetch(st[u], ra + u);
		w[u].A = w[u].B = w[u].C = w[u].D = w[u].E = cb_f2{0.0f, 0.0f};
	}
	const int last = y1 - 1 + L3_LAG;
	for (int g = ra; g <= last; g += L3_RING) {
#pragma unroll
		for (int u = 0; u < L3_RING; ++u) {
			const int r = g + u;
			if (r > last) break;
			Stage &s = st[u % PF];
			// ---- commit row r: values and byte-minimum arm lengths to the ring slot u, flags ----
			const cb_u4 m = bytemin4x4_sdwa(s.a, s.b);
			*(cb_f4 *)&V[u][4 * lane] = cb_f4{__uint_as_float(s.v.x), __uint_as_float(s.v.y), __uint_as_float(s.v.z), __uint_as_float(s.v.w)};
			*(cb_u4 *)&M[u][4 * lane] = m;
			const float nv0 = __uint_as_float(s.v.x), nv1 = __uint_as_float(s.v.y), nv2 = __uint_as_float(s.v.z), nv3 = __uint_as_float(s.v.w);
			bool needA[4];
			{
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
			fetch(s, r + PF);
			// ---- output row yb = r-2: rows r-3, r-2, r-1 are w[u%3], w[(u+1)%3], w[(u+2)%3] ----
			const int yb = r - L3_LAG;
			if (yb >= y0) {
				const C2Row &up = w[u % 3], &own = w[(u + 1) % 3], &dn_ = w[(u + 2) % 3];
				cb_f2 s01 = cb_f2{0.0f, 0.0f}, s23 = cb_f2{0.0f, 0.0f};
				s01 += up.A; s01 += up.B; s01 += up.C;
				s23 += up.C; s23 += up.D; s23 += up.E;
				s01 += own.A; s01 += own.B; s01 += own.C;
				s23 += own.C; s23 += own.D; s23 += own.E;
				s01 += dn_.A; s01 += dn_.B; s01 += dn_.C;
				s23 += dn_.C; s23 += dn_.D; s23 += dn_.E;
				const cb_f2 q01 = div9_pk(s01), q23 = div9_pk(s23);
				float res[4] = {q01.x, q01.y, q23.x, q23.y};
				const float sums[4] = {s01.x, s01.y, s23.x, s23.y};
				bool nj[4], odd = false;
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					nj[j] = needB[j] && inr[j] && valid[j];
					const bool used = !needB[j] && inr[j] && valid[j];
					// outside [2^-95, 2^125): zero, tiny, huge, inf, nan -- the packed form is not proven there
					odd = odd || (used && !(__builtin_fabsf(sums[j]) >= 0x1p-95f && __builtin_fabsf(sums[j]) < 0x1p125f));
				}
				if (__any(odd)) {
#pragma unroll
					for (int j = 0; j < 4; ++j) res[j] = sums[j] / 9.0f;
				}
				if (!wave_all_inr) {
					const float ownv[4] = {own.B.x, own.B.y, own.D.x, own.D.y};
#pragma unroll
					for (int j = 0; j < 4; ++j) res[j] = inr[j] ? res[j] : ownv[j];   // adcensus.cu:353-354: copied through
				}
				// ---- non-minimal supports: each lane re-evaluates its first flagged column from the rings ----
				bool m0 = nj[0], m1 = nj[1], m2 = nj[2], m3 = nj[3];
				while (__any(m0 || m1 || m2 || m3)) {
					const bool act = m0 || m1 || m2 || m3;
					const int jsel = m0 ? 0 : (m1 ? 1 : (m2 ? 2 : 3));
					if (act) {
						const int c = 4 * lane + jsel;
						// rows yb-2 .. yb+2 live in ring slots (u-4 .. u) mod RING: static after unrolling
						cb_u32 mm[5];
						float tv[5][5];
#pragma unroll
						for (int k = 0; k < 5; ++k) {
							constexpr int RR = L3_RING;
							const int slot = (u + RR - 4 + k) % RR;
							mm[k] = M[slot][c];
#pragma unroll
							for (int t = 0; t < 5; ++t) tv[k][t] = V[slot][c + t - 2];
						}
						const cb_u32 ownm = mm[2];
						const int up_n = (int)((ownm >> 16) & 0xff), dn_n = (int)(ownm >> 24);
						bool ok = up_n <= 2 && dn_n <= 2;
						float sum = 0;
						int cnt = 0;
#pragma unroll
						for (int k = 0; k < 5; ++k) {
							const int rel = k - 2;
							const bool ra_ = rel < 0 ? up_n >= -rel : (rel == 0 ? true : dn_n >= rel);
							const int l = (int)(mm[k] & 0xff), rg = (int)((mm[k] >> 8) & 0xff);
							ok = ok && (!ra_ || (l <= 2 && rg <= 2));
#pragma unroll
							for (int t = 0; t < 5; ++t) {
								const int dx = t - 2;
								const bool in = ra_ && (dx < 0 ? l >= -dx : (dx == 0 ? true : rg >= dx));
								sum += in ? tv[k][t] : -0.0f;
							}
							cnt += ra_ ? l + rg + 1 : 0;
						}
						float v = sum / (float)cnt;
						if (!ok) v = general(yb, c, max(max(ra, 0), r - (L3_RING - 1)), min(H - 1, r));
						res[0] = jsel == 0 ? v : res[0];
						res[1] = jsel == 1 ? v : res[1];
						res[2] = jsel == 2 ? v : res[2];
						res[3] = jsel == 3 ? v : res[3];
					}
					// clear the flag that was handled: the first set one of every lane
					m3 = m3 && (m0 || m1 || m2);
					m2 = m2 && (m0 || m1);
					m1 = m1 && m0;
					m0 = false;
				}
				const cb_u32 ob = (cb_u32)((yb * W + xs0) * 4 + lane16);
				if (interior) {
					if (has_out) __builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])}, ro, ob, 0, VOL_AUX);
				} else if (full_out) {
					__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])}, ro, ob, 0, VOL_AUX);
				} else if (any_out) {
#pragma unroll
					for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[j]), ro, xs + j < W ? ob + 4u * j : OOB, 0, 0);
				}
			}
			// ---- row r into the register window (slot of row r-3, which is no longer needed) ----
			{
				const float l3 = lane_from_below(nv3, 0.0f), r0 = lane_from_above(nv0, 0.0f);
				C2Row &nw = w[u % 3];
				nw.A = cb_f2{l3, nv0}; nw.B = cb_f2{nv0, nv1}; nw.C = cb_f2{nv1, nv2}; nw.D = cb_f2{nv2, nv3}; nw.E = cb_f2{nv3, r0};
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) needB[j] = needA[j];
		}
	}
}

int cbca_lean(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int max_arm, hipStream_t st,
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
	A.gx = (int)cdiv(W, L3_STEP);
	// output rows per strip: 4 halo rows per chunk; 40 unless that leaves fewer than ~16 K waves
	const int64_t gy_min = cdiv((int64_t)16384, (int64_t)A.gx * nd);
	const int rb_auto = (int)std::min<int64_t>(40, std::max<int64_t>(16, cdiv((int64_t)H, gy_min)));
	A.rb = cfg.rb > 0 ? cfg.rb : rb_auto;
	A.gy = (int)cdiv(H, A.rb);
	const int64_t waves = (int64_t)cdiv((int64_t)A.gx * A.gy, 8) * 8 * cdiv(nd, 4) * 4;
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)nd * H * W * 4 > ((int64_t)768 << 20);
	if (nt) hipLaunchKernelGGL((cbca_lean_kernel<true>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	else hipLaunchKernelGGL((cbca_lean_kernel<false>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	return check_launch("cbca_lean");
}

}  // namespace mc
