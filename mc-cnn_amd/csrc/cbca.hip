// Cross-based cost aggregation (gfx950): support arms (`cross`) and region mean (`cbca`).
//
// Parity rule: the region sum keeps the reference's order -- rows yy
// ascending, inside a row xx ascending, ONE fp32 accumulator, then an IEEE divide by the
// integer count (adcensus.cu:356-373) -- so that costs, and therefore arg-min disparities,
// are bit-identical.  No separable / prefix-sum shortcut is taken on this path.
#include "cbca_common.h"
#include <algorithm>
#include <type_traits>

namespace mc {

// ---- cross, adcensus.cu:280-322 --------------------------------------------------------------
__global__ void __launch_bounds__(256) cross_kernel(const float *__restrict__ img, float *__restrict__ out, int64_t size, int H, int W,
                                                    int L1, float tau1)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= size) return;
	int64_t t = id;
	const int x = (int)(t % W);
	t /= W;
	const int y = (int)(t % H);
	t /= H;
	const int dir = (int)t;
	const int dx = dir == 0 ? -1 : (dir == 1 ? 1 : 0);
	const int dy = dir == 2 ? -1 : (dir == 3 ? 1 : 0);
	const float c = img[y * W + x];
	int xx, yy;
	for (xx = x + dx, yy = y + dy;; xx += dx, yy += dy) {
		if (xx < 0 || xx >= W || yy < 0 || yy >= H) break;
		const int dist = max(abs(xx - x), abs(yy - y));
		if (dist == 1) continue;
		if (fabsf(c - img[yy * W + xx]) >= tau1) break;  // rule 1
		if (dist >= L1) break;                           // rule 2
	}
	out[id] = (float)(dir <= 1 ? xx : yy);
}

int cross(const float *img, float *arms, int H, int W, int L1, float tau1, hipStream_t st)
{
	const int64_t size = (int64_t)4 * H * W;
	hipLaunchKernelGGL(cross_kernel, dim3(cdiv(size, 256)), dim3(256), 0, st, img, arms, size, H, W, L1, tau1);
	return check_launch("cross");
}

// ---- cbca v1, adcensus.cu:343-377: one thread per voxel, (D,H,W), reads through L1/L2 -------------
__global__ void __launch_bounds__(256) cbca_direct_kernel(const float *__restrict__ x0c, const float *__restrict__ x1c,
                                                          const float *__restrict__ vol, float *__restrict__ out, int D, int H, int W,
                                                          int direction, const uint32_t *__restrict__ only_if)
{
	if (only_if && !*only_if) return;  // standalone fast path handled this call (no arm saturated the packed form)
	const int x = blockIdx.x * 64 + (threadIdx.x & 63);
	const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
	const int d = blockIdx.z;
	if (x >= W || y >= H) return;
	const int64_t HW = (int64_t)H * W;
	const int64_t id = (int64_t)d * HW + (int64_t)y * W + x;
	const int xp = x + d * direction;
	if (xp < 0 || xp >= W) {
		out[id] = vol[id];
		return;
	}
	const float *__restrict__ a0 = x0c, *__restrict__ a1 = x1c;
	const float sh = (float)(d * direction);
	const int yy_s = (int)fmaxf(a0[2 * HW + y * W + x], a1[2 * HW + y * W + xp]);
	const int yy_t = (int)fminf(a0[3 * HW + y * W + x], a1[3 * HW + y * W + xp]);
	float sum = 0;
	int cnt = 0;
	const float *__restrict__ vd = vol + (int64_t)d * HW;
	for (int yy = yy_s + 1; yy < yy_t; yy++) {
		const int xx_s = (int)fmaxf(a0[0 * HW + yy * W + x], a1[0 * HW + yy * W + xp] - sh);
		const int xx_t = (int)fminf(a0[1 * HW + yy * W + x], a1[1 * HW + yy * W + xp] - sh);
		for (int xx = xx_s + 1; xx < xx_t; xx++) {
			sum += vd[yy * W + xx];
			cnt++;
		}
	}
	out[id] = sum / (float)cnt;
}

int cbca(const float *x0c, const float *x1c, const float *vin, float *vout, int D, int H, int W, int direction, hipStream_t st)
{
	hipLaunchKernelGGL(cbca_direct_kernel, dim3(cdiv(W, 64), cdiv(H, 4), D), dim3(256), 0, st, x0c, x1c, vin, vout, D, H, W,
	                   direction, (const uint32_t *)nullptr);
	return check_launch("cbca");
}


// =====================================================================================================
// Packed arm lengths
// =====================================================================================================
// The reference's support of voxel (d,y,x) is, in the frame of the reference pixel, simply the per-arm MINIMUM of the
// two images' arm lengths: with len = |end - coordinate| - 1,
//   yy in [y - min(U0[y,x], U1[y,xp]), y + min(D0[y,x], D1[y,xp])],  xp = x + d*direction,
//   xx in [x - min(L0[yy,x], L1[yy,xp]), x + min(R0[yy,x], R1[yy,xp])]
// (adcensus.cu:359-365: max/min of the exclusive ends, the right image's shifted by -d*direction).  Arm lengths are
// packed once per pair as 4 bytes per pixel (L,R,U,D; saturated at 255) by cbca_pack_kernel.

__global__ void __launch_bounds__(256) cbca_pack_kernel(const float *__restrict__ arms, uint32_t *__restrict__ packed, int H, int W,
                                                        uint32_t *__restrict__ overflow)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t HW = (int64_t)H * W;
	if (id >= HW) return;
	const int x = (int)(id % W), y = (int)(id / W);
	const int l = x - (int)arms[0 * HW + id] - 1;
	const int r = (int)arms[1 * HW + id] - x - 1;
	const int u = y - (int)arms[2 * HW + id] - 1;
	const int d = (int)arms[3 * HW + id] - y - 1;
	auto sat = [](int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
	packed[id] = sat(l) | (sat(r) << 8) | (sat(u) << 16) | (sat(d) << 24);
	if (overflow && (l > 254 || r > 254 || u > 254 || d > 254)) atomicOr(overflow, 1u);
	if (overflow && (l > 4 || r > 4 || u > 4 || d > 4)) atomicOr(overflow + 1, 1u);   // an arm beyond the short-arm kernels (L1 <= 5)
	if (overflow && (l > 13 || r > 13 || u > 13 || d > 13)) atomicOr(overflow + 2, 1u);   // an arm beyond the tile kernel's long-arm instance (L1 <= 14)
}

// =====================================================================================================
// cbca: wave-autonomous strips
// =====================================================================================================
// One wave owns one disparity plane, a strip of 256 staged columns (4 per lane, dwordx4 rows of 1 KB) and RB output
// rows, and walks the strip top to bottom.  Rows are loaded PF steps ahead into registers (volume row, the left image's
// packed lengths, the right image's lengths shifted by d), committed to a wave-private ring of 4 rows in LDS (volume
// values and the byte-wise minimum lengths) and the output row one row behind the newest committed one is produced:
// no block barrier anywhere, LDS traffic is dwordx4, and a lane produces FOUR horizontally adjacent outputs -- columns
// 4*lane+2 .. 4*lane+5 of the frame, so that the 252 outputs of a strip are 63 full dwordx4 stores and the window
// 4*lane .. 4*lane+7 of a lane is two aligned LDS reads per row.  The minimal 3x3 support is computed for all four
// outputs unconditionally (nine additions in the reference's order each).  Outputs with a larger support are few and
// scattered on textured images, so they are COMPACTED: their frame columns go to a small per-wave list, lane i then
// re-runs the reference's loop (rows ascending, x ascending, one accumulator) for list entry i -- out of the ring where
// the support lies inside rows y-2..y+1 / the strip's 256 columns and out of global memory otherwise -- and patches
// the row of results in LDS before it is stored.
// This is the kernel for images on which nearly every support is the minimal 3x3 (Gaussian textures: bandwidth-bound).
// On real scenes most supports are larger and a pass runs at the pace of its largest one: there the window kernel
// (arms <= 4) or, for longer arms, the LISTED instantiation takes over -- the supports that do not fit the window form by
// their SHAPE are then skipped here (the value stored for them is provisional) and come from the pair's list
// (cbca_list_kernel, launched after this kernel on the same stream).
// CS_RING rows per ring, CS_LA rows committed below the current output row (two rows are staged above the first output
// row of a chunk: the window form reaches two rows up), CS_WR column radius of the window form
template <int PF, int CS_RING, int CS_LA, int CS_WR, bool NT, bool LISTED>
__global__ void __launch_bounds__(256) cbca_strip_kernel(const CbcaArgs A)
{
	constexpr int CS_UP = 2;   // rows staged above the first output row: the window form reaches two rows up
	// cache policy of the volume rows: nt (bit 1) for volumes far larger than the 256 MB MALL -- streamed once, only the
	// region's packed lengths should stay cached; smaller volumes (KITTI: 414 MB) are partly served from the MALL on the
	// next iteration and measured faster without the hint
	constexpr int MC_CBCA_VOL_AUX = NT ? 2 : 0;
	auto slot = [](int r) { return (CS_RING & (CS_RING - 1)) == 0 ? (r & (CS_RING - 1)) : (int)((unsigned)(r + 4 * CS_RING) % (unsigned)CS_RING); };
	__shared__ float Vring[4][CS_RING * CS_COLS];
	__shared__ cb_u32 Mring[4][CS_RING * CS_COLS];
	__shared__ float Rrow[4][CS_COLS];          // results of the current row (patched by the compacted pass)
	__shared__ cb_u32 Clist[4][CS_COLS];        // outputs that need the general loop: frame column | up << 16 | down << 24
	if (A.overflow && (A.overflow[0] || (A.by_arm && !A.overflow[1]))) return;   // (by_arm: the pair has short arms only, the window kernel runs)
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: plane descriptors stay in SGPRs
	float *__restrict__ V = Vring[wv];
	cb_u32 *__restrict__ M = Mring[wv];
	float *__restrict__ R = Rrow[wv];
	cb_u32 *__restrict__ CL = Clist[wv];
	const int H = A.H, W = A.W, direction = A.direction;
	const int HWi = H * W;
	// wave -> (region, d).  The four waves of a block take four consecutive disparities of ONE region (strip x row
	// chunk) and the blocks of an XCD (blockIdx % 8) walk all disparity groups of a region before the next region:
	// the packed lengths of a region (left: identical for every d, right: windows shifted by d) are then fetched
	// from HBM once per XCD and served by its L2 for the other D - 1 planes.
	const int dgroups = (A.nd + 3) >> 2;
	const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
	const int region = (kb / dgroups) * 8 + xcd;
	const int d = A.d0 + (kb % dgroups) * 4 + wv;
	if (region >= A.gx * A.gy || d >= A.d0 + A.nd) return;
	const int cx = region % A.gx, cy = region / A.gx;
	const int sh = d * direction;
	const int xs = cx * CS_STEP - 2 + 4 * lane;   // image column of this lane's first staged column
	const int xo = xs + 2;                        // image column of this lane's first output
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const int ra = y0 - CS_UP;                    // first staged row (rows outside the image: loaded as zeros, never used)
	const int plane_bytes = HWi * 4;
	const cb_u32 OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vin + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	const bool full_in = xs >= 0 && xs + 3 < W;                 // all four staged columns exist: one dwordx4 inside the row
	const bool full_out = lane < 63 && xo + 3 < W;
	const bool any_out = lane < 63 && xo < W;
	// per output: bit j = output column exists and its shifted partner is inside the image (adcensus.cu:353-354)
	cb_u32 inr_mask = 0, valid_mask = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xo + j;
		if (lane < 63 && x < W) valid_mask |= 1u << j;
		if (x + sh >= 0 && x + sh < W) inr_mask |= 1u << j;
	}

	struct Stage { cb_u4 v, a, b; };
	auto fetch = [&](Stage &st, int r) {  // row r of the plane -> registers (rows outside the image: zeros)
		const bool rok = r >= 0 && r < H;
		const int base = r * W + xs;
		if (full_in) {
			st.v = __builtin_amdgcn_raw_buffer_load_b128(rv, rok ? (cb_u32)base * 4u : OOB, 0, MC_CBCA_VOL_AUX);
		} else {  // strip edges: per column
			cb_u32 t[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok && xs + k >= 0 && xs + k < W) ? (cb_u32)(base + k) * 4u : OOB, 0, 0);
			st.v = cb_u4{t[0], t[1], t[2], t[3]};
		}
		// lengths: the padded scratch makes any in-row start readable; columns outside the image / the shifted range
		// hold values that are never used
		st.a = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
		st.b = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
	};
	auto commit = [&](const Stage &st, int r) {
		const int o = slot(r) * CS_COLS + 4 * lane;
		*(cb_u4 *)(V + o) = st.v;
		cb_u4 m;
		m.x = bytemin4(st.a.x, st.b.x); m.y = bytemin4(st.a.y, st.b.y); m.z = bytemin4(st.a.z, st.b.z); m.w = bytemin4(st.a.w, st.b.w);
		*(cb_u4 *)(M + o) = m;
	};

	// the reference's loop for the output in frame column c of row yo (any support)
	auto general = [&](int yo, int c, int u, int dn, int lo_row, int hi_row) -> float {
		const int x = cx * CS_STEP - 2 + c;
		float sum = 0;
		int cnt = 0;
		for (int q = yo - u; q <= yo + dn; ++q) {
			const bool row_in = q >= lo_row && q <= hi_row;
			const int rowo = slot(q) * CS_COLS;
			cb_u32 mm;
			if (row_in) mm = M[rowo + c];
			else {
				const int g = q * W + x;
				mm = bytemin4(A.p0[g], A.p1[g + sh]);
			}
			const int l = (int)(mm & 0xff), rg = (int)((mm >> 8) & 0xff);
			const int n = l + rg + 1;
			if (row_in && c - l >= 0 && c + rg < CS_COLS) {
				const float *row = V + rowo + c - l;
				int k = 0;
				for (; k + 4 <= n; k += 4) {
					const float v0 = row[k], v1 = row[k + 1], v2 = row[k + 2], v3 = row[k + 3];
					sum += v0; sum += v1; sum += v2; sum += v3;
				}
				if (k < n) {
					const float v0 = row[k];
					const float v1 = row[min(k + 1, n - 1)], v2 = row[min(k + 2, n - 1)];
					sum += v0;
					if (k + 1 < n) sum += v1;
					if (k + 2 < n) sum += v2;
				}
			} else {  // run leaves the staged frame: from global memory, same order
				const float *row = A.vin + (size_t)d * HWi + q * W + x - l;
				for (int k = 0; k < n; ++k) sum += row[k];
			}
			cnt += n;
		}
		return sum / (float)cnt;
	};

	auto output = [&](int yo) {
		const int s0 = slot(yo) * CS_COLS, sm = slot(yo - 1) * CS_COLS, sp = slot(yo + 1) * CS_COLS;
		const int c0 = 4 * lane;                       // first frame column of this lane's 8-wide window
		const int c1 = lane < 63 ? c0 + 4 : c0;        // (lane 63 has no outputs; keep its reads inside the row)
		cb_u32 mo[4], mu[4], md[4];
		{
			const cb_u2 t0 = *(const cb_u2 *)(M + s0 + c0 + 2), t1 = *(const cb_u2 *)(M + s0 + c1);
			mo[0] = t0.x; mo[1] = t0.y; mo[2] = t1.x; mo[3] = t1.y;
			const cb_u2 u0 = *(const cb_u2 *)(M + sm + c0 + 2), u1 = *(const cb_u2 *)(M + sm + c1);
			mu[0] = u0.x; mu[1] = u0.y; mu[2] = u1.x; mu[3] = u1.y;
			const cb_u2 d0 = *(const cb_u2 *)(M + sp + c0 + 2), d1 = *(const cb_u2 *)(M + sp + c1);
			md[0] = d0.x; md[1] = d0.y; md[2] = d1.x; md[3] = d1.y;
		}
		float ra_[8], rb_[8], rc_[8];
		{
			const cb_f4 a0 = *(const cb_f4 *)(V + sm + c0), a1 = *(const cb_f4 *)(V + sm + c1);
			const cb_f4 b0 = *(const cb_f4 *)(V + s0 + c0), b1 = *(const cb_f4 *)(V + s0 + c1);
			const cb_f4 e0 = *(const cb_f4 *)(V + sp + c0), e1 = *(const cb_f4 *)(V + sp + c1);
#pragma unroll
			for (int k = 0; k < 4; ++k) { ra_[k] = a0[k]; ra_[4 + k] = a1[k]; rb_[k] = b0[k]; rb_[4 + k] = b1[k]; rc_[k] = e0[k]; rc_[4 + k] = e1[k]; }
		}
		float res[4];
		cb_u32 needmask = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			// minimal support <=> own arms all 1 and the rows above / below have left = right = 1 in this column
			const cb_u32 t = (mo[j] ^ 0x01010101u) | ((((mu[j] ^ 0x0101u) | (md[j] ^ 0x0101u)) & 0xffffu) << 16);
			float sum = 0;
			sum += ra_[j + 1]; sum += ra_[j + 2]; sum += ra_[j + 3];
			sum += rb_[j + 1]; sum += rb_[j + 2]; sum += rb_[j + 3];
			sum += rc_[j + 1]; sum += rc_[j + 2]; sum += rc_[j + 3];
			const bool inr = (inr_mask >> j) & 1u;
			res[j] = inr ? sum / 9.0f : rb_[j + 2];   // adcensus.cu:353-354: copied through
			if (t != 0) needmask |= 1u << j;
		}
		needmask &= inr_mask & valid_mask;
		if (__any(needmask != 0)) {
			// compact the (lane, j) pairs that need the general loop into CL[0..n)
			int n = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const bool nj = (needmask >> j) & 1u;
				const uint64_t bal = __ballot(nj);
				const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((cb_u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((cb_u32)bal, 0));
				if (nj) CL[pos] = (cb_u32)(4 * lane + 2 + j) | (mo[j] & 0xffff0000u);
				n += __builtin_popcountll(bal);
			}
			*(cb_f2 *)(R + c0 + 2) = cb_f2{res[0], res[1]};
			if (lane < 63) *(cb_f2 *)(R + c0 + 4) = cb_f2{res[2], res[3]};
			const int lo_row = max(max(ra, 0), yo + CS_LA - (CS_RING - 1)), hi_row = min(H - 1, yo + CS_LA);
			for (int e0 = 0; e0 < n; e0 += 64) {
				const int e = e0 + lane;
				if (e < n) {
					const cb_u32 ent = CL[e];
					const int c = (int)(ent & 0xffffu), u = (int)((ent >> 16) & 0xff), dn = (int)(ent >> 24);
					// Window form: supports inside rows y-2 .. y+LA of the ring and columns x-WR .. x+WR -- nearly all of
					// the scattered ones -- are summed from ONE batch of LDS reads: every tap of the window is read,
					// the taps outside the support add -0.0f (x + -0.0f == x exactly, so the chain of additions is
					// the reference's), rows ascending and x ascending as in the reference.
					constexpr int NWR = 3 + CS_LA;
					cb_u32 mm[NWR];
					float tv[NWR][2 * CS_WR + 1];
#pragma unroll
					for (int k = 0; k < NWR; ++k) {
						const int ro_ = slot(yo + k - 2) * CS_COLS + c;
						mm[k] = M[ro_];
#pragma unroll
						for (int t = 0; t < 2 * CS_WR + 1; ++t) tv[k][t] = V[ro_ + t - CS_WR];
					}
					bool ok = u <= 2 && dn <= CS_LA && yo - u >= lo_row && yo + dn <= hi_row;
					bool fits = u <= 2 && dn <= CS_LA;   // the support's shape alone (cbca_fits_window): inside rows -2 .. +LA, columns +-WR
					float sum = 0;
					int cnt = 0;
#pragma unroll
					for (int k = 0; k < NWR; ++k) {
						const int rel = k - 2;
						const bool act = rel >= -u && rel <= dn;
						const int l = (int)(mm[k] & 0xff), rg = (int)((mm[k] >> 8) & 0xff);
						ok = ok && (!act || (l <= CS_WR && rg <= CS_WR && c - l >= 0 && c + rg < CS_COLS));
						fits = fits && (!act || (l <= CS_WR && rg <= CS_WR));
						const int la = act ? l : -1, rga = act ? rg : -1;   // inactive row: no tap passes
#pragma unroll
						for (int t = 0; t < 2 * CS_WR + 1; ++t) {
							const int dx = t - CS_WR;
							const bool in = dx < 0 ? la >= -dx : (dx == 0 ? act : rga >= dx);
							sum += in ? tv[k][t] : -0.0f;
						}
						cnt += act ? l + rg + 1 : 0;
					}
					// LISTED: a support that does not fit the window by its shape is in the pair's list and is left to
					// cbca_list_kernel, which runs after this launch (the value stored here is provisional)
					if (ok) R[c] = sum / (float)cnt;
					else if (!LISTED || fits) R[c] = general(yo, c, u, dn, lo_row, hi_row);
				}
			}
			const cb_f2 r0 = *(const cb_f2 *)(R + c0 + 2), r1 = *(const cb_f2 *)(R + c1 + (lane < 63 ? 0 : 2));
			res[0] = r0.x; res[1] = r0.y; res[2] = r1.x; res[3] = r1.y;
		}
		const int ob = yo * W + xo;
		if (full_out) {
			__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])},
			                                       ro, (cb_u32)ob * 4u, 0, MC_CBCA_VOL_AUX);
		} else if (any_out) {
#pragma unroll
			for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[j]), ro, xo + j < W ? (cb_u32)(ob + j) * 4u : OOB, 0, 0);
		}
	};

	Stage st[PF];
#pragma unroll
	for (int u = 0; u < PF; ++u) fetch(st[u], ra + u);
	const int last = y1 - 1 + CS_LA;   // newest row that has to be committed for the last output row
	for (int g = ra; g <= last; g += PF) {
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const int r = g + u;
			commit(st[u], r);
			fetch(st[u], r + PF);
			const int yo = r - CS_LA;
			if (yo >= y0 && yo < y1) output(yo);
		}
	}
}


// =====================================================================================================
// cbca: window form for short arms (L1 <= 5) -- every lane walks its own supports
// =====================================================================================================
// Real 8-bit scenes are the opposite regime of the strip kernel's.  Under the KITTI thresholds (L1 = 5, tau1 = 0.13) 90 %
// of the outputs have a support larger than the minimal 3x3 and almost half of the arms sit at the L1-1 = 4 limit
// (measured on the reference's sample pair; tests/util.natural_pair reproduces the statistics): about 40 additions per
// voxel in a fixed order, compute-bound.  The strip kernel's compaction list then holds every output of the row and its
// general loop runs at the pace of the largest support of each pass, from global memory: 6.2 ms per launch at
// 370x1226x228 against 0.8 ms on a Gaussian texture.
//
// Here a wave owns a plane, a strip of 256 staged columns (248 outputs, +-4 halo) and RB rows; rings of 9 rows (values,
// run masks of the minimum arm lengths) cover every support of the output row 4 rows behind the newest one.  A lane owns four
// adjacent outputs.  Per support row it reads its 12-column window once (3 x ds_read_b128) and, per output, turns the
// row's (left, right) into a 9-bit run mask; each of the 9 taps is then  sum += bit ? value : -0.0f  as v_bfe_i32 +
// v_bfi_b32 + v_add_f32.  x + (-0.0f) == x exactly and a value that is not selected is never an operand, so the chain of
// additions is the reference's (rows ascending, x ascending, one accumulator from +0.0) whatever the unselected columns
// hold (NaN triangle included).  The row loop runs over the wave's largest up / down arm and the 5-tap form is used on
// rows where no lane reaches beyond +-2, so textured areas cost a 3 x 5 window.
// (A version of this kernel for arms up to 13 -- 27-row rings, a second walk over +-13 rows -- is in the history, commit
// c898b0c: bit-exact, but slower than one thread per voxel on the realistic pair)
constexpr int CW_ARM = 4;                    // largest arm the window covers
constexpr int CW_STEP = CS_COLS - 2 * CW_ARM;   // 248 output columns per strip (frame columns 4 .. 251)
constexpr int CW_RING = 2 * CW_ARM + 1;
constexpr int CW_VPITCH = CS_COLS + 8;       // 4 columns of padding on either side: the outermost lanes' windows stay inside the row
constexpr int CW_WAVES = 2;                  // waves per block (each has its own rings)

template <bool NT>
__global__ void __launch_bounds__(64 * CW_WAVES) cbca_window_kernel(const CbcaArgs A)
{
	constexpr int VOL_AUX = NT ? 2 : 0;
	__shared__ __attribute__((aligned(16))) float Vring[CW_WAVES][CW_RING][CW_VPITCH];
	__shared__ __attribute__((aligned(16))) unsigned short Mring[CW_WAVES][CW_RING][CS_COLS];
	if (A.overflow && (A.overflow[0] || (A.by_arm && A.overflow[1]))) return;   // (by_arm: the pair has an arm > 4, the strip kernel runs)
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int H = A.H, W = A.W, direction = A.direction;
	const int HWi = H * W;
	// wave -> (region, d) as in the strip kernel: the waves of a block take consecutive disparities of one region, the
	// blocks of an XCD walk all disparity groups of a region before the next one
	const int dgroups = (A.nd + CW_WAVES - 1) / CW_WAVES;
	const int xcd = blockIdx.x & 7, kb = blockIdx.x >> 3;
	const int region = (kb / dgroups) * 8 + xcd;
	const int d = A.d0 + (kb % dgroups) * CW_WAVES + wv;
	if (region >= A.gx * A.gy || d >= A.d0 + A.nd) return;
	const int cx = region % A.gx, cy = region / A.gx;
	const int sh = d * direction;
	const int xs0 = cx * CW_STEP - CW_ARM;        // image column of frame column 0 (wave-uniform)
	const int xs = xs0 + 4 * lane;                // image column of this lane's four columns = its four outputs
	const int y0 = cy * A.rb, y1 = min(H, y0 + A.rb);
	const int ra = y0 - CW_ARM;                   // first staged row
	const int plane_bytes = HWi * 4;
	const cb_u32 OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vin + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(A.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.p1 - CS_PAD), 0, padded_bytes, 0x00020000);
	const bool full_in = xs >= 0 && xs + 3 < W;
	const bool has_out = lane >= 1 && lane <= 62;
	const bool full_out = has_out && xs + 3 < W;
	const bool any_out = has_out && xs < W;
	bool exists[4], inr[4];   // output column exists / its shifted partner is inside the image (adcensus.cu:353-354)
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int x = xs + j;
		exists[j] = has_out && x < W;
		inr[j] = x + sh >= 0 && x + sh < W;
	}
	float *__restrict__ Vw = &Vring[wv][0][0];
	unsigned short *__restrict__ Mw = &Mring[wv][0][0];

	struct Stage { cb_u4 v, a, b; };
	auto fetch = [&](Stage &st, int r) {  // values and lengths of row r -> registers (rows outside the image: zeros)
		const bool rok = r >= 0 && r < H;
		const int base = r * W + xs;
		if (full_in) {
			st.v = __builtin_amdgcn_raw_buffer_load_b128(rv, rok ? (cb_u32)base * 4u : OOB, 0, VOL_AUX);
		} else {
			cb_u32 t[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok && xs + k >= 0 && xs + k < W) ? (cb_u32)(base + k) * 4u : OOB, 0, 0);
			st.v = cb_u4{t[0], t[1], t[2], t[3]};
		}
		st.a = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
		st.b = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
	};
	// per column 16 bits: the row's 9-bit RUN MASK (bit k <-> column offset k - 4 belongs to the run: it depends on the
	// row's left / right lengths only, so it is built once per row here and not once per output row that uses it), up in
	// bits 9..11, down in bits 12..14 (every arm this kernel is launched for is <= CW_ARM = 4)
	auto nib = [](cb_u32 m) -> cb_u32 {
		const cb_u32 l = m & 0xffu, r = (m >> 8) & 0xffu, u = (m >> 16) & 0xffu, d = m >> 24;
		const cb_u32 run = ((1u << (l + r + 1u)) - 1u) << (CW_ARM - l);
		return (run & 0x1ffu) | ((u & 7u) << 9) | ((d & 7u) << 12);
	};
	auto commit = [&](const Stage &st, int slot) {
		*(cb_u4 *)(Vw + slot * CW_VPITCH + 4 + 4 * lane) = st.v;
		const cb_u32 m0 = nib(bytemin4(st.a.x, st.b.x)), m1 = nib(bytemin4(st.a.y, st.b.y));
		const cb_u32 m2 = nib(bytemin4(st.a.z, st.b.z)), m3 = nib(bytemin4(st.a.w, st.b.w));
		*(cb_u2 *)(Mw + slot * CS_COLS + 4 * lane) = cb_u2{m0 | (m1 << 16), m2 | (m3 << 16)};
	};

	auto output = [&](int yo, int rs) {  // rs = ring slot of the newest row yo + CW_ARM
		// own lengths: the up / down arms bound the rows, negative = this output takes no taps at all
		int s0 = rs + (CW_RING - CW_ARM);
		s0 = s0 >= CW_RING ? s0 - CW_RING : s0;
		const cb_u2 mo = *(const cb_u2 *)(Mw + s0 * CS_COLS + 4 * lane);
		const cb_u32 ud[4] = {(mo.x >> 9) & 0x3fu, mo.x >> 25, (mo.y >> 9) & 0x3fu, mo.y >> 25};   // up | down << 3
		int up[4], dn[4], umax = 0, dmax = 0;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const bool take = exists[j] && inr[j];
			up[j] = take ? (int)(ud[j] & 7u) : -1;
			dn[j] = take ? (int)((ud[j] >> 3) & 7u) : -1;
			umax = max(umax, up[j]);
			dmax = max(dmax, dn[j]);
		}
		int Uw = 0, Dw = 0;   // the wave's largest arms (0 .. 4): ballots
#pragma unroll
		for (int t = 1; t <= CW_ARM; ++t) {
			Uw = __any(umax >= t) ? t : Uw;
			Dw = __any(dmax >= t) ? t : Dw;
		}
		// the reference's accumulator starts at +0.0 and the first tap may be -0.0f: the zero is made opaque so that
		// "0 + x" stays an addition
		float sum[4], own[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
		for (int j = 0; j < 4; ++j) asm("v_mov_b32 %0, 0" : "=v"(sum[j]));
		int cnt[4] = {0, 0, 0, 0};
		for (int rel = -Uw; rel <= Dw; ++rel) {
			int sl = rs - CW_ARM + rel;                       // slot of row yo + rel
			sl = sl < 0 ? sl + CW_RING : sl;
			const float *__restrict__ vr = Vw + sl * CW_VPITCH + 4 * lane;   // frame column 4*lane - 4
			const cb_f4 q0 = *(const cb_f4 *)vr, q1 = *(const cb_f4 *)(vr + 4), q2 = *(const cb_f4 *)(vr + 8);
			const float v[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
			const cb_u2 mr = *(const cb_u2 *)(Mw + sl * CS_COLS + 4 * lane);
			const cb_u32 mrow[4] = {mr.x, mr.x >> 16, mr.y, mr.y >> 16};   // low 9 bits: the row's run mask
			if (rel == 0) {
#pragma unroll
				for (int j = 0; j < 4; ++j) own[j] = v[4 + j];
			}
			const int arel = rel < 0 ? -rel : rel;
			cb_u32 mask[4], anywide = 0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int need = rel < 0 ? up[j] : dn[j];
				mask[j] = need >= arel ? (mrow[j] & 0x1ffu) : 0u;
				cnt[j] += __builtin_popcount(mask[j]);
				anywide |= mask[j];
			}
			const bool wide = __any((anywide & 0x183u) != 0);   // a tap at +-3 or +-4 somewhere in the wave
			auto taps = [&](auto first_k, auto last_k) {
#pragma unroll
				for (int j = 0; j < 4; ++j) {
#pragma unroll
					for (int k = first_k; k <= last_k; ++k) {
						const cb_u32 keep = (cb_u32)(((int)(mask[j] << (31 - k))) >> 31);   // all ones if the tap is in the run
						float t;   // keep ? value : -0.0f as ONE bit-field insert (left to itself the compiler shares partial
						           // and/or terms between taps and ends up with four to five operations per tap)
						asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(t) : "v"(keep), "v"(v[j + k]), "s"(0x80000000u));
						sum[j] += t;
					}
				}
			};
			if (wide) taps(std::integral_constant<int, 0>(), std::integral_constant<int, 8>());
			else taps(std::integral_constant<int, 2>(), std::integral_constant<int, 6>());
		}
		float res[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) res[j] = inr[j] ? sum[j] / (float)cnt[j] : own[j];   // adcensus.cu:353-354: copied through
		const int ob = yo * W + xs;
		if (full_out) {
			__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])},
			                                       ro, (cb_u32)ob * 4u, 0, VOL_AUX);
		} else if (any_out) {
#pragma unroll
			for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res[j]), ro, xs + j < W ? (cb_u32)(ob + j) * 4u : OOB, 0, 0);
		}
	};

	constexpr int PF = 2;
	Stage st[PF];
#pragma unroll
	for (int u = 0; u < PF; ++u) fetch(st[u], ra + u);
	const int last = y1 - 1 + CW_ARM;
	int rs = 0;   // ring slot of the row being committed
	for (int g = ra; g <= last; g += PF) {
#pragma unroll
		for (int u = 0; u < PF; ++u) {
			const int r = g + u;
			if (r > last) break;
			commit(st[u], rs);
			fetch(st[u], r + PF);
			const int yo = r - CW_ARM;
			if (yo >= y0 && yo < y1) output(yo, rs);
			rs = rs + 1 == CW_RING ? 0 : rs + 1;
		}
	}
}

// =====================================================================================================
// cbca for long arms (L1 > 5): supports sorted by size, once per pair
// =====================================================================================================
// On real scenes under the Middlebury thresholds (L1 = 14, tau1 = 0.02) half of the supports are the minimal 3x3, 84 %
// hold at most 25 taps and 4-6 % are flat regions of up to 27 x 27 = 729 taps that carry two thirds of all additions
// (tests/util.natural_pair).  Every output is ONE serial chain of additions, so a wave runs at the pace of its largest
// support: the strip kernel's per-row passes (56 ms per launch at 1000x1500x256) and one thread per voxel (21 ms) both
// spend most of their lanes waiting.  The supports do not change between the 2 + 16 iterations of a pair, so they are
// classified ONCE per pair and direction: every output whose support does not fit the strip kernel's window form by its
// shape goes, by size class, into a list of voxel indices (count pass, prefix, fill pass).  Per iteration the strip kernel
// (LISTED) then skips those outputs and cbca_list_kernel walks the list, a lane per entry: the lanes of a wave hold
// supports of one size class, and a persistent grid strides over the list so that all CUs work on the same class at a time.
constexpr int CL_BUCKETS = 8;
constexpr int CL_BLOCKS = 2048;   // persistent grid of the classification passes (8 blocks per CU)
struct CbcaListHdr {
	uint32_t total;
	uint32_t pad[31];
	uint32_t block[CL_BLOCKS][CL_BUCKETS];   // count pass: entries of block k in class b; after the scan: its first entry
};

// A list entry: the voxel index and, where the support has at most 9 rows and arms <= 15, its shape -- w[0] bits 0..3 up,
// 4..7 down, then one byte (left | right << 4) per support row from the top one -- so that the list kernel fetches an
// entry with ONE coalesced 16-byte load instead of two scattered length loads per support row (it was bound by the
// L1's line rate: 4.4 G line accesses per launch, 21 per load instruction).  w[2] bit 31: no shape, walk the packed maps.
struct __attribute__((aligned(16))) CbcaListEntry { uint32_t id, w[3]; };
constexpr uint32_t CL_NOSHAPE = 0x80000000u;

__device__ __forceinline__ int cl_bucket(int size)
{
	return size <= 12 ? 0 : size <= 20 ? 1 : size <= 32 ? 2 : size <= 56 ? 3 : size <= 100 ? 4 : size <= 200 ? 5 : size <= 400 ? 6 : 7;
}

// shape test shared with the strip kernel (its `fits`): rows -2 .. +1, columns +-2 -- and the support's tap count
__device__ __forceinline__ bool cbca_listed(const uint32_t *__restrict__ p0, const uint32_t *__restrict__ p1, int y, int x, int sh, int W,
                                            int &size)
{
	const int g0 = y * W + x;
	const uint32_t own = bytemin4(p0[g0], p1[g0 + sh]);
	const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
	bool fits = u <= 2 && dn <= 1;
	int n = 0;
	for (int q = y - u; q <= y + dn; ++q) {
		const int g = q * W + x;
		const uint32_t mm = q == y ? own : bytemin4(p0[g], p1[g + sh]);
		const int l = (int)(mm & 0xff), r = (int)((mm >> 8) & 0xff);
		fits = fits && l <= 2 && r <= 2;
		n += l + r + 1;
	}
	size = n;
	return !fits;
}

// A block owns a contiguous range of (plane, 4 rows, 64 columns) tiles and keeps its eight class counters in LDS:
// FILL = false classifies its outputs (class byte per voxel, 0xff = not listed) and counts them per class, FILL = true
// (after the scan has turned the counts into first positions) walks the same tiles again, reads the class bytes and
// writes the voxel indices.  No global atomics.
template <bool FILL>
__global__ void __launch_bounds__(256) cbca_list_build_kernel(const uint32_t *__restrict__ p0, const uint32_t *__restrict__ p1,
                                                              CbcaListHdr *__restrict__ hdr, CbcaListEntry *__restrict__ list,
                                                              uint8_t *__restrict__ cls, int D, int H, int W, int direction)
{
	__shared__ uint32_t cur[CL_BUCKETS];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x < CL_BUCKETS) cur[threadIdx.x] = FILL ? hdr->block[blockIdx.x][threadIdx.x] : 0u;
	__syncthreads();
	const int tx = (W + 63) >> 6, ty = (H + 3) >> 2;
	const int64_t tiles = (int64_t)tx * ty * D;
	const int64_t per = (tiles + gridDim.x - 1) / gridDim.x;
	const int64_t t0 = per * blockIdx.x, t1 = min(tiles, t0 + per);
	// what a tile row needs from memory is fetched one tile ahead: the loop is a chain of round trips otherwise (a block
	// walks ~750 tiles one after the other)
	struct Pre {
		int x, y, sh;
		bool inside, valid;
		uint32_t id;
		uint32_t a0, a1, o0, o1, b0, b1;   // count pass: lengths of rows y-1, y, y+1 in both images
		int cls;                           // fill pass: the class byte
	};
	// tile coordinates advance by counters (a 64-bit division per tile was most of the kernel's instructions)
	int fbx = (int)(t0 % tx), fby = (int)((t0 / tx) % ty), fd = (int)(t0 / ((int64_t)tx * ty));
	auto fetch = [&](int64_t t) -> Pre {
		Pre q;
		const int bx = fbx, by = fby, d = fd;
		if (++fbx == tx) {
			fbx = 0;
			if (++fby == ty) { fby = 0; ++fd; }
		}
		q.x = bx * 64 + lane; q.y = by * 4 + wv;
		q.sh = d * direction;
		q.inside = t < t1 && q.x < W && q.y < H;
		q.valid = q.inside && q.x + q.sh >= 0 && q.x + q.sh < W;
		q.id = (uint32_t)((d * H + q.y) * W + q.x);
		q.a0 = q.a1 = q.o0 = q.o1 = q.b0 = q.b1 = 0;
		q.cls = -1;
		if (FILL) {
			if (q.inside) q.cls = (int)(int8_t)cls[q.id];   // 0xff -> -1
		} else if (q.valid) {
			const int g = q.y * W + q.x, ga = max(q.y - 1, 0) * W + q.x, gb = min(q.y + 1, H - 1) * W + q.x;
			q.a0 = p0[ga]; q.a1 = p1[ga + q.sh]; q.o0 = p0[g]; q.o1 = p1[g + q.sh]; q.b0 = p0[gb]; q.b1 = p1[gb + q.sh];
		}
		return q;
	};
	Pre nxt = fetch(t0);
	for (int64_t t = t0; t < t1; ++t) {
		const Pre c = nxt;
		nxt = fetch(t + 1);
		const int x = c.x, y = c.y, sh = c.sh;
		const bool inside = c.inside;
		const uint32_t id = c.id;
		int bucket = -1;
		if (FILL) {
			bucket = c.cls;
		} else {
			if (c.valid) {
				// most supports reach one row up and down: the three rows' lengths were fetched together; only taller
				// supports walk their rows one after the other
				const uint32_t own = bytemin4(c.o0, c.o1);
				const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
				if (u <= 1 && dn <= 1) {
					const uint32_t ma = bytemin4(c.a0, c.a1), mb = bytemin4(c.b0, c.b1);
					auto lr = [](uint32_t m, int &l, int &r) { l = (int)(m & 0xff); r = (int)((m >> 8) & 0xff); };
					int l, r, n;
					lr(own, l, r);
					bool fits = l <= 2 && r <= 2;
					n = l + r + 1;
					if (u == 1) { lr(ma, l, r); fits = fits && l <= 2 && r <= 2; n += l + r + 1; }
					if (dn == 1) { lr(mb, l, r); fits = fits && l <= 2 && r <= 2; n += l + r + 1; }
					if (!fits) bucket = cl_bucket(n);
				} else {
					int size;
					if (cbca_listed(p0, p1, y, x, sh, W, size)) bucket = cl_bucket(size);
				}
			}
			if (inside) cls[id] = (uint8_t)bucket;
		}
		if (!__any(bucket >= 0)) continue;   // wave-uniform: nothing listed in this row of the tile
		CbcaListEntry e;
		if (FILL && bucket >= 0) {
			// the entry's shape: lengths of its rows, top to bottom (all listed lanes of the row together)
			e.id = id;
			const uint32_t own = bytemin4(p0[y * W + x], p1[y * W + x + sh]);
			const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
			unsigned long long lo = (unsigned long long)((u & 15) | ((dn & 15) << 4));   // bytes 0..7 of the 12
			uint32_t hi = 0;                                                               // bytes 8..11
			bool shape = u + dn + 1 <= 9;
			if (shape) {
				for (int k = 0; k <= u + dn; ++k) {
					const int g = (y - u + k) * W + x;
					const uint32_t mm = k == u ? own : bytemin4(p0[g], p1[g + sh]);
					const uint32_t l = mm & 0xff, r = (mm >> 8) & 0xff;
					shape = shape && l <= 15 && r <= 15;
					const uint32_t byte = (l & 15u) | ((r & 15u) << 4);
					if (k < 7) lo |= (unsigned long long)byte << (8 * (k + 1));
					else hi |= byte << (8 * (k - 7));
				}
			}
			e.w[0] = (uint32_t)lo; e.w[1] = (uint32_t)(lo >> 32); e.w[2] = shape ? hi : CL_NOSHAPE;
		}
#pragma unroll
		for (int b = 0; b < CL_BUCKETS; ++b) {
			const unsigned long long m = __ballot(bucket == b);
			if (m == 0) continue;   // wave-uniform
			const int leader = __builtin_ctzll(m);
			uint32_t start = 0;
			if (lane == leader) start = atomicAdd(&cur[b], (uint32_t)__builtin_popcountll(m));
			if (FILL) {
				start = (uint32_t)__builtin_amdgcn_readlane((int)start, leader);
				const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
				if (bucket == b) list[start + rank] = e;
			}
		}
	}
	if (!FILL) {
		__syncthreads();
		if (threadIdx.x < CL_BUCKETS) hdr->block[blockIdx.x][threadIdx.x] = cur[threadIdx.x];
	}
}

// counts -> first positions: block after block (a block's tiles are one compact region of the volume, so that region's
// entries are consecutive in the list and meet in one L2), inside a block class after class (the 64 consecutive entries
// of a wave are of one size class except at the few class boundaries)
__global__ void __launch_bounds__(256) cbca_list_scan_kernel(CbcaListHdr *__restrict__ hdr)
{
	constexpr int PER = CL_BLOCKS / 256;
	__shared__ uint32_t part[256];
	uint32_t mine = 0;
	for (int k = threadIdx.x * PER; k < (threadIdx.x + 1) * PER; ++k)
		for (int b = 0; b < CL_BUCKETS; ++b) mine += hdr->block[k][b];
	part[threadIdx.x] = mine;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (int t = 0; t < 256; ++t) {
			const uint32_t n = part[t];
			part[t] = run;
			run += n;
		}
		hdr->total = run;
	}
	__syncthreads();
	uint32_t run = part[threadIdx.x];
	for (int k = threadIdx.x * PER; k < (threadIdx.x + 1) * PER; ++k) {
		for (int b = 0; b < CL_BUCKETS; ++b) {
			const uint32_t n = hdr->block[k][b];
			hdr->block[k][b] = run;
			run += n;
		}
	}
}

// unsigned 32-bit division by a launch-time constant: q = (t + ((n - t) >> 1)) >> (s - 1), t = umulhi(n, M)
struct ClDiv { uint32_t M, s1; };
static ClDiv cl_div_make(uint32_t dv)   // dv >= 2
{
	uint32_t s = 0;
	while (((uint64_t)1 << s) < dv) ++s;
	ClDiv r;
	r.M = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << s) - dv)) / dv + 1);
	r.s1 = s - 1;
	return r;
}
__device__ __forceinline__ uint32_t cl_div(uint32_t n, ClDiv dv)
{
	const uint32_t t = __umulhi(n, dv.M);
	return (t + ((n - t) >> 1)) >> dv.s1;
}

// one lane per list entry; the reference's loop (adcensus.cu:356-373) with the lengths from the packed maps.  The lanes
// of a wave hold supports of one size class, so they leave the loops together.  Most entries are small supports whose
// cost is per-entry and per-row overhead, not additions: the voxel index is split with multiplications, the next row's
// lengths are fetched before the current row is summed, a run of up to 8 / 16 values is fetched whole with two / four
// 16-byte loads (dword-aligned: the run starts anywhere; values behind the run's end are fetched but never added) and
// longer runs 16 values at a time.
typedef float cl_f4u __attribute__((ext_vector_type(4), aligned(4)));   // four floats at any dword address

template <bool NT>
__global__ void __launch_bounds__(256) cbca_list_kernel(const CbcaListEntry *__restrict__ list, const CbcaListHdr *__restrict__ hdr, const CbcaArgs A,
                                                        const ClDiv divHW, const ClDiv divW)
{
	const uint32_t total = hdr->total;
	const int H = A.H, W = A.W, HWi = H * W;
	const uint32_t nvox = (uint32_t)A.D * (uint32_t)HWi;
	// the grid takes the list in windows of gridDim.x chunks of 256 entries; inside a window the blocks of one XCD
	// (blockIdx % 8) take CONSECUTIVE chunks, so that neighbouring supports meet in one L2
	const uint32_t stride = gridDim.x * 256u;
	const uint32_t slot = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
	for (uint32_t i = slot * 256u + threadIdx.x; i < total; i += stride) {
		const cb_u4 ent = *(const cb_u4 *)&list[i];
		const uint32_t id = ent.x;
		const uint32_t d = cl_div(id, divHW);
		const uint32_t rem = id - d * (uint32_t)HWi;
		const int y = (int)cl_div(rem, divW), x = (int)rem - y * W;
		const float *__restrict__ plane = A.vin + (size_t)d * HWi;
		const uint32_t vbase = d * (uint32_t)HWi;
		float sum = 0;
		int cnt = 0;
		// n values of the run that starts at element ro of the plane
		auto add_run = [&](int ro, int n) {
			const float *__restrict__ row = plane + ro;
			// 16 floats from the run's start stay inside the volume (the last rows of the last plane take the exact path)
			const bool whole = n <= 16 && vbase + (uint32_t)ro + 16u <= nvox;
			if (whole) {
				const cl_f4u v0 = *(const cl_f4u *)row;
				cl_f4u v1 = cl_f4u{0.0f, 0.0f, 0.0f, 0.0f};
				if (n > 4) v1 = *(const cl_f4u *)(row + 4);   // (a lane whose run ends inside the first 16 bytes issues no second access)
				const float a[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
				for (int t = 0; t < 8; ++t) sum += t < n ? a[t] : -0.0f;   // x + (-0.0f) == x: the values behind the run are no operands
				if (n > 8) {
					const cl_f4u v2 = *(const cl_f4u *)(row + 8), v3 = *(const cl_f4u *)(row + 12);
					const float b[8] = {v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
					for (int t = 0; t < 8; ++t) sum += 8 + t < n ? b[t] : -0.0f;
				}
			} else {
				int k = 0;
				for (; k + 16 <= n; k += 16) {
					cl_f4u v[4];
#pragma unroll
					for (int t = 0; t < 4; ++t) v[t] = *(const cl_f4u *)(row + k + 4 * t);
#pragma unroll
					for (int t = 0; t < 4; ++t) { sum += v[t].x; sum += v[t].y; sum += v[t].z; sum += v[t].w; }
				}
				if (k + 8 <= n) {
					const cl_f4u v0 = *(const cl_f4u *)(row + k), v1 = *(const cl_f4u *)(row + k + 4);
					sum += v0.x; sum += v0.y; sum += v0.z; sum += v0.w;
					sum += v1.x; sum += v1.y; sum += v1.z; sum += v1.w;
					k += 8;
				}
				if (k + 4 <= n) {
					const cl_f4u v0 = *(const cl_f4u *)(row + k);
					sum += v0.x; sum += v0.y; sum += v0.z; sum += v0.w;
					k += 4;
				}
				if (k < n) {   // 1 .. 3 left: single loads (a 16-byte load could reach past the end of the volume)
					const float a0 = row[k], a1 = row[min(k + 1, n - 1)], a2 = row[min(k + 2, n - 1)];
					sum += a0;
					if (k + 1 < n) sum += a1;
					if (k + 2 < n) sum += a2;
				}
			}
			cnt += n;
		};
		if (!(ent.w & CL_NOSHAPE)) {
			// the shape travels with the entry: no length loads at all
			const int u = (int)(ent.y & 15u), dn = (int)((ent.y >> 4) & 15u);
			uint32_t w0 = ent.y >> 8, w1 = ent.z, w2 = ent.w;   // a byte per row, next row = low byte
			for (int q = y - u; q <= y + dn; ++q) {
				const int l = (int)(w0 & 15u), r = (int)((w0 >> 4) & 15u);
				w0 = (w0 >> 8) | (w1 << 16);   // 24 + 32 + 32 bits shifted right by a byte
				w1 = (w1 >> 8) | (w2 << 24);
				w2 >>= 8;
				add_run(q * W + x - l, l + r + 1);
			}
		} else {
			const int sh = (int)d * A.direction;
			const uint32_t *__restrict__ q0 = A.p0 + x, *__restrict__ q1 = A.p1 + x + sh;
			const uint32_t own = bytemin4(q0[y * W], q1[y * W]);
			const int u = (int)((own >> 16) & 0xff), dn = (int)(own >> 24);
			uint32_t nxt = bytemin4(q0[(y - u) * W], q1[(y - u) * W]);
			for (int q = y - u; q <= y + dn; ++q) {
				const uint32_t mm = nxt;
				const int qn = min(q + 1, y + dn);
				nxt = bytemin4(q0[qn * W], q1[qn * W]);   // the next row's lengths travel while this row is summed
				const int l = (int)(mm & 0xff), r = (int)((mm >> 8) & 0xff);
				add_run(q * W + x - l, l + r + 1);
			}
		}
		const float res = sum / (float)cnt;
		if (NT) __builtin_nontemporal_store(res, A.vout + id);
		else A.vout[id] = res;
	}
}

// header + one 16-byte entry per output (worst case: every output is listed) + one class byte per voxel
static size_t cl_list_entries(int D, int H, int W) { return ((size_t)D * H * W + 63) & ~(size_t)63; }
size_t cbca_list_bytes(int D, int H, int W)
{
	return (sizeof(CbcaListHdr) + cl_list_entries(D, H, W) * sizeof(CbcaListEntry) + (size_t)D * H * W + 255) & ~(size_t)255;
}

// classification of a (pair, direction): which outputs the list kernel owns, sorted by support size.  Arms <= 254 and
// D*H*W < 2^31 required (callers check).
int cbca_list_build(const void *packed, void *listmem, int D, int H, int W, int direction, hipStream_t st)
{
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	CbcaListHdr *hdr = (CbcaListHdr *)listmem;
	CbcaListEntry *list = (CbcaListEntry *)((char *)listmem + sizeof(CbcaListHdr));
	uint8_t *cls = (uint8_t *)(list + cl_list_entries(D, H, W));
	const dim3 grid(CL_BLOCKS), block(256);
	hipLaunchKernelGGL((cbca_list_build_kernel<false>), grid, block, 0, st, cs.p0, cs.p1, hdr, list, cls, D, H, W, direction);
	hipLaunchKernelGGL(cbca_list_scan_kernel, dim3(1), dim3(256), 0, st, hdr);
	hipLaunchKernelGGL((cbca_list_build_kernel<true>), grid, block, 0, st, cs.p0, cs.p1, hdr, list, cls, D, H, W, direction);
	return check_launch("cbca_list_build");
}

size_t cbca_scratch_bytes(int H, int W) { return (((size_t)2 * H * W + 3 * CS_PAD + CS_FLAGS) * sizeof(uint32_t) + 255) & ~(size_t)255; }

int cbca_pack(const float *x0c, const float *x1c, void *scratch, int H, int W, hipStream_t st)
{
	const CbcaScratch cs = cbca_scratch(scratch, H, W);
	uint32_t *p0 = cs.p0, *p1 = cs.p1, *flag = cs.flag;
	const int64_t HW = (int64_t)H * W;
	const hipError_t e = hipMemsetAsync(flag, 0, CS_FLAGS * sizeof(uint32_t), st);
	if (e != hipSuccess) {
		set_error("cbca_pack: %s", hipGetErrorString(e));
		return (int)e;
	}
	hipLaunchKernelGGL(cbca_pack_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, x0c, p0, H, W, flag);
	hipLaunchKernelGGL(cbca_pack_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, x1c, p1, H, W, flag);
	return check_launch("cbca_pack");
}

// v1 kernel that runs only when cbca_pack flagged a saturated arm (and the tiled kernel therefore stood down)
int cbca_if_overflow(const float *x0c, const float *x1c, const void *packed, const float *vin, float *vout, int D, int H, int W,
                     int direction, hipStream_t st)
{
	const uint32_t *flag = cbca_scratch(packed, H, W).flag;
	hipLaunchKernelGGL(cbca_direct_kernel, dim3(cdiv(W, 64), cdiv(H, 4), D), dim3(256), 0, st, x0c, x1c, vin, vout, D, H, W,
	                   direction, flag);
	return check_launch("cbca (overflow path)");
}

// max_arm: largest arm length that can occur (L1-1 when L1 is known, else < 0: the pack kernel's overflow flag decides).
// Arm lengths saturate at 255 in the packed form: callers route max_arm > 254 to the direct kernel.
// cfg (mc_common.h): rows per strip, cache policy and plane range; zero / negative fields = derived from the size.
int cbca_strips(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int max_arm, hipStream_t st,
                const CbcaCfg &cfg, const void *listmem)
{
	const int d0 = cfg.nd > 0 ? cfg.d0 : 0, nd = cfg.nd > 0 ? cfg.nd : D;
	CbcaArgs A;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = vin; A.vout = vout;
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.d0 = d0; A.nd = nd;
	A.overflow = max_arm < 0 ? cs.flag : nullptr;  // unknown arm bound: honour cbca_pack's flag
	A.gx = (int)cdiv(W, CS_STEP);
	// output rows per strip: 40 (5 % of halo rows) unless that leaves fewer than ~16 K waves -- at KITTI size (5 strips x
	// 228 planes) 27 and 20 rows measured 5 % faster than 40, 53 rows 18 % slower
	const int64_t gy_min = cdiv((int64_t)16384, (int64_t)A.gx * nd);
	const int rb_auto = (int)std::min<int64_t>(40, std::max<int64_t>(16, cdiv((int64_t)H, gy_min)));
	A.rb = cfg.rb > 0 ? cfg.rb : rb_auto;
	A.gy = (int)cdiv(H, A.rb);
	const int64_t waves = (int64_t)cdiv((int64_t)A.gx * A.gy, 8) * 8 * cdiv(nd, 4) * 4;
	// non-temporal volume accesses for volumes far larger than the 256 MB Infinity Cache (see cbca_strip_kernel)
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)nd * H * W * 4 > ((int64_t)768 << 20);
	// short arms (L1 <= 5, the KITTI parameter sets): every lane walks its own supports out of 9-row rings; on a Gaussian
	// texture it is within 8 % of the strip kernel, on real-scene arm statistics 4.6 x faster.  With the arm bound unknown
	// (the standalone operator) both kernels are launched and cbca_pack's second flag word lets exactly one of them run.
	const bool by_arm = cfg.form == 0 && max_arm < 0 && !listmem;
	const bool window = cfg.form == 2 || (cfg.form == 0 && max_arm >= 0 && max_arm <= CW_ARM) || by_arm;
	A.by_arm = by_arm ? 1 : 0;
	if (window) {
		CbcaArgs B = A;
		B.gx = (int)cdiv(W, CW_STEP);
		const int64_t gy_w = cdiv((int64_t)16384, (int64_t)B.gx * nd);
		B.rb = cfg.rb > 0 ? cfg.rb : (int)std::min<int64_t>(40, std::max<int64_t>(16, cdiv((int64_t)H, gy_w)));
		B.gy = (int)cdiv(H, B.rb);
		const int64_t wv = (int64_t)cdiv((int64_t)B.gx * B.gy, 8) * 8 * cdiv(nd, CW_WAVES) * CW_WAVES;
		if (nt) hipLaunchKernelGGL((cbca_window_kernel<true>), dim3((unsigned)cdiv(wv, CW_WAVES)), dim3(64 * CW_WAVES), 0, st, B);
		else hipLaunchKernelGGL((cbca_window_kernel<false>), dim3((unsigned)cdiv(wv, CW_WAVES)), dim3(64 * CW_WAVES), 0, st, B);
		if (!by_arm) return check_launch("cbca_window");
	}
	// prefetch 2 rows, ring of 4 rows, 1 row of look-ahead, window form +-2 columns (+-4 measured slower at KITTI and 1000x1500)
	if (listmem) {
		// the pair's list (cbca_list_build) owns the supports that do not fit the window form; the list kernel follows on
		// the same stream and overwrites the provisional values the strip kernel stored for them
		if (nt) hipLaunchKernelGGL((cbca_strip_kernel<2, 4, 1, 2, true, true>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
		else hipLaunchKernelGGL((cbca_strip_kernel<2, 4, 1, 2, false, true>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
		const CbcaListHdr *hdr = (const CbcaListHdr *)listmem;
		const CbcaListEntry *list = (const CbcaListEntry *)((const char *)listmem + sizeof(CbcaListHdr));
		const dim3 lgrid(256 * 8);   // persistent: 8 blocks per CU stride over the list
		const ClDiv dHW = cl_div_make((uint32_t)H * (uint32_t)W), dW = cl_div_make((uint32_t)W);
		if (nt) hipLaunchKernelGGL((cbca_list_kernel<true>), lgrid, dim3(256), 0, st, list, hdr, A, dHW, dW);
		else hipLaunchKernelGGL((cbca_list_kernel<false>), lgrid, dim3(256), 0, st, list, hdr, A, dHW, dW);
		return check_launch("cbca_strip + cbca_list");
	}
	if (nt) hipLaunchKernelGGL((cbca_strip_kernel<2, 4, 1, 2, true, false>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	else hipLaunchKernelGGL((cbca_strip_kernel<2, 4, 1, 2, false, false>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	return check_launch("cbca_strip");
}

}  // namespace mc
