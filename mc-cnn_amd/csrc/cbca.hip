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
	if (only_if && only_if[CF_ROUTE] != CR_DIRECT) return;  // (the packed forms handle this pair)
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

int cbca_lean_rows(int D, int H, int W, int rb, bool two_pass);   // cbca_lean.hip

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
                                                        uint32_t *__restrict__ flags, int image)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t HW = (int64_t)H * W;
	bool unit = false, gt254 = false, gt4 = false, gt13 = false;
	if (id < HW) {
		const int x = (int)(id % W), y = (int)(id / W);
		const int l = x - (int)arms[0 * HW + id] - 1;
		const int r = (int)arms[1 * HW + id] - x - 1;
		const int u = y - (int)arms[2 * HW + id] - 1;
		const int d = (int)arms[3 * HW + id] - y - 1;
		auto sat = [](int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
		packed[id] = sat(l) | (sat(r) << 8) | (sat(u) << 16) | (sat(d) << 24);
		const int longest = max(max(l, r), max(u, d));
		gt254 = longest > 254;
		gt4 = longest > 4;      // an arm beyond the tile kernel's short-arm instance (L1 <= 5)
		gt13 = longest > 13;    // ... beyond its long-arm instance (L1 <= 14)
		unit = longest <= 1;
	}
	// one atomic per BLOCK and flag, and none once a flag is up (on real scenes nearly every wave has an arm > 4: a million
	// atomics on one word took 0.45 ms per image at 1000 x 1500)
	__shared__ uint32_t blk[4];
	if (threadIdx.x < 4) blk[threadIdx.x] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	if (__any(gt254) && lane == 0) blk[0] = 1;
	if (__any(gt4) && lane == 0) blk[1] = 1;
	if (__any(gt13) && lane == 0) blk[2] = 1;
	// pixels whose four arms are all at the minimum: where nearly every pixel of both images is one, nearly every support
	// is the minimal 3 x 3 and aggregation is a bandwidth problem (the strip kernel's regime)
	const unsigned long long m = __ballot(unit);
	if (lane == 0 && m) atomicAdd(&blk[3], (uint32_t)__builtin_popcountll(m));
	__syncthreads();
	if (threadIdx.x == 0) {
		if (blk[0] && !flags[CF_SATURATED]) atomicOr(flags + CF_SATURATED, 1u);
		if (blk[1] && !flags[CF_ARM_GT4]) atomicOr(flags + CF_ARM_GT4, 1u);
		if (blk[2] && !flags[CF_ARM_GT13]) atomicOr(flags + CF_ARM_GT13, 1u);
		if (blk[3]) atomicAdd(flags + CF_UNIT_PIXELS + image, blk[3]);
	}
}

// the kernel the pair's arms call for (one thread, after both cbca_pack_kernel launches)
__global__ void cbca_route_kernel(uint32_t *__restrict__ flags, int64_t HW)
{
	// measured (1000x1500x256, L1 = 14): Gaussian texture -- 87 / 93 % of the two images' pixels have unit arms -- strip kernel 0.97 ms
	// per launch against the tile kernel's 1.75; real-scene statistics -- 28 / 30 % -- 56 ms against 3.8.  Threshold: 75 % on average.
	const bool textured = ((int64_t)flags[CF_UNIT_PIXELS] + (int64_t)flags[CF_UNIT_PIXELS + 1]) * 2 >= HW * 3;
	flags[CF_ROUTE] = flags[CF_SATURATED] ? CR_DIRECT : (!flags[CF_ARM_GT4] ? CR_TILE4 : ((flags[CF_ARM_GT13] || textured) ? CR_STRIP : CR_TILE13));
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
// This is the kernel for images on which nearly every support is the minimal 3x3 (Gaussian textures: bandwidth-bound),
// and for arms beyond the tile kernel's (L1 > 14).  On real scenes most supports are larger and a pass runs at the pace of
// its largest one: there the tile kernel (cbca_tile.hip) takes over.
// CS_RING rows per ring, CS_LA rows committed below the current output row (two rows are staged above the first output
// row of a chunk: the window form reaches two rows up), CS_WR column radius of the window form
template <int PF, int CS_RING, int CS_LA, int CS_WR, bool NT>
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
	if (A.route == CR_STRIP_IF_NO_LIST || A.route == CR_NOT_DIRECT_IF_NO_LIST) {   // fallback of the lean + list kernels: the pair's list did not fit (or was never written)
		if (!cbca_gate(A.flags, A.route == CR_STRIP_IF_NO_LIST ? (int)CR_STRIP : (int)CR_NOT_DIRECT) ||
		    (A.plan && list_valid((const uint32_t *)A.plan, A.D, A.H, A.W, A.direction, A.lean_rb))) return;   // (no plan area: no list)
	} else if (A.route == CR_STRIP_IF_LIST) {   // a single pass of mc_predict's planned passes on a texture whose list is usable
		if (!A.plan || !cbca_gate_planned(A.flags, A.route, (const uint32_t *)A.plan, A.D, A.H, A.W, A.direction, A.lean_rb)) return;
	} else if (!cbca_gate(A.flags, A.route)) return;   // (the pair's arms call for another kernel)
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
					float sum = 0;
					int cnt = 0;
#pragma unroll
					for (int k = 0; k < NWR; ++k) {
						const int rel = k - 2;
						const bool act = rel >= -u && rel <= dn;
						const int l = (int)(mm[k] & 0xff), rg = (int)((mm[k] >> 8) & 0xff);
						ok = ok && (!act || (l <= CS_WR && rg <= CS_WR && c - l >= 0 && c + rg < CS_COLS));
						const int la = act ? l : -1, rga = act ? rg : -1;   // inactive row: no tap passes
#pragma unroll
						for (int t = 0; t < 2 * CS_WR + 1; ++t) {
							const int dx = t - CS_WR;
							const bool in = dx < 0 ? la >= -dx : (dx == 0 ? act : rga >= dx);
							sum += in ? tv[k][t] : -0.0f;
						}
						cnt += act ? l + rg + 1 : 0;
					}
					if (ok) R[c] = sum / (float)cnt;
					else R[c] = general(yo, c, u, dn, lo_row, hi_row);
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
	hipLaunchKernelGGL(cbca_pack_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, x0c, p0, H, W, flag, 0);
	hipLaunchKernelGGL(cbca_pack_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, st, x1c, p1, H, W, flag, 1);
	hipLaunchKernelGGL(cbca_route_kernel, dim3(1), dim3(1), 0, st, flag, HW);
	return check_launch("cbca_pack");
}

// v1 kernel that runs only when cbca_pack saw an arm the packed form cannot hold (and the other kernels therefore stood down)
int cbca_if_overflow(const float *x0c, const float *x1c, const void *packed, const float *vin, float *vout, int D, int H, int W,
                     int direction, hipStream_t st)
{
	const uint32_t *flag = cbca_scratch(packed, H, W).flag;
	hipLaunchKernelGGL(cbca_direct_kernel, dim3(cdiv(W, 64), cdiv(H, 4), D), dim3(256), 0, st, x0c, x1c, vin, vout, D, H, W,
	                   direction, flag);
	return check_launch("cbca (overflow path)");
}

// route >= 0 (the caller does not know the arms): the launch stands down unless cbca_pack's route word equals it.
// cfg (mc_common.h): rows per strip, cache policy and plane range; zero / negative fields = derived from the size.
int cbca_strips(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int route, hipStream_t st,
                const CbcaCfg &cfg)
{
	const int d0 = cfg.nd > 0 ? cfg.d0 : 0, nd = cfg.nd > 0 ? cfg.nd : D;
	CbcaArgs A;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	A.p0 = cs.p0; A.p1 = cs.p1;
	A.vin = vin; A.vout = vout;
	A.D = D; A.H = H; A.W = W; A.direction = direction;
	A.d0 = d0; A.nd = nd;
	A.flags = route >= 0 ? cs.flag : nullptr;
	A.route = route;
	A.plan = cfg.plan;   // (CR_STRIP_IF_NO_LIST: the list header the launch looks at)
	A.lean_rb = cbca_lean_rows(D, H, W, cfg.lean_rb, cfg.lean_two_pass || route == CR_STRIP_IF_LIST);   // (mc_predict's lists are the two-pass records)
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
	// prefetch 2 rows, ring of 4 rows, 1 row of look-ahead, window form +-2 columns (+-4 measured slower at KITTI and 1000x1500)
	if (nt) hipLaunchKernelGGL((cbca_strip_kernel<2, 4, 1, 2, true>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	else hipLaunchKernelGGL((cbca_strip_kernel<2, 4, 1, 2, false>), dim3((unsigned)cdiv(waves, 4)), dim3(256), 0, st, A);
	return check_launch("cbca_strip");
}

}  // namespace mc
