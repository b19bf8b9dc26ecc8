// Cross-based cost aggregation (adcensus.cu:343-377) for real-scene arm statistics: LDS tiles, supports sorted by height
// inside a tile, vertical run sharing.
//
// The reference's region sum is ONE serial chain of fp32 additions per output -- rows ascending, inside a row x ascending --
// and that order is kept (cbca.hip, parity rule).  What can be shared without touching the order: the horizontal run of
// support row yy in column x,  [x - min(L0[yy,x], L1[yy,xp]), x + min(R0[yy,x], R1[yy,xp])],  depends on (d, yy, x)
// only, not on the output row y.  So the outputs (y, x), (y+1, x), ... of one COLUMN walk the same runs, each from its own
// first row to its own last row.  A lane therefore owns an ITEM = one column x 4 consecutive output rows: it walks the rows
// from the topmost first row to the bottommost last row of its four outputs, fetches every run value ONCE from LDS and
// adds it to all four accumulators; an accumulator is reset to +0.0 at its output's first row and read out after its last
// row -- what it collected outside its own rows never reaches a result.  A value beyond a lane's own run length enters as
// -0.0f (x + -0.0f == x exactly), so the lanes of a wave walk rows of different run lengths in lockstep.  Per tap and wave:
// one ds_read_b32, one compare + select, four additions.
//
// A block stages one tile of ONE disparity plane (TW x TH outputs + the arm halo) in LDS: values, per staged pixel the
// run (4 * left, length), per output its (up, down).  A wave runs at the pace of its tallest item and longest runs, and
// real scenes mix 3 x 3 supports with flat regions of (2 L1 - 1)^2 taps: the tile's items are SORTED by height (counting
// sort in LDS, a few instructions per item) and the waves of the block pull 64-item chunks, tallest first, from a shared
// counter.  Results go to an LDS tile and leave as full rows.
#include "cbca_common.h"
#include <type_traits>

namespace mc {

namespace {

template <int A, int TW, int TH>
struct TileGeo {
	static constexpr int AH = (A + 3) & ~3;           // horizontal halo, a multiple of 4 columns (16-byte rows)
	static constexpr int SW = TW + 2 * AH;            // staged columns
	static constexpr int SH = TH + 2 * A;             // staged rows
	static constexpr int NI = TW * (TH / 4);          // items: column x 4 rows
	static constexpr int NKEY = 2 * A + 4 + 1;        // item heights 0 .. 2A + 4
	static constexpr int V_BYTES = SH * SW * 4;
	static constexpr int M_BYTES = (SH * TW * 2 + 15) & ~15;   // runs: output columns only (a run is looked up in the item's own column)
	static constexpr int UD_BYTES = TH * TW * 2;
	static constexpr int OUT_BYTES = TH * TW * 4;
	static constexpr int TAB_BYTES = NI * 2;
	static constexpr int MISC_BYTES = 2 * 64 * 4 + 16;
	static constexpr int LDS_BYTES = V_BYTES + M_BYTES + UD_BYTES + OUT_BYTES + TAB_BYTES + MISC_BYTES;
};

// NQ tap slots of a row: every lane fetches NQ values from its run's start (the values behind its own run are fetched
// and never added), then  sum_j += t < n ? v[t] : -0.0f  for its four accumulators
template <int NQ>
__device__ __forceinline__ void tile_taps(const float *__restrict__ p, int n, float (&sum)[4])
{
	float v[NQ];
#pragma unroll
	for (int t = 0; t < NQ; ++t) v[t] = p[t];
#pragma unroll
	for (int t = 0; t < NQ; ++t) {
		const float tv = t < n ? v[t] : -0.0f;
		sum[0] += tv; sum[1] += tv; sum[2] += tv; sum[3] += tv;
	}
}

}  // namespace

template <int A, int TW, int TH, int NWAVES, bool NT>
__global__ void __launch_bounds__(64 * NWAVES) cbca_tile_kernel(const CbcaArgs P, const int tiles_x, const int tiles_per_plane,
                                                                const int gate)
{
	using G = TileGeo<A, TW, TH>;
	constexpr int AH = G::AH, SW = G::SW, SH = G::SH, NI = G::NI, NKEY = G::NKEY;
	constexpr int NTHREADS = 64 * NWAVES;
	constexpr int VOL_AUX = NT ? 2 : 0;
	static_assert(TH % 4 == 0 && TW % 4 == 0 && TW <= 256 && TH / 4 <= 256 && NKEY <= 64, "tile geometry");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	float *__restrict__ Vl = (float *)smem;
	unsigned short *__restrict__ Ml = (unsigned short *)(smem + G::V_BYTES);
	unsigned short *__restrict__ UDl = (unsigned short *)(smem + G::V_BYTES + G::M_BYTES);
	float *__restrict__ OUTl = (float *)(smem + G::V_BYTES + G::M_BYTES + G::UD_BYTES);
	unsigned short *__restrict__ TABl = (unsigned short *)(smem + G::V_BYTES + G::M_BYTES + G::UD_BYTES + G::OUT_BYTES);
	cb_u32 *__restrict__ HISTl = (cb_u32 *)(smem + G::V_BYTES + G::M_BYTES + G::UD_BYTES + G::OUT_BYTES + G::TAB_BYTES);
	cb_u32 *__restrict__ BASEl = HISTl + 64;
	cb_u32 *__restrict__ CTRl = BASEl + 64;   // [0] next chunk, [1] / [2] the tile's largest up / down arm

	// cbca_pack's flags (arm bound unknown to the caller): [0] an arm saturated the packed form, [1] an arm > 4, [2] an arm > 13.
	// gate bit 0: run only if no arm > 4; bit 1: only if some arm > 4; bit 2: only if no arm > 13
	if (P.overflow) {
		if (P.overflow[0]) return;
		if ((gate & 1) && P.overflow[1]) return;
		if ((gate & 2) && !P.overflow[1]) return;
		if ((gate & 4) && P.overflow[2]) return;
	}
	const int tid = threadIdx.x, lane = tid & 63;
	const int H = P.H, W = P.W;
	const int HWi = H * W;
	// block -> (plane, tile): the blocks of one XCD (blockIdx % 8) take every 8th plane and walk its tiles in row-major
	// order, so the halo a tile shares with its neighbours is met in that XCD's L2
	const int xcd = blockIdx.x & 7, s = blockIdx.x >> 3;
	const int d = P.d0 + (s / tiles_per_plane) * 8 + xcd;
	if (d >= P.d0 + P.nd) return;
	const int tin = s % tiles_per_plane;
	const int tx0 = (tin % tiles_x) * TW, ty0 = (tin / tiles_x) * TH;
	const int sh = d * P.direction;
	const int sx0 = tx0 - AH, sy0 = ty0 - A;
	const cb_u32 OOB = 0x80000000u;
	const int plane_bytes = HWi * 4;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(P.vin + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(P.vout + (size_t)d * HWi), 0, plane_bytes, 0x00020000);
	const int padded_bytes = (HWi + 2 * CS_PAD) * 4;
	const __amdgpu_buffer_rsrc_t rp0 = __builtin_amdgcn_make_buffer_rsrc((void *)(P.p0 - CS_PAD), 0, padded_bytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)(P.p1 - CS_PAD), 0, padded_bytes, 0x00020000);

	if (tid < 64) HISTl[tid] = 0;
	if (tid == 0) { CTRl[0] = 0; CTRl[1] = 0; CTRl[2] = 0; }
	__syncthreads();

	// ---- stage the tile ----------------------------------------------------------------------------------------------
	// (a) the arm lengths of the OUTPUT rows: runs, (up, down), and the tile's largest up / down arm -- only the rows some
	// output of this tile reaches are staged in (b): textured tiles stage a few halo rows, flat ones up to A on either side
	constexpr int UB = 4;          // loads in flight per thread and batch
	constexpr int APR = TW / 4;    // 4-column units per row of output columns
	auto col_ok = [&](int xc) { return xc >= 0 && xc < W && xc + sh >= 0 && xc + sh < W; };   // pixel exists, partner inside the image (adcensus.cu:353)
	auto arms_rows = [&](int rfirst, int nrows, int rskip0, int rskip1, bool outputs) {   // staged rows [rfirst, rfirst + nrows) except [rskip0, rskip1)
		int umax = 0, dmax = 0;
		const int nskip = max(0, rskip1 - rskip0);
		const int n = (nrows - nskip) * APR;
		for (int q0 = tid; q0 < n; q0 += NTHREADS * UB) {
			cb_u4 a[UB], b[UB];
#pragma unroll
			for (int k = 0; k < UB; ++k) {
				const int q = q0 + k * NTHREADS;
				int r = rfirst + q / APR;
				if (r >= rskip0) r += nskip;
				const int u = q % APR;
				const int y = sy0 + r, x = tx0 + 4 * u;
				const bool rok = q < n && y >= 0 && y < H;
				const int base = y * W + x;
				// the padded scratch makes any in-row start readable; columns outside the image / the shifted range are masked below
				a[k] = __builtin_amdgcn_raw_buffer_load_b128(rp0, rok ? (cb_u32)(base + CS_PAD) * 4u : OOB, 0, 0);
				b[k] = __builtin_amdgcn_raw_buffer_load_b128(rp1, rok ? (cb_u32)(base + sh + CS_PAD) * 4u : OOB, 0, 0);
			}
#pragma unroll
			for (int k = 0; k < UB; ++k) {
				const int q = q0 + k * NTHREADS;
				if (q >= n) continue;
				int r = rfirst + q / APR;
				if (r >= rskip0) r += nskip;
				const int u = q % APR;
				const int y = sy0 + r, x = tx0 + 4 * u;
				const bool rok = y >= 0 && y < H;
				const cb_u32 mm[4] = {bytemin4(a[k].x, b[k].x), bytemin4(a[k].y, b[k].y), bytemin4(a[k].z, b[k].z), bytemin4(a[k].w, b[k].w)};
				cb_u32 run[4], ud[4];
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					const bool ok = rok && col_ok(x + t);
					const cb_u32 l = mm[t] & 0xffu, rr = (mm[t] >> 8) & 0xffu;
					run[t] = ok ? ((4u * l) | ((l + rr + 1u) << 8)) : 0u;
					ud[t] = ok ? (mm[t] >> 16) : 0xffffu;
					if (outputs && ok) {
						umax = max(umax, (int)((mm[t] >> 16) & 0xffu));
						dmax = max(dmax, (int)(mm[t] >> 24));
					}
				}
				*(cb_u2 *)(Ml + r * TW + 4 * u) = cb_u2{run[0] | (run[1] << 16), run[2] | (run[3] << 16)};
				if (outputs) *(cb_u2 *)(UDl + (r - A) * TW + 4 * u) = cb_u2{ud[0] | (ud[1] << 16), ud[2] | (ud[3] << 16)};
			}
		}
		if (outputs) {
			if (umax) atomicMax(&CTRl[1], (cb_u32)umax);
			if (dmax) atomicMax(&CTRl[2], (cb_u32)dmax);
		}
	};
	arms_rows(A, TH, 0, 0, true);
	__syncthreads();
	const int Umax = min((int)CTRl[1], A), Dmax = min((int)CTRl[2], A);
	// (b) values of the rows the tile's outputs reach, lengths of the halo rows among them
	arms_rows(A - Umax, Umax + TH + Dmax, A, A + TH, false);
	{
		constexpr int UPR = SW / 4;   // 4-column units per staged row
		const int r0 = A - Umax, n = (Umax + TH + Dmax) * UPR;
		for (int q0 = tid; q0 < n; q0 += NTHREADS * UB) {
			cb_u4 v[UB];
#pragma unroll
			for (int k = 0; k < UB; ++k) {
				const int q = q0 + k * NTHREADS;
				const int r = r0 + q / UPR, u = q % UPR;
				const int y = sy0 + r, x = sx0 + 4 * u;
				const bool rok = q < n && y >= 0 && y < H;
				const int base = y * W + x;
				if (x >= 0 && x + 3 < W) {
					v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, rok ? (cb_u32)base * 4u : OOB, 0, VOL_AUX);
				} else {
					cb_u32 t[4];
#pragma unroll
					for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_raw_buffer_load_b32(rv, (rok && x + e >= 0 && x + e < W) ? (cb_u32)(base + e) * 4u : OOB, 0, 0);
					v[k] = cb_u4{t[0], t[1], t[2], t[3]};
				}
			}
#pragma unroll
			for (int k = 0; k < UB; ++k) {
				const int q = q0 + k * NTHREADS;
				if (q >= n) continue;
				const int r = r0 + q / UPR, u = q % UPR;
				*(cb_u4 *)(Vl + r * SW + 4 * u) = v[k];
				const int orow = r - A, ocol = 4 * u - AH;
				if (orow >= 0 && orow < TH && ocol >= 0 && ocol < TW) *(cb_u4 *)(OUTl + orow * TW + ocol) = v[k];   // adcensus.cu:353-354: outputs without a partner are copied through
			}
		}
	}
	__syncthreads();

	// ---- items sorted by height (tallest first): counting sort --------------------------------------------------------
	// item i = (column c, row group g): outputs rows 4g .. 4g+3 of column c; height = rows from the topmost first row to the
	// bottommost last row of its outputs that have a partner
	auto item_rows = [&](int c, int g, int (&s0)[4], int (&e0)[4], int &top, int &bot) {
		top = 1 << 20; bot = -1;
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const cb_u32 ud = UDl[(4 * g + j) * TW + c];
			const bool ok = ud != 0xffffu;
			const int up = (int)(ud & 0xffu), dn = (int)(ud >> 8);
			s0[j] = ok ? 4 * g + j + A - up : 1 << 20;    // staged row of the output's first / last support row
			e0[j] = ok ? 4 * g + j + A + dn : -1;
			top = min(top, s0[j]);
			bot = max(bot, e0[j]);
		}
	};
	constexpr int IPT = (NI + NTHREADS - 1) / NTHREADS;
	cb_u32 keyrank[IPT];
#pragma unroll
	for (int k = 0; k < IPT; ++k) {
		const int i = tid + k * NTHREADS;
		keyrank[k] = 0;
		if (i < NI) {
			const int c = i % TW, g = i / TW;
			int s0[4], e0[4], top, bot;
			item_rows(c, g, s0, e0, top, bot);
			const int ext = bot >= top ? bot - top + 1 : 0;
			const cb_u32 rank = atomicAdd(&HISTl[ext], 1u);
			keyrank[k] = (cb_u32)ext | (rank << 8);
		}
	}
	__syncthreads();
	if (tid < 64) {   // first position of every height, tallest first
		cb_u32 below = 0;
		for (int k = NKEY - 1; k > tid; --k) below += HISTl[k];
		if (tid < NKEY) BASEl[tid] = below;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < IPT; ++k) {
		const int i = tid + k * NTHREADS;
		if (i < NI) {
			const int c = i % TW, g = i / TW;
			const cb_u32 key = keyrank[k] & 0xffu, rank = keyrank[k] >> 8;
			TABl[BASEl[key] + rank] = (unsigned short)(c | (g << 8));
		}
	}
	__syncthreads();
	const int nz = NI - (int)HISTl[0];          // items with at least one output to compute
	const int nchunks = (nz + 63) >> 6;

	// ---- chunks of 64 items, tallest first --------------------------------------------------------------------------
	for (;;) {
		int chunk = 0;
		if (lane == 0) chunk = (int)atomicAdd(&CTRl[0], 1u);
		chunk = __builtin_amdgcn_readfirstlane(chunk);
		if (chunk >= nchunks) break;
		const int idx = chunk * 64 + lane;
		const bool has = idx < nz;
		const cb_u32 ent = TABl[has ? idx : 0];
		const int c = (int)(ent & 0xffu), g = (int)(ent >> 8);
		int s0[4], e0[4], top, bot;
		item_rows(c, g, s0, e0, top, bot);
		const int ext = has ? bot - top + 1 : 0;
		const int E = __builtin_amdgcn_readfirstlane(ext);   // lane 0 holds the chunk's tallest item
		int srel[4], erel[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			srel[j] = has ? s0[j] - top : 1 << 20;
			erel[j] = has ? e0[j] - top : -1;
		}
		float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
		int cb[4] = {0, 0, 0, 0}, cnt[4] = {1, 1, 1, 1};
		int Pn = 0;
		int rowoff = top * SW + c + AH, mrow = top * TW + c;
		for (int i = 0; i < E; ++i, rowoff += SW, mrow += TW) {
			const bool act = i < ext;
			const cb_u32 m = act ? (cb_u32)Ml[mrow] : 0u;
			const int n = (int)(m >> 8);
			const float *__restrict__ p = (const float *)((const char *)(Vl + rowoff) - (m & 0xffu));
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const bool st = i == srel[j];   // the output's first row: its chain starts from +0.0 here
				sum[j] = st ? 0.0f : sum[j];
				cb[j] = st ? Pn : cb[j];
			}
			if (A > 4 && __any(n > 9)) {
				if (__any(n > 18)) tile_taps<27>(p, n, sum);
				else if (__any(n > 13)) tile_taps<18>(p, n, sum);
				else tile_taps<13>(p, n, sum);
			} else if (__any(n > 5)) {
				tile_taps<9>(p, n, sum);
			} else if (__any(n > 3)) {
				tile_taps<5>(p, n, sum);
			} else {
				tile_taps<3>(p, n, sum);
			}
			Pn += n;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const bool en = i == erel[j];   // the output's last row
				res[j] = en ? sum[j] : res[j];
				cnt[j] = en ? Pn - cb[j] : cnt[j];
			}
		}
#pragma unroll
		for (int j = 0; j < 4; ++j)
			if (erel[j] >= 0) OUTl[(4 * g + j) * TW + c] = res[j] / (float)cnt[j];
	}
	__syncthreads();

	// ---- results leave as rows ---------------------------------------------------------------------------------------
	constexpr int OPR = TW / 4;
	for (int q = tid; q < TH * OPR; q += NTHREADS) {
		const int r = q / OPR, u = q - r * OPR;
		const int y = ty0 + r, x = tx0 + 4 * u;
		if (y >= H || x >= W) continue;
		const cb_f4 o = *(const cb_f4 *)(OUTl + r * TW + 4 * u);
		const int ob = y * W + x;
		if (x + 3 < W) {
			__builtin_amdgcn_raw_buffer_store_b128(cb_u4{__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)}, ro,
			                                       (cb_u32)ob * 4u, 0, VOL_AUX);
		} else {
			const float oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
			for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oo[k]), ro, x + k < W ? (cb_u32)(ob + k) * 4u : OOB, 0, 0);
		}
	}
}

template <int A, int TW, int TH, int NWAVES>
static int cbca_tiles_launch(const CbcaArgs &P, bool nt, int gate, hipStream_t st)
{
	using G = TileGeo<A, TW, TH>;
	const int tiles_x = (int)cdiv(P.W, TW), tiles_y = (int)cdiv(P.H, TH);
	const int tpp = tiles_x * tiles_y;
	const int64_t blocks = (int64_t)cdiv(P.nd, 8) * 8 * tpp;
	if (blocks > 0x7fffffff) {
		set_error("cbca_tiles: %lld blocks", (long long)blocks);
		return MC_EINVAL;
	}
	auto kern_nt = cbca_tile_kernel<A, TW, TH, NWAVES, true>;
	auto kern = cbca_tile_kernel<A, TW, TH, NWAVES, false>;
	static bool attr_done = false;   // (idempotent; a race sets the same value twice)
	if (!attr_done) {
		(void)hipFuncSetAttribute((const void *)kern_nt, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
		(void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
		attr_done = true;
	}
	if (nt) hipLaunchKernelGGL(kern_nt, dim3((unsigned)blocks), dim3(64 * NWAVES), G::LDS_BYTES, st, P, tiles_x, tpp, gate);
	else hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NWAVES), G::LDS_BYTES, st, P, tiles_x, tpp, gate);
	return check_launch("cbca_tile");
}

// arm_class 4: every arm <= 4 (L1 <= 5); 13: every arm <= 13 (L1 <= 14).  gate != 0 (arm bound unknown to the caller): the
// launch stands down unless cbca_pack's flags say its arm class holds (bits: see the kernel).
int cbca_tiles(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int arm_class, int gate,
               hipStream_t st, const CbcaCfg &cfg)
{
	const int d0 = cfg.nd > 0 ? cfg.d0 : 0, nd = cfg.nd > 0 ? cfg.nd : D;
	CbcaArgs P;
	const CbcaScratch cs = cbca_scratch(packed, H, W);
	P.p0 = cs.p0; P.p1 = cs.p1;
	P.vin = vin; P.vout = vout;
	P.D = D; P.H = H; P.W = W; P.direction = direction;
	P.d0 = d0; P.nd = nd;
	P.rb = 0; P.by_arm = 0; P.gx = P.gy = 0;
	P.overflow = gate ? cs.flag : nullptr;
	const bool nt = cfg.nt >= 0 ? cfg.nt != 0 : (int64_t)nd * H * W * 4 > ((int64_t)768 << 20);
	// cfg.rb selects the tile geometry (test / tuning hook; 0 = the product's choice)
	if (arm_class <= 4) {
		switch (cfg.rb) {
		case 1: return cbca_tiles_launch<4, 128, 32, 8>(P, nt, gate, st);
		case 2: return cbca_tiles_launch<4, 128, 32, 4>(P, nt, gate, st);
		case 3: return cbca_tiles_launch<4, 256, 16, 8>(P, nt, gate, st);
		default: return cbca_tiles_launch<4, 128, 16, 4>(P, nt, gate, st);
		}
	}
	switch (cfg.rb) {
	case 1: return cbca_tiles_launch<13, 128, 32, 8>(P, nt, gate, st);
	case 2: return cbca_tiles_launch<13, 128, 32, 4>(P, nt, gate, st);
	case 3: return cbca_tiles_launch<13, 128, 16, 8>(P, nt, gate, st);
	default: return cbca_tiles_launch<13, 128, 16, 4>(P, nt, gate, st);
	}
}

}  // namespace mc
