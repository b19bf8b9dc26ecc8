// Microbenchmark (tuning aid): what does the CBCA tile access pattern itself cost on a (D,H,W) volume?
//   tilecopy<ORDER, P1, BAR>: the staging skeleton of cbca_tile_kernel -- per disparity each wave loads its 5 rows of a
//   (16+4) x 64 frame (+ the shifted packed-lengths row when P1), two disparities ahead, and writes 4 x 60 outputs.
//   ORDER 0: per-XCD runs, y-fastest (cbca.hip)   1: flat, x-fastest   2: per-XCD runs, x-fastest   3: flat, d-fastest
//   BAR: also go through LDS + __syncthreads like the real kernel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int TY = 16, LW = 64, HALO = 2, RY = TY + 2 * HALO, TXO = LW - 2 * HALO, NR = RY / 4;

template <int ORDER, bool P1, bool BAR, int ND>
__global__ void __launch_bounds__(256) tilecopy(const float *vin, const uint32_t *p1, float *vout, int D, int H, int W, int gx, int gy,
                                                int gz)
{
	__shared__ float Vt[2][RY * LW];
	__shared__ uint32_t Mt[2][RY * LW];
	const int tid = threadIdx.x, lx = tid & 63, wv = tid >> 6;
	const int ntiles = gx * gy * gz;
	const int b = blockIdx.x;
	int t, bx, by, bz;
	if (ORDER == 0 || ORDER == 2) {
		const int per = (ntiles + 7) >> 3;
		t = (b & 7) * per + (b >> 3);
	} else t = b;
	if (t >= ntiles) return;
	if (ORDER == 0) { by = t % gy; bx = (t / gy) % gx; bz = t / (gy * gx); }
	else if (ORDER == 3) { bz = t % gz; bx = (t / gz) % gx; by = t / (gz * gx); }
	else { bx = t % gx; by = (t / gx) % gy; bz = t / (gy * gx); }
	const int x0 = bx * TXO, y0 = by * TY, d0 = bz * ND, d1 = min(D, d0 + ND);
	const int ry0 = max(0, y0 - HALO), ry1 = min(H, y0 + TY + HALO), nrows = ry1 - ry0, hu = y0 - ry0;
	const int ty_n = min(TY, H - y0);
	const int xs = x0 - HALO + lx;
	const bool xs_in = xs >= 0 && xs < W;
	const bool out_lane = lx >= HALO && lx < HALO + TXO && xs < W;
	const int HWi = H * W;
	const unsigned OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rp1 = __builtin_amdgcn_make_buffer_rsrc((void *)p1, 0, HWi * 4, 0x00020000);
	unsigned voff[NR];
#pragma unroll
	for (int i = 0; i < NR; ++i) { const int r = wv + 4 * i; voff[i] = (r < nrows && xs_in) ? (unsigned)((ry0 + r) * W + xs) * 4u : OOB; }
	struct Stage { float pv[NR]; uint32_t pm[NR]; };
	auto fetch = [&](Stage &st, int d) {
		const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
		const bool pok = xs_in && xs - d >= 0;
#pragma unroll
		for (int i = 0; i < NR; ++i) {
			st.pv[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv, voff[i], 0, 0));
			if (P1) st.pm[i] = __builtin_amdgcn_raw_buffer_load_b32(rp1, pok ? voff[i] - (unsigned)(d * 4) : OOB, 0, 0);
			else st.pm[i] = 0;
		}
	};
	Stage sa, sb;
	fetch(sa, d0);
	if (d0 + 1 < d1) fetch(sb, d0 + 1);
	if (BAR) {
#pragma unroll
		for (int i = 0; i < NR; ++i) { const int r = wv + 4 * i; Vt[0][r * LW + lx] = sa.pv[i]; Mt[0][r * LW + lx] = sa.pm[i]; }
	}
	for (int d = d0; d < d1; ++d) {
		const int buf = (d - d0) & 1;
		Stage cur = sa;  // (no-LDS variant: the rows this wave loaded ARE rows wv+4i; it writes those that are output rows)
		if (d + 2 < d1) fetch(sa, d + 2);
		const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(vout + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
		if (BAR) {
			__syncthreads();
			if (out_lane) {
#pragma unroll
				for (int i = 0; i < TY / 4; ++i) {
					const int r = wv + 4 * i;
					if (r < ty_n) {
						const int o = (r + hu) * LW + lx;
						const float v = Vt[buf][o] + (Mt[buf][o] == 12345u ? 1.0f : 0.0f);
						__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, (unsigned)((y0 + r) * W + xs) * 4u, 0, 0);
					}
				}
			}
			if (d + 1 < d1) {
#pragma unroll
				for (int i = 0; i < NR; ++i) { const int r = wv + 4 * i; Vt[buf ^ 1][r * LW + lx] = sb.pv[i]; Mt[buf ^ 1][r * LW + lx] = sb.pm[i]; }
			}
		} else {
#pragma unroll
			for (int i = 0; i < NR; ++i) {
				const int r = wv + 4 * i - hu;  // output row index of staged row
				const bool ok = out_lane && r >= 0 && r < ty_n;
				const float v = cur.pv[i] + (cur.pm[i] == 12345u ? 1.0f : 0.0f);
				__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro, ok ? (unsigned)((y0 + r) * W + xs) * 4u : OOB, 0, 0);
			}
		}
#pragma unroll
		for (int i = 0; i < NR; ++i) { if (!BAR) { /* cur was sa */ } }
		// rotate: (sa holds d+2) ; sb -> next current
		if (!BAR) { Stage tmp = sb; sb = sa; sa = tmp; }
		else {
#pragma unroll
			for (int i = 0; i < NR; ++i) { sb.pv[i] = sa.pv[i]; sb.pm[i] = sa.pm[i]; }
		}
	}
}

// rows variant: a wave owns 256 consecutive columns (float4 per lane) and walks down `RB` rows of one disparity plane
template <int RB>
__global__ void __launch_bounds__(256) rowcopy(const float *vin, float *vout, int D, int H, int W)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int gxw = (W + 255) / 256;
	const int gyb = (H + RB - 1) / RB;
	int w = blockIdx.x * 4 + wv;
	const int cx = w % gxw; w /= gxw;
	const int cy = w % gyb; const int d = w / gyb;
	if (d >= D) return;
	const int x = cx * 256 + lane * 4;
	const int HWi = H * W;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(vout + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	typedef float f4 __attribute__((ext_vector_type(4)));
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	const int y0 = cy * RB, y1 = min(H, y0 + RB);
	for (int y = y0; y < y1; y += 4) {
		u4 v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, (y + k < y1 && x < W) ? (unsigned)((y + k) * W + x) * 4u : 0x80000000u, 0, 0);
#pragma unroll
		for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b128(v[k], ro, (y + k < y1 && x < W) ? (unsigned)((y + k) * W + x) * 4u : 0x80000000u, 0, 0);
	}
}


// strip variant: the skeleton of a wave-autonomous design -- a wave owns 256 staged columns (4 per lane, dwordx4, possibly
// only dword-aligned), walks RB rows of one plane, loads V / P0 / P1(shifted by d) per row and stores 252 columns.
//   ORDER 0: wave -> (strip fastest, row chunk, d)    1: (d fastest, strip, chunk)
//   ORDER 2: XCD x = d % 8; inside an XCD (d/8 fastest, strip, chunk)      3: XCD x = d % 8; inside (strip fastest, chunk, d/8)
template <int ORDER, bool PL>
__global__ void __launch_bounds__(256) stripcopy(const float *vin, const uint32_t *p0, const uint32_t *p1, float *vout, int D, int H,
                                                 int W, int RB, int gxs, int gyc, unsigned *chk)
{
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int cx, cy, d;
	if (ORDER == 0) { int w = blockIdx.x * 4 + wv; cx = w % gxs; w /= gxs; cy = w % gyc; d = w / gyc; }
	else if (ORDER == 1) { int w = blockIdx.x * 4 + wv; d = w % D; w /= D; cx = w % gxs; cy = w / gxs; }
	else {
		const int xcd = blockIdx.x & 7; int w = (blockIdx.x >> 3) * 4 + wv; const int D8 = D / 8;
		if (ORDER == 2) { d = (w % D8) * 8 + xcd; w /= D8; cx = w % gxs; cy = w / gxs; }
		else { cx = w % gxs; w /= gxs; cy = w % gyc; d = (w / gyc) * 8 + xcd; }
	}
	if (d >= D || cy >= gyc) return;
	const int xs = cx * 252 - 2 + lane * 4;
	const int HWi = H * W;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(vin + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(vout + (size_t)d * HWi), 0, HWi * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void *)p0, 0, HWi * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)p1, 0, HWi * 4, 0x00020000);
	const int y0 = cy * RB, y1 = min(H, y0 + RB);
	const bool in = xs >= 0 && xs + 3 < W;
	unsigned acc = 0;
	for (int y = y0; y < y1; y += 4) {
		u4 v[4], a[4], b[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const unsigned off = (y + k < y1 && in) ? (unsigned)((y + k) * W + xs) * 4u : 0x80000000u;
			v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, off, 0, 0);
			if (PL) {
				a[k] = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, 0);
				b[k] = __builtin_amdgcn_raw_buffer_load_b128(r1, (y + k < y1 && in && xs - d >= 0) ? off - (unsigned)d * 4u : 0x80000000u, 0, 0);
			}
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (PL) { acc += a[k].x ^ b[k].y; if ((a[k].z & b[k].w) == 0xdeadbeefu) v[k].x = 0; }
			__builtin_amdgcn_raw_buffer_store_b128(v[k], ro, (y + k < y1 && in) ? (unsigned)((y + k) * W + xs) * 4u : 0x80000000u, 0, 0);
		}
	}
	if (chk && acc == 0x12345u) chk[0] = acc;
}

// tiled variant: the same strip walk, but the volume is stored strip-major -- [d][strip][row][256 staged columns] --
// so that a wave streams ONE contiguous region (what a tiled internal layout between cbca iterations would give)
template <bool PL>
__global__ void __launch_bounds__(256) tiledcopy(const float *vin, const uint32_t *p0, const uint32_t *p1, float *vout, int D, int H,
                                                 int W, int RB, int gxs, int gyc, unsigned *chk)
{
	typedef unsigned u4 __attribute__((ext_vector_type(4)));
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int xcd = blockIdx.x & 7; int w = (blockIdx.x >> 3) * 4 + wv; const int D8 = D / 8;
	const int cx = w % gxs; w /= gxs; const int cy = w % gyc; const int d = (w / gyc) * 8 + xcd;
	if (d >= D || cy >= gyc) return;
	const int xs = cx * 248 - 4 + lane * 4;
	const int HWi = H * W;
	const size_t tile = (size_t)H * 256;   // floats per (d, strip)
	const float *tin = vin + ((size_t)d * gxs + cx) * tile;
	float *tout = vout + ((size_t)d * gxs + cx) * tile;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)tin, 0, (int)(tile * 4), 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)tout, 0, (int)(tile * 4), 0x00020000);
	const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void *)p0, 0, HWi * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void *)p1, 0, HWi * 4, 0x00020000);
	const int y0 = cy * RB, y1 = min(H, y0 + RB);
	const bool in = xs >= 0 && xs + 3 < W;
	unsigned acc = 0;
	for (int y = y0; y < y1; y += 4) {
		u4 v[4], a[4], b[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const unsigned toff = (y + k < y1) ? (unsigned)((y + k) * 256 + lane * 4) * 4u : 0x80000000u;
			const unsigned off = (y + k < y1 && in) ? (unsigned)((y + k) * W + xs) * 4u : 0x80000000u;
			v[k] = __builtin_amdgcn_raw_buffer_load_b128(rv, toff, 0, 0);
			if (PL) {
				a[k] = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, 0);
				b[k] = __builtin_amdgcn_raw_buffer_load_b128(r1, (y + k < y1 && in && xs - d >= 0) ? off - (unsigned)d * 4u : 0x80000000u, 0, 0);
			}
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (PL) { acc += a[k].x ^ b[k].y; if ((a[k].z & b[k].w) == 0xdeadbeefu) v[k].x = 0; }
			__builtin_amdgcn_raw_buffer_store_b128(v[k], ro, (y + k < y1) ? (unsigned)((y + k) * 256 + lane * 4) * 4u : 0x80000000u, 0, 0);
		}
	}
	if (chk && acc == 0x12345u) chk[0] = acc;
}

template <typename F> float timeit(F f, int reps)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f(); f();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) f();
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	return ms / reps;
}

int main()
{
	const int D = 256, H = 1000, W = 1500;
	const size_t n = (size_t)D * H * W;
	float *a, *o; uint32_t *p1;
	CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&o, n * 4)); CK(hipMalloc(&p1, (size_t)H * W * 4));
	CK(hipMemset(a, 0, n * 4)); CK(hipMemset(p1, 1, (size_t)H * W * 4));
	const double GB = 2.0 * n * 4 / 1e9;
	const int gx = (W + TXO - 1) / TXO, gy = (H + TY - 1) / TY;
#define RUN(ORDER, P1, BAR, ND) do { const int gz = (D + ND - 1) / ND; const int nt = gx * gy * gz; const int grid = (nt + 7) / 8 * 8; \
	float ms = timeit([&] { hipLaunchKernelGGL((tilecopy<ORDER, P1, BAR, ND>), dim3(grid), dim3(256), 0, 0, a, p1, o, D, H, W, gx, gy, gz); }, 5); \
	printf("tilecopy order=%d p1=%d lds+barrier=%d nd=%d: %.3f ms  %.0f GB/s (2V)\n", ORDER, P1, BAR, ND, ms, GB / ms * 1e3); } while (0)
	RUN(0, false, false, 8); RUN(1, false, false, 8); RUN(2, false, false, 8); RUN(3, false, false, 8);
	RUN(0, true, false, 8); RUN(1, true, false, 8);
	RUN(0, true, true, 8); RUN(1, true, true, 8); RUN(2, true, true, 8);
	RUN(0, true, true, 16); RUN(1, true, true, 32);
	{ float ms = timeit([&] { hipLaunchKernelGGL((rowcopy<40>), dim3((6 * 25 * D + 3) / 4), dim3(256), 0, 0, a, o, D, H, W); }, 5);
	  printf("rowcopy RB=40: %.3f ms  %.0f GB/s\n", ms, GB / ms * 1e3); }
	{ float ms = timeit([&] { hipLaunchKernelGGL((rowcopy<200>), dim3((6 * 5 * D + 3) / 4), dim3(256), 0, 0, a, o, D, H, W); }, 5);
	  printf("rowcopy RB=200: %.3f ms  %.0f GB/s\n", ms, GB / ms * 1e3); }

	{
		// correctness of dword-aligned (not 16B-aligned) dwordx4 buffer ops: the strip copy must reproduce the input
		float *h = (float *)malloc(n * 4 / 64); for (size_t i = 0; i < n / 64; ++i) h[i] = (float)(i % 100003);
		CK(hipMemcpy(a, h, n * 4 / 64, hipMemcpyHostToDevice)); CK(hipMemset(o, 0xff, n * 4 / 64));
		const int Ht = 1000, Wt = 1498, Dt = 4, RBt = 40, gxs = (Wt + 251) / 252, gyc = (Ht + RBt - 1) / RBt;  // W % 4 == 2: odd rows misaligned
		hipLaunchKernelGGL((stripcopy<0, false>), dim3((gxs * gyc * Dt + 3) / 4), dim3(256), 0, 0, a, p1, p1, o, Dt, Ht, Wt, RBt, gxs, gyc, (unsigned *)nullptr);
		float *g = (float *)malloc((size_t)Dt * Ht * Wt * 4); CK(hipMemcpy(g, o, (size_t)Dt * Ht * Wt * 4, hipMemcpyDeviceToHost));
		size_t bad = 0; for (size_t i = 0; i < (size_t)Dt * Ht * Wt; ++i) { const int x = (int)(i % Wt); if (x + 2 < Wt - 0 && g[i] != h[i] && x < (Wt / 4) * 4 - 2) ++bad; }
		printf("misaligned dwordx4 strip copy: %zu mismatches (interior columns)\n", bad);
		CK(hipMemset(a, 0, n * 4));
	}
#define RUNS(ORDER, PL, RB) do { const int gxs = (W + 251) / 252, gyc = (H + RB - 1) / RB; const int nw = gxs * gyc * D; \
	float ms = timeit([&] { hipLaunchKernelGGL((stripcopy<ORDER, PL>), dim3((nw + 3) / 4 + 8), dim3(256), 0, 0, a, p1, p1, o, D, H, W, RB, gxs, gyc, (unsigned *)nullptr); }, 5); \
	printf("stripcopy order=%d p-loads=%d RB=%d: %.3f ms  %.0f GB/s (2V)\n", ORDER, PL, RB, ms, GB / ms * 1e3); } while (0)
	RUNS(0, false, 40); RUNS(0, true, 40); RUNS(1, true, 40); RUNS(2, true, 40); RUNS(3, true, 40);
	RUNS(0, true, 100); RUNS(2, true, 100); RUNS(3, true, 100); RUNS(3, true, 200);
	{
		// tiled layout: 7 strips of 256 stored columns per row instead of 1500 (needs 1.2x the bytes: reuse the buffers for D*0.8 planes)
		const int Dt = 200, gxs = (W + 247) / 248;
		for (int RB : {40, 100}) {
			const int gyc = (H + RB - 1) / RB; const int nw = gxs * gyc * Dt;
			const double GBt = 2.0 * Dt * gxs * H * 256 * 4 / 1e9;
			float ms = timeit([&] { hipLaunchKernelGGL((tiledcopy<true>), dim3((nw + 3) / 4 + 8), dim3(256), 0, 0, a, p1, p1, o, Dt, H, W, RB, gxs, gyc, (unsigned *)nullptr); }, 5);
			printf("tiledcopy p-loads=1 RB=%d (D=%d): %.3f ms  %.0f GB/s of tile bytes, %.0f GB/s of useful (248/256) bytes\n", RB, Dt, ms, GBt / ms * 1e3, GBt / ms * 1e3 * 248 / 256);
			ms = timeit([&] { hipLaunchKernelGGL((tiledcopy<false>), dim3((nw + 3) / 4 + 8), dim3(256), 0, 0, a, p1, p1, o, Dt, H, W, RB, gxs, gyc, (unsigned *)nullptr); }, 5);
			printf("tiledcopy p-loads=0 RB=%d (D=%d): %.3f ms  %.0f GB/s of tile bytes\n", RB, Dt, ms, GBt / ms * 1e3);
		}
	}
	{ float ms = timeit([&] { CK(hipMemcpyAsync(o, a, n * 4, hipMemcpyDeviceToDevice, 0)); }, 5); printf("memcpy D2D: %.3f ms %.0f GB/s\n", ms, GB / ms * 1e3); }
	return 0;
}
