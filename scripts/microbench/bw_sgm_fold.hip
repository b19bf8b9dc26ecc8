// Microbenchmark (tuning aid, not product): could the LAST SGM sweep write the volume as (D,H,W) itself, so that the transpose behind it (and, with it,
// a third of the stand-alone layout changes at 1000 x 1500 x 256: VERDICT r4 #5) disappears?  The up sweep runs one wave per image column, holds all D
// values of one pixel per step and steps through the rows.  (D,H,W) wants x contiguous: a block of NC adjacent columns (NC waves) can exchange through LDS
// and write, per step, D pieces of NC floats -- 16 bytes for the 4 waves of today's blocks, 64 bytes for 16 waves -- instead of NC runs of D floats.
// Same reads (2R, as the up sweep), the write as
//   0  (H,W,ds) runs of 1 KB                                   (today; the transpose then costs its own read + write of the volume)
//   1  (D,H,W) pieces of NC floats through an LDS tile, NC = 4
//   2  ... NC = 16 (1 024 threads per block)
// and, for scale, the transpose kernel's own time is in the bench line (0.59 ms per 1.5 GB volume).
//   hipcc --offload-arch=gfx950 -O3 bw_sgm_fold.hip -o bw_sgm_fold.bin && ./bw_sgm_fold.bin [H W D]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
	const float *c, *a;
	float *out;
	int H, W, ds, nvol;
	size_t vol;
};

template <int NC, int U, bool FOLD>
__global__ void __launch_bounds__(64 * NC) up(const Args A)
{
	__shared__ float tile[FOLD ? NC : 1][FOLD ? 260 : 1];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int col = blockIdx.x * NC + wv;                 // column over both volumes
	const int v = col / A.W, x = col - v * A.W;
	const bool live = v < A.nvol;                          // (whole blocks stay: the exchange needs every wave at the barrier)
	const bool lok = lane * 4 < A.ds;
	const float *c = A.c + (live ? v : 0) * A.vol, *a = A.a + (live ? v : 0) * A.vol;
	float *out = A.out + (live ? v : 0) * A.vol;
	const int x0 = x - wv;                                 // first column of the block (the block's columns are adjacent and in one volume when W % NC == 0)
	const size_t HW = (size_t)A.H * A.W;
	auto off = [&](int s) -> size_t { return ((size_t)(A.H - 1 - s) * A.W + x) * A.ds + lane * 4; };
	f4 rc[U], ra[U];
	auto load = [&](int u, int s) {
		const size_t o = off(s < A.H ? s : A.H - 1);
		const bool ok = live && lok;
		rc[u] = ok ? __builtin_nontemporal_load((const f4 *)(c + o)) : f4{0, 0, 0, 0};
		ra[u] = ok ? __builtin_nontemporal_load((const f4 *)(a + o)) : f4{0, 0, 0, 0};
	};
#pragma unroll
	for (int u = 0; u < U; ++u) load(u, u);
	float carry = 0.0f;
	for (int g = 0; g < A.H; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			if (s < A.H) {
				f4 t = rc[u] + ra[u];
				carry = fminf(carry, t.x) + t.y;
				t.x += carry;
				if (!FOLD) {
					if (live && lok) __builtin_nontemporal_store(t, (f4 *)(out + off(s)));
				} else {
					// this wave's D values -> its row of the tile; then every thread writes pieces [d][y][x0 .. x0 + NC - 1]
					if (lok) *(f4 *)&tile[wv][lane * 4] = t;
					__syncthreads();
					const int y = A.H - 1 - s;
					for (int d = threadIdx.x; d < A.ds; d += 64 * NC) {   // (NC = 4: one piece per thread and step; NC = 16: a quarter of the threads)
						float *dst = out + (size_t)d * HW + (size_t)y * A.W + x0;
						if (NC == 4) {
							const f4 p = {tile[0][d], tile[1][d], tile[2][d], tile[3][d]};
							if (live) __builtin_nontemporal_store(p, (f4 *)dst);
						} else {
#pragma unroll
							for (int q = 0; q < NC; q += 4) {
								const f4 p = {tile[q][d], tile[q + 1][d], tile[q + 2][d], tile[q + 3][d]};
								if (live) __builtin_nontemporal_store(p, (f4 *)(dst + q));
							}
						}
					}
					__syncthreads();
				}
			}
			load(u, s + U);
		}
	}
}

template <typename F> float timeit(F f, int reps)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	f();
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) f();
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	return ms / reps;
}

int main(int argc, char **argv)
{
	const int H = argc > 1 ? atoi(argv[1]) : 1000, W = argc > 2 ? atoi(argv[2]) : 1504, D = argc > 3 ? atoi(argv[3]) : 256;   // (W a multiple of 16: blocks of adjacent columns)
	const int nvol = 2, ds = (D + 3) / 4 * 4;
	const size_t vol = (size_t)H * W * ds;
	float *buf[3];
	for (int i = 0; i < 3; ++i) { CK(hipMalloc(&buf[i], nvol * vol * 4)); CK(hipMemset(buf[i], 0, nvol * vol * 4)); }
	Args A; A.c = buf[0]; A.a = buf[1]; A.out = buf[2]; A.H = H; A.W = W; A.ds = ds; A.nvol = nvol; A.vol = vol;
	const double V = (double)vol * 4 * nvol / 1e9;
	printf("H=%d W=%d D=%d, %d volumes per launch, %.3f GB per stream; up sweep 2R + 1W, ms per launch (TB/s of 3 streams)\n", H, W, D, nvol, V);
	const float t0 = timeit([&] { hipLaunchKernelGGL((up<4, 16, false>), dim3(nvol * W / 4), dim3(256), 0, 0, A); }, 8);
	printf("write (H,W,ds) runs (today)                         : %7.3f  (%5.2f)\n", t0, 3 * V / t0);
	const float t1 = timeit([&] { hipLaunchKernelGGL((up<4, 16, true>), dim3(nvol * W / 4), dim3(256), 0, 0, A); }, 8);
	printf("write (D,H,W) pieces of 16 B, blocks of  4 columns : %7.3f  (%5.2f)\n", t1, 3 * V / t1);
	const float t2 = timeit([&] { hipLaunchKernelGGL((up<16, 8, true>), dim3(nvol * W / 16), dim3(1024), 0, 0, A); }, 8);
	printf("write (D,H,W) pieces of 64 B, blocks of 16 columns : %7.3f  (%5.2f)\n", t2, 3 * V / t2);
	CK(hipGetLastError());
	printf("(the transpose this would replace: 0.59 ms per 1.5 GB volume, i.e. %.2f ms for these %d volumes)\n", 0.59 * nvol * vol * 4 / 1.536e9, nvol);
	return 0;
}
