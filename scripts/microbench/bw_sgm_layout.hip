// Microbenchmark (tuning aid, not product): the access shape of the three SGM launches (sgm.hip: right+left sweeps concurrently,
// down sweep, up sweep; one scan line per wave64, a register ring of U steps in flight, non-temporal 16-byte accesses) WITHOUT the
// recurrence, on a volume laid out as ROW TILES -- (H / ht, W, ht, ds): the ht rows of a tile interleaved per pixel -- for ht = 1 (the
// product's (H, W, ds)), 4, 8, 16.  VERDICT r4 #3: with ht > 1 the horizontal launch's wave front is H / ht contiguous pieces of
// ht * ds floats instead of H runs 1.1 MB apart; the vertical launches pay for it (a column's next row is ds floats on inside a tile,
// then a jump of W * ht * ds).  Bytes per launch as in the product: horizontal 2 x (1R + 1W), down 3R + 1W, up 2R + 1W per volume.
//   hipcc --offload-arch=gfx950 -O3 bw_sgm_layout.hip -o bw_sgm_layout.bin && ./bw_sgm_layout.bin [H W D]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
	const float *c, *a, *a2;   // cost volume, running sum(s)
	float *out, *out2;
	int H, W, ds, ht, nvol;
	size_t vol;                // floats per volume (rows padded to whole tiles)
};

__device__ __forceinline__ size_t pix_off(const Args &A, int y, int x)
{
	const int t = y / A.ht, r = y - t * A.ht;
	return (((size_t)t * A.W + x) * A.ht + r) * A.ds;
}

// DIRN 0: right (waves [0, n)) and left (waves [n, 2n)) sweeps in one launch, 1R + 1W each; 2: down, NIN reads; 3: up
template <int DIRN, int NIN, int U>
__global__ void __launch_bounds__(256) sweep(const Args A)
{
	const int lane = threadIdx.x & 63;
	int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
	const int nlines = DIRN == 0 ? A.H : A.W, nsteps = DIRN == 0 ? A.W : A.H;
	const int nw = A.nvol * nlines;
	bool second = false;
	if (DIRN == 0) {
		if (wave >= 2 * nw) return;
		second = wave >= nw;
		if (second) wave -= nw;
	} else if (wave >= nw) return;
	const int v = wave / nlines, line = wave - v * nlines;
	if (lane * 4 >= A.ds) return;
	const float *c = A.c + v * A.vol, *a = A.a + v * A.vol, *a2 = A.a2 + v * A.vol;
	float *out = (second ? A.out2 : A.out) + v * A.vol;
	auto off = [&](int s) -> size_t {
		if (DIRN == 0) return pix_off(A, line, second ? A.W - 1 - s : s) + lane * 4;
		return pix_off(A, DIRN == 2 ? s : A.H - 1 - s, line) + lane * 4;
	};
	f4 rc[U], ra[U], rb[U];
	auto load = [&](int u, int s) {
		const size_t o = off(s < nsteps ? s : nsteps - 1);
		rc[u] = __builtin_nontemporal_load((const f4 *)(c + o));
		if (NIN > 1) ra[u] = __builtin_nontemporal_load((const f4 *)(a + o));
		if (NIN > 2) rb[u] = __builtin_nontemporal_load((const f4 *)(a2 + o));
	};
#pragma unroll
	for (int u = 0; u < U; ++u) load(u, u);
	float carry = 0.0f;
	for (int g = 0; g < nsteps; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			if (s < nsteps) {
				f4 t = rc[u];
				if (NIN > 1) t += ra[u];
				if (NIN > 2) t += rb[u];
				carry = fminf(carry, t.x) + t.y;   // a serial dependency between the steps, like the recurrence
				t.x += carry;
				__builtin_nontemporal_store(t, (f4 *)(out + off(s)));
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
	const int H = argc > 1 ? atoi(argv[1]) : 370, W = argc > 2 ? atoi(argv[2]) : 1226, D = argc > 3 ? atoi(argv[3]) : 228;
	const int nvol = 2, ds = (D + 3) / 4 * 4;
	const size_t volmax = (size_t)((H + 15) / 16 * 16) * W * ds;
	float *buf[5];
	for (int i = 0; i < 5; ++i) { CK(hipMalloc(&buf[i], nvol * volmax * 4)); CK(hipMemset(buf[i], 0, nvol * volmax * 4)); }
	const double V = (double)H * W * ds * 4 * nvol / 1e9;   // GB per stream, both volumes
	printf("H=%d W=%d D=%d ds=%d, %d volumes per launch: %.3f GB per stream;  ms per launch (TB/s)\n", H, W, D, ds, nvol, V);
	printf("%-6s %-22s %-22s %-22s %s\n", "ht", "horizontal 2x(1R+1W)", "down 3R+1W, U=16", "up 2R+1W, U=16", "sum ms");
	for (int ht : {1, 2, 4, 8, 16}) {
		Args A;
		A.c = buf[0]; A.a = buf[1]; A.a2 = buf[2]; A.out = buf[3]; A.out2 = buf[4];
		A.H = H; A.W = W; A.ds = ds; A.ht = ht; A.nvol = nvol;
		A.vol = (size_t)((H + ht - 1) / ht * ht) * W * ds;
		const float th = timeit([&] { hipLaunchKernelGGL((sweep<0, 1, 8>), dim3((2 * nvol * H * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		A.out = buf[4];   // (down / up write where the product does: over a running sum / the other buffer; any buffer serves here)
		const float td = timeit([&] { hipLaunchKernelGGL((sweep<2, 3, 16>), dim3((nvol * W * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		const float tu = timeit([&] { hipLaunchKernelGGL((sweep<3, 2, 16>), dim3((nvol * W * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		CK(hipGetLastError());
		printf("%-6d %7.3f (%5.2f)        %7.3f (%5.2f)        %7.3f (%5.2f)        %7.3f\n", ht, th, 4 * V / th, td, 4 * V / td, tu, 3 * V / tu, th + td + tu);
	}
	// the horizontal launch alone at other prefetch depths and with one wave per block (are 1 480 long-lived waves short of something?)
	for (int ht : {1, 4}) {
		Args A;
		A.c = buf[0]; A.a = buf[1]; A.a2 = buf[2]; A.out = buf[3]; A.out2 = buf[4];
		A.H = H; A.W = W; A.ds = ds; A.ht = ht; A.nvol = nvol;
		A.vol = (size_t)((H + ht - 1) / ht * ht) * W * ds;
		const float t4 = timeit([&] { hipLaunchKernelGGL((sweep<0, 1, 4>), dim3((2 * nvol * H * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		const float t16 = timeit([&] { hipLaunchKernelGGL((sweep<0, 1, 16>), dim3((2 * nvol * H * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		const float t32 = timeit([&] { hipLaunchKernelGGL((sweep<0, 1, 32>), dim3((2 * nvol * H * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		printf("ht=%d horizontal, U = 4 / 16 / 32: %7.3f (%5.2f)  %7.3f (%5.2f)  %7.3f (%5.2f)\n", ht, t4, 4 * V / t4, t16, 4 * V / t16, t32, 4 * V / t32);
	}
	return 0;
}
