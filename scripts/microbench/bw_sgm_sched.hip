// Microbenchmark (tuning aid, not product): would another SCHEDULE of the four SGM sweeps close the horizontal launch's gap?
// bw_sgm_layout showed that the access shape is not what slows the horizontal launch (its data movement alone runs at the vertical launches'
// rate); the product's horizontal launch is 0.146 ms slower than its data movement at KITTI size (0.764 against 0.618) -- 1 480 long-lived
// waves on 1 024 SIMDs, each alternating between a chain of ~93 dependent instructions per step and its loads.  Same bytes, more waves at once:
//   today      launch 1: right + left (1R + 1W each)      launch 2: down (3R + 1W)      launch 3: up (2R + 1W)                 = 11 V
//   candidate  launch 1: right + left + DOWN, each 1R + 1W (L2 on its own: 3 932 waves)   launch 2: up, 4R + 1W                 = 11 V
// (the reference's sum order (((0 + L0) + L1) + L2) + L3 is kept by the last sweep reading the three partial volumes).  Each wave-step here
// carries a chain of WORK dependent v_min / v_add pairs, so that a lone wave is as busy per step as the product's.
//   hipcc --offload-arch=gfx950 -O3 bw_sgm_sched.hip -o bw_sgm_sched.bin && ./bw_sgm_sched.bin [H W D]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
	const float *c, *a, *a2, *a3;
	float *out, *out2, *out3;
	int H, W, ds, nvol, work;
	size_t vol;
};

// one line: DIRN 0 right, 1 left, 2 down, 3 up; NIN arrays read, one written
template <int NIN, int U>
__device__ __forceinline__ void line(const Args &A, int dirn, int v, int ln, float *outp, int lane)
{
	const int nsteps = dirn <= 1 ? A.W : A.H;
	if (lane * 4 >= A.ds) return;
	const float *c = A.c + v * A.vol, *a = A.a + v * A.vol, *a2 = A.a2 + v * A.vol, *a3 = A.a3 + v * A.vol;
	float *out = outp + v * A.vol;
	auto off = [&](int s) -> size_t {
		const int y = dirn == 0 || dirn == 1 ? ln : (dirn == 2 ? s : A.H - 1 - s);
		const int x = dirn == 0 ? s : (dirn == 1 ? A.W - 1 - s : ln);
		return ((size_t)y * A.W + x) * A.ds + lane * 4;
	};
	f4 rc[U], ra[U], rb[U], rd[U];
	auto load = [&](int u, int s) {
		const size_t o = off(s < nsteps ? s : nsteps - 1);
		rc[u] = __builtin_nontemporal_load((const f4 *)(c + o));
		if (NIN > 1) ra[u] = __builtin_nontemporal_load((const f4 *)(a + o));
		if (NIN > 2) rb[u] = __builtin_nontemporal_load((const f4 *)(a2 + o));
		if (NIN > 3) rd[u] = __builtin_nontemporal_load((const f4 *)(a3 + o));
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
				if (NIN > 3) t += rd[u];
				float w = carry + t.x;
				for (int k = 0; k < A.work; ++k) {   // the recurrence's chain: dependent instructions, as many as the product's step has
					w = fminf(w, t.y) + t.z;
					asm volatile("" : "+v"(w));
				}
				carry = w;
				t.x += carry;
				__builtin_nontemporal_store(t, (f4 *)(out + off(s)));
			}
			load(u, s + U);
		}
	}
}

// KIND 0: right + left (today's launch 1); 1: down 3R + 1W; 2: up 2R + 1W; 3: right + left + down, each 1R + 1W; 4: up 4R + 1W
template <int KIND, int U>
__global__ void __launch_bounds__(256) sweep(const Args A)
{
	const int lane = threadIdx.x & 63;
	int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
	const int nh = A.nvol * A.H, nv = A.nvol * A.W;
	if (KIND == 0 || KIND == 3) {
		if (wave < 2 * nh) {
			const bool second = wave >= nh;
			if (second) wave -= nh;
			line<1, 8>(A, second ? 1 : 0, wave / A.H, wave % A.H, second ? A.out2 : A.out, lane);
			return;
		}
		if (KIND == 0) return;
		wave -= 2 * nh;
		if (wave >= nv) return;
		line<1, U>(A, 2, wave / A.W, wave % A.W, A.out3, lane);
	} else {
		if (wave >= nv) return;
		if (KIND == 1) line<3, U>(A, 2, wave / A.W, wave % A.W, A.out, lane);
		if (KIND == 2) line<2, U>(A, 3, wave / A.W, wave % A.W, A.out, lane);
		if (KIND == 4) line<4, U>(A, 3, wave / A.W, wave % A.W, A.out, lane);
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
	const size_t vol = (size_t)H * W * ds;
	float *buf[7];
	for (int i = 0; i < 7; ++i) { CK(hipMalloc(&buf[i], nvol * vol * 4)); CK(hipMemset(buf[i], 0, nvol * vol * 4)); }
	const double V = (double)vol * 4 * nvol / 1e9;
	printf("H=%d W=%d D=%d ds=%d, %d volumes per launch: %.3f GB per stream; ms per launch\n", H, W, D, ds, nvol, V);
	printf("%-6s | %-34s | %-26s | %s\n", "work", "today: h 2x(1R+1W) + down 3R+1W + up 2R+1W", "candidate: h+down 3x(1R+1W) + up 4R+1W (U 8 / 16)", "sums");
	for (int work : {0, 20, 45, 70}) {
		Args A;
		A.c = buf[0]; A.a = buf[1]; A.a2 = buf[2]; A.a3 = buf[3]; A.out = buf[4]; A.out2 = buf[5]; A.out3 = buf[6];
		A.H = H; A.W = W; A.ds = ds; A.nvol = nvol; A.work = work; A.vol = vol;
		const int bh = (2 * nvol * H * 64 + 255) / 256, bv = (nvol * W * 64 + 255) / 256;
		const float t0 = timeit([&] { hipLaunchKernelGGL((sweep<0, 16>), dim3(bh), dim3(256), 0, 0, A); }, 10);
		const float t1 = timeit([&] { hipLaunchKernelGGL((sweep<1, 16>), dim3(bv), dim3(256), 0, 0, A); }, 10);
		const float t2 = timeit([&] { hipLaunchKernelGGL((sweep<2, 16>), dim3(bv), dim3(256), 0, 0, A); }, 10);
		const float t3 = timeit([&] { hipLaunchKernelGGL((sweep<3, 16>), dim3(bh + bv), dim3(256), 0, 0, A); }, 10);
		const float t4 = timeit([&] { hipLaunchKernelGGL((sweep<4, 8>), dim3(bv), dim3(256), 0, 0, A); }, 10);
		const float t5 = timeit([&] { hipLaunchKernelGGL((sweep<4, 16>), dim3(bv), dim3(256), 0, 0, A); }, 10);
		CK(hipGetLastError());
		printf("%-6d | %7.3f + %7.3f + %7.3f            | %7.3f + %7.3f / %7.3f     | %7.3f  vs  %7.3f / %7.3f\n", work, t0, t1, t2, t3, t4, t5, t0 + t1 + t2, t3 + t4, t3 + t5);
	}
	return 0;
}
