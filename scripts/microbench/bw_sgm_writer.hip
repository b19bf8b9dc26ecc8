// Microbenchmark (tuning aid, not product): the horizontal SGM launch with the stores taken OFF the line waves.
// profiles/r05_sgm_ablation.txt: the launch's time is its reads-time plus its writes-time, and on gfx9 a wave's loads and stores share
// one in-order counter (a wait for a prefetched cost run also waits for every store issued before it).  Candidate: a line wave writes
// each finished run to an LDS ring; one WRITER wave per block drains the rings of the block's line waves with 4-step pieces.
//   A  today's shape: 2 x nvol x H line waves, each 1R + 1W per step (U = 8 steps in flight, stores of 4 steps issued together)
//   B  blocks of LW line waves + 1 writer wave; rings of RING runs per line wave in LDS
// Each wave-step carries a chain of WORK dependent v_min / v_add pairs, so that a lone wave is as busy per step as the product's.
//   hipcc --offload-arch=gfx950 -O3 bw_sgm_writer.hip -o bw_sgm_writer.bin && ./bw_sgm_writer.bin [H W D]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

struct Args {
	const float *c;
	float *out, *out2;
	int H, W, ds, nvol, work;
	size_t vol;
};

__device__ __forceinline__ size_t pix_off(const Args &A, int dirn, int ln, int s, int lane)
{
	const int x = dirn == 0 ? s : A.W - 1 - s;
	return ((size_t)ln * A.W + x) * A.ds + lane * 4;
}

__device__ __forceinline__ f4 recur(const Args &A, f4 t, float &carry)
{
	float w = carry + t.x;
	for (int k = 0; k < A.work; ++k) {   // the recurrence's chain: dependent instructions, as many as the product's step has
		w = fminf(w, t.y) + t.z;
		asm volatile("" : "+v"(w));
	}
	carry = w;
	t.x += carry;
	return t;
}

// A: loads and stores in the line wave
template <int U>
__global__ void __launch_bounds__(256) sweep_a(const Args A)
{
	const int lane = threadIdx.x & 63;
	int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
	const int nh = A.nvol * A.H;
	if (wave >= 2 * nh) return;
	const bool second = wave >= nh;
	if (second) wave -= nh;
	const int dirn = second ? 1 : 0, v = wave / A.H, ln = wave % A.H, nsteps = A.W;
	if (lane * 4 >= A.ds) return;
	const float *c = A.c + v * A.vol;
	float *out = (second ? A.out2 : A.out) + v * A.vol;
	f4 rc[U], ob[4];
#pragma unroll
	for (int u = 0; u < U; ++u) rc[u] = __builtin_nontemporal_load((const f4 *)(c + pix_off(A, dirn, ln, u, lane)));
	float carry = 0.0f;
	for (int g = 0; g < nsteps; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			if (s < nsteps) ob[u & 3] = recur(A, rc[u], carry);
			if ((u & 3) == 3) {
#pragma unroll
				for (int k = 0; k < 4; ++k)
					if (s - 3 + k < nsteps) __builtin_nontemporal_store(ob[k], (f4 *)(out + pix_off(A, dirn, ln, s - 3 + k, lane)));
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int sn = s - 3 + k + U;
					rc[u - 3 + k] = __builtin_nontemporal_load((const f4 *)(c + pix_off(A, dirn, ln, sn < nsteps ? sn : nsteps - 1, lane)));
				}
			}
		}
	}
}

// B: LW line waves + one writer wave per block
template <int U, int LW, int RING>
__global__ void __launch_bounds__(64 * (LW + 1)) sweep_b(const Args A)
{
	__shared__ f4 ring[LW][RING][64];
	__shared__ int prod[LW], cons[LW];
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int nh = A.nvol * A.H, nsteps = A.W;
	if (threadIdx.x < LW) { prod[threadIdx.x] = 0; cons[threadIdx.x] = 0; }
	__syncthreads();
	auto line_of = [&](int l, int &dirn, int &v, int &ln, bool &live) {
		int wave = blockIdx.x * LW + l;
		live = wave < 2 * nh;
		const bool second = wave >= nh;
		if (second) wave -= nh;
		dirn = second ? 1 : 0;
		v = wave / A.H;
		ln = wave % A.H;
	};
	if (wv < LW) {
		int dirn, v, ln;
		bool live;
		line_of(wv, dirn, v, ln, live);
		if (!live) return;
		const float *c = A.c + v * A.vol;
		const bool act = lane * 4 < A.ds;
		f4 rc[U];
		const int ll = act ? lane : 0;
#pragma unroll
		for (int u = 0; u < U; ++u) rc[u] = __builtin_nontemporal_load((const f4 *)(c + pix_off(A, dirn, ln, u, ll)));
		float carry = 0.0f;
		int seen = 0;   // the writer's progress as last read
		for (int g = 0; g < nsteps; g += U) {
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int s = g + u;
				if ((u & 3) == 0) {
					// room for four more runs?  (the counter was read a batch ago; re-read only when that value says no)
					while (s + 4 - seen > RING) {
						seen = __builtin_amdgcn_readfirstlane(*(volatile int *)&cons[wv]);
						if (s + 4 - seen > RING) __builtin_amdgcn_s_sleep(2);
					}
				}
				if (s < nsteps) ring[wv][s % RING][lane] = recur(A, rc[u], carry);
				if ((u & 3) == 3) {
					if (lane == 0) *(volatile int *)&prod[wv] = s + 1 < nsteps ? s + 1 : nsteps;   // LDS executes a wave's operations in order: the runs are in before this
					seen = __builtin_amdgcn_readfirstlane(*(volatile int *)&cons[wv]);
#pragma unroll
					for (int k = 0; k < 4; ++k) {
						const int sn = s - 3 + k + U;
						rc[u - 3 + k] = __builtin_nontemporal_load((const f4 *)(c + pix_off(A, dirn, ln, sn < nsteps ? sn : nsteps - 1, ll)));
					}
				}
			}
		}
		return;
	}
	// the writer
	int done[LW], dirn[LW], vv[LW], lnn[LW];
	bool live[LW];
	int left = 0;
#pragma unroll
	for (int l = 0; l < LW; ++l) {
		line_of(l, dirn[l], vv[l], lnn[l], live[l]);
		done[l] = live[l] ? 0 : nsteps;
		left += live[l] ? 1 : 0;
	}
	const bool act = lane * 4 < A.ds;
	while (left > 0) {
		bool any = false;
#pragma unroll
		for (int l = 0; l < LW; ++l) {
			if (done[l] >= nsteps) continue;
			const int p = __builtin_amdgcn_readfirstlane(*(volatile int *)&prod[l]);
			const int n = p - done[l] >= 4 ? 4 : (p == nsteps ? p - done[l] : 0);
			if (n <= 0) continue;
			any = true;
			f4 t[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) t[k] = ring[l][(done[l] + (k < n ? k : 0)) % RING][lane];
			__builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the runs are in registers before the line wave may overwrite them
			if (lane == 0) *(volatile int *)&cons[l] = done[l] + n;
			float *out = (dirn[l] ? A.out2 : A.out) + vv[l] * A.vol;
#pragma unroll
			for (int k = 0; k < 4; ++k)
				if (k < n && act) __builtin_nontemporal_store(t[k], (f4 *)(out + pix_off(A, dirn[l], lnn[l], done[l] + k, lane)));
			done[l] += n;
			if (done[l] >= nsteps) --left;
		}
		if (!any) __builtin_amdgcn_s_sleep(8);
	}
}


// C: TWO waves per line, each owning half of the pixel's run (8 bytes per lane): twice the waves for the same bytes.  What a split recurrence would
// have to exchange per step -- its half's minimum and the two values at the seam -- goes through LDS with one block barrier per step (slots
// alternate by step parity).  The chain per wave is WORKC = 0.4 WORK (per-step part) + 0.3 WORK (per-value part, halved) long.
template <int U>
__global__ void __launch_bounds__(128) sweep_c(const Args A)
{
	__shared__ float xch[2][2][4];
	const int lane = threadIdx.x & 63, h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	int wave = blockIdx.x;
	const int nh = A.nvol * A.H, nsteps = A.W;
	if (wave >= 2 * nh) return;
	const bool second = wave >= nh;
	if (second) wave -= nh;
	const int dirn = second ? 1 : 0, v = wave / A.H, ln = wave % A.H;
	const int half = ((A.ds / 2) + 1) & ~1;            // floats per wave (even)
	const int f0 = h * half + lane * 2;                // this lane's first float inside the run
	const bool act = lane * 2 < half && f0 < A.ds;
	const float *c = A.c + v * A.vol;
	float *out = (second ? A.out2 : A.out) + v * A.vol;
	typedef float f2 __attribute__((ext_vector_type(2)));
	auto off = [&](int s) -> size_t {
		const int x = dirn == 0 ? s : A.W - 1 - s;
		return ((size_t)ln * A.W + x) * A.ds + (act ? f0 : 0);
	};
	f2 rc[U];
#pragma unroll
	for (int u = 0; u < U; ++u) rc[u] = __builtin_nontemporal_load((const f2 *)(c + off(u)));
	float carry = 0.0f;
	const int workc = (A.work * 7 + 9) / 10;
	for (int g = 0; g < nsteps; g += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = g + u;
			f2 t = rc[u];
			float w = carry + t.x;
			for (int k = 0; k < workc; ++k) {
				w = fminf(w, t.y) + t.x;
				asm volatile("" : "+v"(w));
			}
			// the exchange: this half's minimum (here: w) and seam values to the other wave, theirs back
			if (lane == 0) { xch[s & 1][h][0] = w; xch[s & 1][h][1] = t.x; xch[s & 1][h][2] = t.y; }
			__syncthreads();
			const float pw = xch[s & 1][1 - h][0], px = xch[s & 1][1 - h][1];
			carry = fminf(w, pw) + px * 0.0f;
			t.x += carry;
			if (s < nsteps && act) __builtin_nontemporal_store(t, (f2 *)(out + off(s)));
			const int sn = s + U;
			rc[u] = __builtin_nontemporal_load((const f2 *)(c + off(sn < nsteps ? sn : nsteps - 1)));
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
	const size_t vol = (size_t)H * W * ds;
	float *buf[3], *chk[2];
	for (int i = 0; i < 3; ++i) { CK(hipMalloc(&buf[i], nvol * vol * 4)); CK(hipMemset(buf[i], 0, nvol * vol * 4)); }
	for (int i = 0; i < 2; ++i) { CK(hipMalloc(&chk[i], nvol * vol * 4)); }
	// a non-trivial input, so that A and B can be compared
	{
		float *h = (float *)malloc(nvol * vol * 4);
		for (size_t i = 0; i < nvol * vol; ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) * 0.01f;
		CK(hipMemcpy(buf[0], h, nvol * vol * 4, hipMemcpyHostToDevice));
		free(h);
	}
	const double V = (double)vol * 4 * nvol / 1e9;
	printf("H=%d W=%d D=%d ds=%d, %d volumes: %.3f GB per stream, the launch moves %.3f GB; ms per launch\n", H, W, D, ds, nvol, V, 4 * V);
	printf("%-5s | %-9s | %-28s | %-28s | %-28s\n", "work", "A (today)", "B LW 3, ring 16 / 32", "B LW 7, ring 8 / 16", "B LW 1 / LW 15 ring 8");
	for (int work : {0, 5, 10, 20, 45}) {
		Args A;
		A.c = buf[0]; A.out = buf[1]; A.out2 = buf[2];
		A.H = H; A.W = W; A.ds = ds; A.nvol = nvol; A.work = work; A.vol = vol;
		const int nl = 2 * nvol * H;
		const float ta = timeit([&] { hipLaunchKernelGGL((sweep_a<8>), dim3((nl * 64 + 255) / 256), dim3(256), 0, 0, A); }, 10);
		CK(hipMemcpy(chk[0], buf[1], nvol * vol * 4, hipMemcpyDeviceToDevice));
		CK(hipMemcpy(chk[1], buf[2], nvol * vol * 4, hipMemcpyDeviceToDevice));
		CK(hipMemset(buf[1], 0, nvol * vol * 4)); CK(hipMemset(buf[2], 0, nvol * vol * 4));
		const float tb0 = timeit([&] { hipLaunchKernelGGL((sweep_b<8, 3, 16>), dim3((nl + 2) / 3), dim3(256), 0, 0, A); }, 10);
		// same results?
		{
			float *h0 = (float *)malloc(nvol * vol * 4), *h1 = (float *)malloc(nvol * vol * 4);
			size_t bad = 0;
			for (int k = 0; k < 2; ++k) {
				CK(hipMemcpy(h0, chk[k], nvol * vol * 4, hipMemcpyDeviceToHost));
				CK(hipMemcpy(h1, buf[1 + k], nvol * vol * 4, hipMemcpyDeviceToHost));
				for (size_t i = 0; i < nvol * vol; ++i) bad += h0[i] != h1[i];
			}
			if (bad) printf("  (B differs from A in %zu values)\n", bad);
			free(h0); free(h1);
		}
		const float tb1 = timeit([&] { hipLaunchKernelGGL((sweep_b<8, 3, 32>), dim3((nl + 2) / 3), dim3(256), 0, 0, A); }, 10);
		const float tb2 = timeit([&] { hipLaunchKernelGGL((sweep_b<8, 7, 8>), dim3((nl + 6) / 7), dim3(512), 0, 0, A); }, 10);
		const float tb3 = timeit([&] { hipLaunchKernelGGL((sweep_b<8, 7, 16>), dim3((nl + 6) / 7), dim3(512), 0, 0, A); }, 10);
		const float tb4 = timeit([&] { hipLaunchKernelGGL((sweep_b<8, 1, 16>), dim3(nl), dim3(128), 0, 0, A); }, 10);
		const float tb5 = timeit([&] { hipLaunchKernelGGL((sweep_b<8, 15, 8>), dim3((nl + 14) / 15), dim3(1024), 0, 0, A); }, 10);
		const float tc0 = timeit([&] { hipLaunchKernelGGL((sweep_c<8>), dim3(nl), dim3(128), 0, 0, A); }, 10);
		const float tc1 = timeit([&] { hipLaunchKernelGGL((sweep_c<16>), dim3(nl), dim3(128), 0, 0, A); }, 10);
		CK(hipGetLastError());
		printf("%-5d | %9.3f | %13.3f %13.3f  | %13.3f %13.3f  | %13.3f %13.3f | C (two waves per line, U 8 / 16) %8.3f %8.3f\n", work, ta, tb0, tb1, tb2, tb3, tb4, tb5, tc0, tc1);
	}
	return 0;
}
