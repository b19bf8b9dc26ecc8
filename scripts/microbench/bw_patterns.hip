// Microbenchmark (tuning aid, not product): what HBM bandwidth do the SGM access patterns reach on MI355X
// without any arithmetic?  Compares a flat float4 copy with wave-per-line streaming (1 KiB per wave-step).
//   hipcc --offload-arch=gfx950 -O3 bw_patterns.hip -o bw_patterns && ./bw_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) copy4(const float4 *in, float4 *out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// NIN input streams summed, one output stream, flat
template <int NIN> __global__ void __launch_bounds__(256) flat_rw(const float4 *a, const float4 *b, const float4 *c, float4 *out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		float4 v = a[i];
		if (NIN > 1) { float4 w = b[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
		if (NIN > 2) { float4 w = c[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
		out[i] = v;
	}
}
// wave per line; each step one pixel run of ds floats (lanes*4 floats); vertical => stride W*ds between steps
template <int NIN, int U> __global__ void __launch_bounds__(256) line_rw(const float *a, const float *b, const float *c, float *out,
                                                                       int H, int W, int ds, int vertical)
{
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
	const int nlines = vertical ? W : H, nsteps = vertical ? H : W;
	if (wave >= nlines) return;
	if (lane * 4 >= ds) return;
	const size_t base = vertical ? (size_t)wave * ds : (size_t)wave * W * ds;
	const size_t step = vertical ? (size_t)W * ds : (size_t)ds;
	float4 acc = make_float4(0, 0, 0, 0);
	for (int s0 = 0; s0 < nsteps; s0 += U) {
		float4 va[U], vb[U], vc[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = s0 + u < nsteps ? s0 + u : nsteps - 1;
			const size_t o = base + s * step + lane * 4;
			va[u] = *(const float4 *)(a + o);
			if (NIN > 1) vb[u] = *(const float4 *)(b + o);
			if (NIN > 2) vc[u] = *(const float4 *)(c + o);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			if (s0 + u >= nsteps) break;
			float4 v = va[u];
			if (NIN > 1) { v.x += vb[u].x; v.y += vb[u].y; v.z += vb[u].z; v.w += vb[u].w; }
			if (NIN > 2) { v.x += vc[u].x; v.y += vc[u].y; v.z += vc[u].z; v.w += vc[u].w; }
			acc.x = fminf(acc.x, v.x) + v.y;  // serial dependency between steps, like the recurrence
			v.x += acc.x;
			*(float4 *)(out + base + (size_t)(s0 + u) * step + lane * 4) = v;
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
	const int nvol = 2;  // both volumes in one launch, as mc_predict does: lines = nvol*H (buffers are nvol volumes long)
	const int ds = (D + 3) / 4 * 4;
	const size_t n = (size_t)nvol * H * W * ds;
	float *a, *b, *c, *o;
	CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&o, n * 4));
	CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
	const double GB = n * 4 / 1e9;
	printf("H=%d W=%d D=%d ds=%d, %d volumes: %.3f GB per stream\n", H, W, D, ds, nvol, GB);
	float ms;
	ms = timeit([&] { hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, (const float4 *)a, (float4 *)o, n / 4); }, 10);
	printf("flat copy 1R+1W          : %7.3f ms  %6.2f TB/s\n", ms, 2 * GB / ms);
	ms = timeit([&] { hipLaunchKernelGGL(flat_rw<2>, dim3(2048), dim3(256), 0, 0, (const float4 *)a, (const float4 *)b, (const float4 *)c, (float4 *)o, n / 4); }, 10);
	printf("flat 2R+1W               : %7.3f ms  %6.2f TB/s\n", ms, 3 * GB / ms);
	ms = timeit([&] { hipLaunchKernelGGL(flat_rw<3>, dim3(2048), dim3(256), 0, 0, (const float4 *)a, (const float4 *)b, (const float4 *)c, (float4 *)o, n / 4); }, 10);
	printf("flat 3R+1W               : %7.3f ms  %6.2f TB/s\n", ms, 4 * GB / ms);
	const int HH = nvol * H;  // horizontal: nvol*H lines of W steps ; vertical: treat as one image of nvol*H rows? no: W lines per volume
#define LINE(NIN, U, VERT, LABEL)                                                                                       \
	ms = timeit([&] {                                                                                                   \
		if (VERT) { for (int v = 0; v < nvol; ++v) hipLaunchKernelGGL((line_rw<NIN, U>), dim3((W * 64 + 255) / 256), dim3(256), 0, 0, \
		      a + (size_t)v * H * W * ds, b + (size_t)v * H * W * ds, c + (size_t)v * H * W * ds, o + (size_t)v * H * W * ds, H, W, ds, 1); } \
		else hipLaunchKernelGGL((line_rw<NIN, U>), dim3((HH * 64 + 255) / 256), dim3(256), 0, 0, a, b, c, o, HH, W, ds, 0);  \
	}, 10);                                                                                                             \
	printf("%-25s: %7.3f ms  %6.2f TB/s\n", LABEL, ms, (NIN + 1) * GB / ms);
	LINE(1, 4, 0, "line horiz 1R+1W U=4");
	LINE(1, 8, 0, "line horiz 1R+1W U=8");
	LINE(2, 4, 0, "line horiz 2R+1W U=4");
	LINE(3, 4, 0, "line horiz 3R+1W U=4");
	LINE(1, 4, 1, "line vert  1R+1W U=4 (2 launches)");
	LINE(2, 4, 1, "line vert  2R+1W U=4 (2 launches)");
	LINE(3, 4, 1, "line vert  3R+1W U=4 (2 launches)");
	LINE(3, 8, 1, "line vert  3R+1W U=8 (2 launches)");
	return 0;
}
