// Microbenchmark (tuning aid): flat streaming variants -- what is the practical HBM ceiling on this MI355X?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

template <int UNR, int NT> __global__ void __launch_bounds__(256) copyk(const float4 *in, float4 *out, size_t n)
{
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
		float4 v[UNR];
#pragma unroll
		for (int u = 0; u < UNR; ++u) { if (NT) { f4v t = __builtin_nontemporal_load((const f4v *)(in + i + u * stride)); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = in[i + u * stride]; }
#pragma unroll
		for (int u = 0; u < UNR; ++u) { if (NT) { f4v t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, (f4v *)(out + i + u * stride)); } else out[i + u * stride] = v[u]; }
	}
	for (; i < n; i += stride) out[i] = in[i];
}
template <int UNR> __global__ void __launch_bounds__(256) readk(const float4 *in, float *out, size_t n)
{
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	float acc = 0;
	for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
		float4 v[UNR];
#pragma unroll
		for (int u = 0; u < UNR; ++u) v[u] = in[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
	}
	if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) writek(float4 *out, size_t n)
{
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = make_float4(1, 2, 3, 4);
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
int main(int argc, char **argv)
{
	const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 2.0) * (double)(1ull << 30)) / 4096 * 4096;
	const size_t n = bytes / 16;
	float4 *a, *o;
	CK(hipMalloc(&a, bytes)); CK(hipMalloc(&o, bytes));
	CK(hipMemset(a, 0, bytes));
	const double GB = bytes / 1e9;
	printf("%.2f GB per stream\n", GB);
	for (int grid : {2048, 16384}) {
		float ms;
		ms = timeit([&] { hipLaunchKernelGGL((copyk<1, 0>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 10);
		printf("grid %5d copy unr1    : %7.3f ms %6.2f TB/s\n", grid, ms, 2 * GB / ms);
		ms = timeit([&] { hipLaunchKernelGGL((copyk<4, 0>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 10);
		printf("grid %5d copy unr4    : %7.3f ms %6.2f TB/s\n", grid, ms, 2 * GB / ms);
		ms = timeit([&] { hipLaunchKernelGGL((copyk<8, 0>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 10);
		printf("grid %5d copy unr8    : %7.3f ms %6.2f TB/s\n", grid, ms, 2 * GB / ms);
		ms = timeit([&] { hipLaunchKernelGGL((copyk<4, 1>), dim3(grid), dim3(256), 0, 0, a, o, n); }, 10);
		printf("grid %5d copy unr4 nt : %7.3f ms %6.2f TB/s\n", grid, ms, 2 * GB / ms);
		ms = timeit([&] { hipLaunchKernelGGL((readk<4>), dim3(grid), dim3(256), 0, 0, a, (float *)o, n); }, 10);
		printf("grid %5d read  unr4    : %7.3f ms %6.2f TB/s\n", grid, ms, GB / ms);
		ms = timeit([&] { hipLaunchKernelGGL(writek, dim3(grid), dim3(256), 0, 0, o, n); }, 10);
		printf("grid %5d write         : %7.3f ms %6.2f TB/s\n", grid, ms, GB / ms);
	}
	float ms = timeit([&] { CK(hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
	printf("hipMemcpy D2D          : %7.3f ms %6.2f TB/s\n", ms, 2 * GB / ms);
	return 0;
}
