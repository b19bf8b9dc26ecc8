// Microbenchmark (tuning aid, not product): streaming bandwidth against WORKING-SET size on MI355X.
// A float4 copy ping-pongs between two buffers of S bytes each (working set 2S); for 2S well below the 256 MB
// Infinity Cache the next launch finds its input on die.  Answers: what does blocking a multi-iteration stencil
// (CBCA: 18 iterations over the same volume) into cache-sized slabs buy, and what is the ceiling of a plain copy.
//   hipcc --offload-arch=gfx950 -O3 bw_sizes.hip -o bw_sizes.bin && ./bw_sizes.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NT> __global__ void __launch_bounds__(256) copyk(const f4v *__restrict__ in, f4v *__restrict__ out, size_t n)
{
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
		else out[i] = in[i];
	}
}
template <int NT> __global__ void __launch_bounds__(256) readk(const f4v *__restrict__ in, float *out, size_t n)
{
	const size_t stride = (size_t)gridDim.x * 256;
	float acc = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
		const f4v v = NT ? __builtin_nontemporal_load(in + i) : in[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if (acc == 12345.678f) out[0] = acc;
}
int main()
{
	const size_t maxb = (size_t)2 << 30;
	f4v *a, *b;
	CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb));
	CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	printf("%10s %14s %14s %14s %14s\n", "S (MB)", "copy TB/s", "copy nt TB/s", "read TB/s", "read nt TB/s");
	for (size_t mb : {8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
		const size_t bytes = mb << 20, n = bytes / 16;
		const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
		const int reps = (int)(((size_t)16 << 30) / bytes) < 20 ? 20 : (int)(((size_t)16 << 30) / bytes);
		double res[4];
		for (int mode = 0; mode < 4; ++mode) {
			auto go = [&](int i) {
				const f4v *src = (i & 1) ? b : a;
				f4v *dst = (i & 1) ? a : b;
				if (mode == 0) hipLaunchKernelGGL((copyk<0>), dim3(grid), dim3(256), 0, 0, src, dst, n);
				else if (mode == 1) hipLaunchKernelGGL((copyk<1>), dim3(grid), dim3(256), 0, 0, src, dst, n);
				else if (mode == 2) hipLaunchKernelGGL((readk<0>), dim3(grid), dim3(256), 0, 0, a, (float *)b, n);
				else hipLaunchKernelGGL((readk<1>), dim3(grid), dim3(256), 0, 0, a, (float *)b, n);
			};
			for (int i = 0; i < 4; ++i) go(i);
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0));
			for (int i = 0; i < reps; ++i) go(i);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			res[mode] = (mode < 2 ? 2.0 : 1.0) * bytes / 1e12 / (ms / reps * 1e-3);
		}
		printf("%10zu %14.2f %14.2f %14.2f %14.2f\n", mb, res[0], res[1], res[2], res[3]);
	}
	// round 4: is the guide's 6.29 TB/s float4 copy reachable at the working set of this pipeline's volumes (2 x 1.5 ... 2 GB,
	// touched once per kernel)?  Grid sizes from one block per CU to one thread per element, both cache policies, at S = 1.5 GB and 2 GB.
	printf("\n%10s %10s %14s %14s\n", "S (MB)", "blocks", "copy TB/s", "copy nt TB/s");
	for (size_t mb : {1536, 2048}) {
		const size_t bytes = mb << 20, n = bytes / 16;
		for (long blocks : {256L, 512L, 1024L, 2048L, 4096L, 8192L, 16384L, 65536L, (long)((n + 255) / 256)}) {
			double res[2];
			for (int mode = 0; mode < 2; ++mode) {
				auto go = [&](int i) {
					const f4v *src = (i & 1) ? b : a;
					f4v *dst = (i & 1) ? a : b;
					if (mode == 0) hipLaunchKernelGGL((copyk<0>), dim3((unsigned)blocks), dim3(256), 0, 0, src, dst, n);
					else hipLaunchKernelGGL((copyk<1>), dim3((unsigned)blocks), dim3(256), 0, 0, src, dst, n);
				};
				for (int i = 0; i < 2; ++i) go(i);
				CK(hipDeviceSynchronize());
				CK(hipEventRecord(e0));
				for (int i = 0; i < 10; ++i) go(i);
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				res[mode] = 2.0 * bytes / 1e12 / (ms / 10 * 1e-3);
			}
			printf("%10zu %10ld %14.2f %14.2f\n", mb, blocks, res[0], res[1]);
		}
	}
	return 0;
}
