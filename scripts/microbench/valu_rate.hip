// Microbenchmark (tuning aid): issue rates of the instructions the cross-based aggregation walk is made of -- v_add_f32,
// v_pk_add_f32 (op_sel broadcast), v_cndmask_b32, v_cmp, ds_read_b32 -- per SIMD and per CU, at 1 .. 8 waves per SIMD.
// The question it answers: is a wave64 VALU instruction 2 or 4 cycles on gfx950, and does v_pk_add_f32 run at full rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;

// MODE 0: 8 independent v_add_f32 chains; 1: 4 independent v_pk_add_f32 chains (8 adds) with a broadcast operand;
// 2: 8 x (v_cndmask + v_add); 3: 8 x ds_read_b32 + 8 v_add; 4: v_cmp + 8 v_cndmask;  5: one serial v_add chain
template <int MODE> __global__ void __launch_bounds__(256) rate(float *out, int iters)
{
	__shared__ float lds[4096];
	for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i * 1e-9f;
	__syncthreads();
	float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	float v = 1e-9f * threadIdx.x;
	const float *p = lds + (threadIdx.x & 63);
	int n = threadIdx.x & 7;
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0) {
			asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
			             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(v));
		} else if (MODE == 1) {
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2 s0 = {a0, a1}, s1 = {a2, a3}, s2 = {a4, a5}, s3 = {a6, a7}, vv = {v, v};
			asm volatile("v_pk_add_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, %4 op_sel_hi:[1,0]\n"
			             "v_pk_add_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
			             : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : "v"(vv));
			a0 = s0.x; a1 = s0.y; a2 = s1.x; a3 = s1.y; a4 = s2.x; a5 = s2.y; a6 = s3.x; a7 = s3.y;
		} else if (MODE == 2) {
			float t;
			asm volatile("v_cmp_gt_i32 vcc, %9, %10\n"
			             "v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %0, %0, %8\n v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %1, %1, %8\n"
			             "v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %2, %2, %8\n v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %3, %3, %8\n"
			             "v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %4, %4, %8\n v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %5, %5, %8\n"
			             "v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %6, %6, %8\n v_cndmask_b32 %8, %11, %12, vcc\n v_add_f32 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t)
			             : "v"(n), "v"(it), "v"(-0.0f), "v"(v) : "vcc");
		} else if (MODE == 3) {
			float t0, t1, t2, t3, t4, t5, t6, t7;
			asm volatile("ds_read_b32 %8, %16\n ds_read_b32 %9, %16 offset:4\n ds_read_b32 %10, %16 offset:8\n ds_read_b32 %11, %16 offset:12\n"
			             "ds_read_b32 %12, %16 offset:16\n ds_read_b32 %13, %16 offset:20\n ds_read_b32 %14, %16 offset:24\n ds_read_b32 %15, %16 offset:28\n"
			             "s_waitcnt lgkmcnt(0)\n"
			             "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %11\n"
			             "v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %13\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %15\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2),
			               "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
			             : "v"((unsigned)(size_t)p));
		} else if (MODE == 5) {
			asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
			             "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
			             : "+v"(a0) : "v"(v));
		} else if (MODE == 6) {   // v_cmpx-predicated: exec narrowed per "tap", 4 adds under it, exec restored
			asm volatile("s_mov_b64 s[20:21], exec\n"
			             "v_cmpx_gt_i32 %4, %5\n v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n"
			             "s_mov_b64 exec, s[20:21]\n"
			             "v_cmpx_gt_i32 %4, %5\n v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %6\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %6\n"
			             "s_mov_b64 exec, s[20:21]\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(n + 100000), "v"(it), "v"(v) : "s20", "s21", "vcc");
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *name, double valu_per_iter, double adds_per_iter, float *out)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
		const int blocks = 256 * wps;         // 256 threads = 4 waves = one per SIMD; wps blocks per CU
		hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 64);
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, ITERS);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		const double waves = (double)blocks * 4;
		const double winstr = waves * ITERS * valu_per_iter;   // wave-instructions
		const double per_simd_cycles = ms * 1e-3 * 2.4e9 / (winstr / 1024.0);
		printf("%-28s waves/SIMD %d: %.3f ms, %.1f G wave-VALU/s, %.2f cycles@2.4GHz per wave-instr per SIMD, %.2f T adds/s\n", name, wps, ms,
		       winstr / ms * 1e-6, per_simd_cycles, waves * 64 * ITERS * adds_per_iter / ms * 1e-9);
	}
}

int main()
{
	float *out;
	CK(hipMalloc(&out, 256 * 8 * 256 * 4));
	run<0>("8 indep v_add_f32", 8, 8, out);
	run<5>("8 serial v_add_f32", 8, 8, out);
	run<1>("4 v_pk_add_f32 (bcast)", 4, 8, out);
	run<2>("cmp + 8 (cndmask+add)", 17, 8, out);
	run<3>("8 ds_read_b32 + 8 add", 8, 8, out);
	run<6>("2 x (cmpx + 4 add)", 10, 8, out);
	return 0;
}
