// Microbenchmark (tuning aid, not product): in which ORDER do the fp32 matrix-core instructions of gfx950 add the K products of one
// instruction to the accumulator?  stereo_join.hip relies on v_mfma_f32_32x32x2_f32 being fma(a1, b1, fma(a0, b0, c)) (k ascending, one
// rounding per product-add).  A lead for cross-based aggregation: with A = 1.0 a matrix-core instruction is a
// serial ADDER -- D[i][j] = ((C[i][j] + B[0][j]) + B[1][j]) + ... -- which would add one value to 16 accumulators of 16 columns in one
// instruction instead of cmpx + four v_add_f32 per tap.  That is only exact if v_mfma_f32_16x16x4_f32 adds its four products one after
// the other, k ascending.  This program decides it: random operands of mixed magnitude, the device result against (a) the sequential fmaf
// chain k = 0 .. K-1, (b) the reverse chain, (c) products summed pairwise first.
//   hipcc --offload-arch=gfx950 -O3 mfma_order.hip -o mfma_order.bin && ./mfma_order.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// 16x16x4: A[i][k]: lane = 16 k + i; B[k][j]: lane = 16 k + j; C/D[i][j]: lane = 16 (i / 4) + j, register i % 4
__global__ void k16(const float *A, const float *B, const float *C, float *D)
{
	const int lane = threadIdx.x;
	const float a = A[(lane & 15) * 4 + (lane >> 4)], b = B[(lane >> 4) * 16 + (lane & 15)];
	f4 c;
	for (int r = 0; r < 4; ++r) c[r] = C[(4 * (lane >> 4) + r) * 16 + (lane & 15)];
	c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
	for (int r = 0; r < 4; ++r) D[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}
// 32x32x2: A[i][k]: lane = 32 k + i; B[k][j]: lane = 32 k + j; C/D[i][j]: lane = 32 ((i / 4) % 2) + j, register (i % 4) + 4 (i / 8)
__global__ void k32(const float *A, const float *B, const float *C, float *D)
{
	const int lane = threadIdx.x;
	const float a = A[(lane & 31) * 2 + (lane >> 5)], b = B[(lane >> 5) * 32 + (lane & 31)];
	f16v c;
	for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)];
	c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
	for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}

static float rnd(unsigned &s)
{
	s = s * 1664525u + 1013904223u;
	const float m = (float)((s >> 8) & 0xffff) / 65536.0f + 0.5f;
	const int e = (int)((s >> 24) % 25) - 12;      // magnitudes 2^-12 .. 2^12: the order of the additions shows
	return ((s >> 7) & 1 ? -m : m) * ldexpf(1.0f, e);
}

template <int M, int K, typename KERN>
static void run(const char *name, KERN kern, bool ones)
{
	float hA[M * K], hB[K * M], hC[M * M], hD[M * M];
	float *dA, *dB, *dC, *dD;
	CK(hipMalloc(&dA, sizeof hA)); CK(hipMalloc(&dB, sizeof hB)); CK(hipMalloc(&dC, sizeof hC)); CK(hipMalloc(&dD, sizeof hD));
	long seq = 0, rev = 0, pair = 0, total = 0;
	unsigned s = 12345u;
	for (int trial = 0; trial < 200; ++trial) {
		for (float &v : hA) v = ones ? 1.0f : rnd(s);
		for (float &v : hB) v = rnd(s);
		for (float &v : hC) v = rnd(s);
		CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
		CK(hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
		CK(hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost));
		for (int i = 0; i < M; ++i)
			for (int j = 0; j < M; ++j) {
				float a = hC[i * M + j], r = hC[i * M + j];
				for (int k = 0; k < K; ++k) a = fmaf(hA[i * K + k], hB[k * M + j], a);
				for (int k = K - 1; k >= 0; --k) r = fmaf(hA[i * K + k], hB[k * M + j], r);
				double p = 0;   // the products summed exactly first, one rounding at the end
				for (int k = 0; k < K; ++k) p += (double)hA[i * K + k] * (double)hB[k * M + j];
				const float q = (float)(p + (double)hC[i * M + j]);
				const float d = hD[i * M + j];
				seq += !memcmp(&d, &a, 4); rev += !memcmp(&d, &r, 4); pair += !memcmp(&d, &q, 4); ++total;
			}
	}
	printf("%-34s A %-6s: == k-ascending fmaf chain %ld / %ld, == k-descending %ld, == exact sum rounded once %ld  -> %s\n", name, ones ? "= 1.0" : "random",
	       seq, total, rev, pair, seq == total ? "SEQUENTIAL, k ascending" : "NOT the sequential chain");
}

int main()
{
	run<16, 4>("v_mfma_f32_16x16x4_f32", k16, false);
	run<16, 4>("v_mfma_f32_16x16x4_f32", k16, true);
	run<32, 2>("v_mfma_f32_32x32x2_f32", k32, false);
	run<32, 2>("v_mfma_f32_32x32x2_f32", k32, true);
	return 0;
}
