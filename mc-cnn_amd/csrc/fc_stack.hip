// Accurate-architecture matching cost: the per-disparity fully-connected stack of arch `slow`
// (main.lua:958-983; layers nn.SpatialConvolution1_fw = 1x1 convolutions done as addmm + bias,
// SpatialConvolution1_fw.lua:11-31; net_te2 of main.lua:688-695: 2*fm -> nh2, (l2-1) x nh2 -> nh2 with ReLU,
// nh2 -> 1, Sigmoid).  For every pixel x and disparity d with x-d >= 0 the reference feeds
// concat(L[:,y,x], R[:,y,x-d]) through the stack and stores the result at volL[d,y,x] / (second direction)
// volR[d,y,x-d]; here each value is computed once and stored to both volumes.
//
// This is the dense GEMM chain of the path (1.06 MFLOP per voxel, 99 TFLOP per KITTI-228 pair) and runs on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, the reference's precision):
//   * layer 1 is linear in the concatenation: W1 [L;R] = W1L L(x) + W1R R(x-d).  fc_project_kernel computes the two
//     per-pixel projections once (2 x nh2 x HW floats), so layer 1 costs one add per voxel instead of a K=2C GEMM.
//   * a block owns 96 consecutive voxels of one (y, d) (three 32-row M tiles) and keeps their activations transposed
//     in LDS, Ht[k][m] with row stride 97 (conflict-free for the MFMA A-operand reads -- lanes along m -- and for the
//     write-back -- lanes along n).  Each of the 4 waves owns nh2/4 output columns (three 32-column N tiles): 9
//     accumulator tiles per wave stay in registers across the K loop, the layer's outputs are written back over the
//     inputs after a barrier (single LDS buffer), the B operand (pre-transposed weights Wt[k][n], coalesced, L2-resident)
//     is shared by the three M tiles.
//   * the last layer (nh2 -> 1) is a dot product per voxel; sigmoid in fp32.
// The reference's own GEMMs are cuBLAS/cudnn calls whose summation order is not pinned, so parity for this operator is
// by tolerance (1e-4 on the sigmoid outputs), not bit-exact; tests/test_gpu_fc.py states the bar.
#include "mc_common.h"

namespace mc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int FC_NH = 384;            // hidden width (nh2 of every accurate preset, main.lua:77,124)
constexpr int FC_MT = 3;              // 32-row M tiles per block
constexpr int FC_M = 32 * FC_MT;      // voxels per block
constexpr int FC_LD = FC_M + 1;       // LDS row stride of Ht[k][m]

// out[p][n] = sum_c Wt[c][n] * X[c][p]   (X: (C,HW) feature map; Wt: (C, NH) transposed weight slice; out: (HW, NH))
__global__ void __launch_bounds__(256) fc_project_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                         float *__restrict__ out, int C, int64_t HW)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int nl = lane & 31, kh = lane >> 5;
	const int64_t p0 = (int64_t)blockIdx.x * 32;
	const int n0 = wave * (FC_NH / 4);
	floatx16 acc[3];
#pragma unroll
	for (int t = 0; t < 3; ++t)
#pragma unroll
		for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
	const int64_t p = p0 + nl;
	for (int kk = 0; kk < (C + 1) / 2; ++kk) {
		const int c = 2 * kk + kh;
		const float a = (c < C && p < HW) ? X[(int64_t)c * HW + p] : 0.0f;
#pragma unroll
		for (int t = 0; t < 3; ++t) {
			const float b = c < C ? Wt[(int64_t)c * FC_NH + n0 + 32 * t + nl] : 0.0f;
			acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
		}
	}
#pragma unroll
	for (int t = 0; t < 3; ++t)
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int m = (i & 3) + 8 * (i >> 2) + 4 * kh;
			if (p0 + m < HW) out[(p0 + m) * FC_NH + n0 + 32 * t + nl] = acc[t][i];
		}
}

struct FcArgs {
	const float *PL, *PR;         // (HW, NH) projections of the left / right features through layer 1
	const float *b1;              // (NH)
	const float *Wt[6];           // hidden layers: transposed weights (NH, NH): Wt[k][n]
	const float *bh[6];           // hidden biases (NH)
	int n_hidden;                 // number of NH x NH layers
	const float *wlast;           // (NH)
	const float *blast;           // (1)
	float *volL, *volR;           // (D,H,W)
	int D, H, W;
};

__global__ void __launch_bounds__(256) fc_stack_kernel(const FcArgs A)
{
	extern __shared__ __attribute__((aligned(16))) float Ht[];  // [FC_NH][FC_LD]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int nl = lane & 31, kh = lane >> 5;
	const int W = A.W, H = A.H;
	// XCD-aware: all blocks of one image row on one XCD (its PL / PR rows, 2 x W x NH floats, stay in that L2)
	const int tiles_x = (W + FC_M - 1) / FC_M;
	const int b = blockIdx.x;
	const int xcd = b & 7, k = b >> 3;
	const int per_row = tiles_x * A.D;
	const int y = (k / per_row) * 8 + xcd;
	if (y >= H) return;
	const int rem = k % per_row;
	const int d = rem / tiles_x;
	const int x0 = d + (rem - d * tiles_x) * FC_M;   // voxels x0 .. x0+95 of (y, d); valid while x < W
	if (x0 >= W) return;
	const int nvox = min(FC_M, W - x0);
	const int64_t HW = (int64_t)H * W;
	const int64_t rowpix = (int64_t)y * W;

	// ---- layer 1: relu((PL[x] + PR[x-d]) + b1), stored transposed ----
	// (8 independent load pairs in flight per thread: the rows are L2-resident, the loop is latency-bound otherwise)
	constexpr int FILL_U = 8;
	static_assert((FC_M * FC_NH) % (256 * FILL_U) == 0, "fill loop shape");
	for (int base = threadIdx.x; base < FC_M * FC_NH; base += 256 * FILL_U) {
		float pl[FILL_U], pr[FILL_U];
#pragma unroll
		for (int u = 0; u < FILL_U; ++u) {
			const int idx = base + 256 * u;
			const int m = idx / FC_NH, n = idx - m * FC_NH;
			const int64_t px = rowpix + x0 + (m < nvox ? m : 0);
			pl[u] = A.PL[px * FC_NH + n];
			pr[u] = A.PR[(px - d) * FC_NH + n];
		}
#pragma unroll
		for (int u = 0; u < FILL_U; ++u) {
			const int idx = base + 256 * u;
			const int m = idx / FC_NH, n = idx - m * FC_NH;
			const float v = fmaxf((pl[u] + pr[u]) + A.b1[n], 0.0f);
			Ht[n * FC_LD + m] = m < nvox ? v : 0.0f;
		}
	}
	__syncthreads();

	// ---- hidden layers: Ht <- relu(W Ht + b), in place ----
	const int n0 = wave * (FC_NH / 4);
	for (int l = 0; l < A.n_hidden; ++l) {
		const float *__restrict__ Wt = A.Wt[l];
		floatx16 acc[FC_MT][3];
#pragma unroll
		for (int mt = 0; mt < FC_MT; ++mt)
#pragma unroll
			for (int t = 0; t < 3; ++t)
#pragma unroll
				for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.0f;
		// One wave per SIMD: nothing but prefetch covers latency.  The weight operand comes from L2 (a 590 KB matrix per
		// layer, ~1 us away): it is fetched FC_BQ k-steps ahead through a register ring (every slot has one load
		// statement in the unrolled body, so hipcc can count the loads still in flight); the activation operand comes from
		// LDS one k-step ahead.
		constexpr int FC_BQ = 4;
		static_assert((FC_NH / 2) % FC_BQ == 0, "k loop shape");
		float bq[FC_BQ][3], an[FC_MT];
		auto load_b = [&](float (&b)[3], int kq) {
			const int kc = kq < FC_NH / 2 ? kq : FC_NH / 2 - 1;
#pragma unroll
			for (int t = 0; t < 3; ++t) b[t] = Wt[(int64_t)(2 * kc + kh) * FC_NH + n0 + 32 * t + nl];
		};
#pragma unroll
		for (int u = 0; u < FC_BQ; ++u) load_b(bq[u], u);
#pragma unroll
		for (int mt = 0; mt < FC_MT; ++mt) an[mt] = Ht[kh * FC_LD + 32 * mt + nl];
		for (int kk0 = 0; kk0 < FC_NH / 2; kk0 += FC_BQ) {
#pragma unroll
			for (int u = 0; u < FC_BQ; ++u) {
				const int kk = kk0 + u;
				float a[FC_MT];
#pragma unroll
				for (int mt = 0; mt < FC_MT; ++mt) a[mt] = an[mt];
				const int kn = kk + 1 < FC_NH / 2 ? kk + 1 : kk;
#pragma unroll
				for (int mt = 0; mt < FC_MT; ++mt) an[mt] = Ht[(2 * kn + kh) * FC_LD + 32 * mt + nl];
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int mt = 0; mt < FC_MT; ++mt)
#pragma unroll
					for (int t = 0; t < 3; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bq[u][t], acc[mt][t], 0, 0, 0);
				// the slot is consumed (its MFMAs are issued) and immediately refilled FC_BQ steps ahead: one definition per
				// slot and no value live across it, so no copies of in-flight loads at the loop edge; the barriers keep the
				// refill here instead of being sunk to just before its use
				__builtin_amdgcn_sched_barrier(0);
				load_b(bq[u], kk + FC_BQ);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		__syncthreads();  // every wave has read the whole of Ht
		const float *__restrict__ bias = A.bh[l];
#pragma unroll
		for (int t = 0; t < 3; ++t) {
			const int n = n0 + 32 * t + nl;
			const float bv = bias[n];
#pragma unroll
			for (int mt = 0; mt < FC_MT; ++mt)
#pragma unroll
				for (int i = 0; i < 16; ++i) {
					const int m = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * kh;
					Ht[n * FC_LD + m] = fmaxf(acc[mt][t][i] + bv, 0.0f);
				}
		}
		__syncthreads();
	}

	// ---- output layer: sigmoid(w . h + b) ----
	if (threadIdx.x < FC_M) {
		const int m = threadIdx.x;
		if (m < nvox) {
			float s = 0.0f;
			for (int kx = 0; kx < FC_NH; kx += 8) {  // 8 LDS reads in flight, summed in k order
				float h[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) h[u] = Ht[(kx + u) * FC_LD + m];
#pragma unroll
				for (int u = 0; u < 8; ++u) s = fmaf(A.wlast[kx + u], h[u], s);
			}
			s += A.blast[0];
			const float r = 1.0f / (1.0f + expf(-s));
			const int x = x0 + m;
			A.volL[(int64_t)d * HW + rowpix + x] = r;
			A.volR[(int64_t)d * HW + rowpix + x - d] = r;
		}
	}
}

// transposes a (rows, cols) row-major matrix slice: out[c][r] = in[r][col0 + c], c < ncols
__global__ void __launch_bounds__(256) fc_transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int ld,
                                                           int col0, int ncols)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= (int64_t)rows * ncols) return;
	const int c = (int)(id / rows), r = (int)(id - (int64_t)c * rows);
	out[id] = in[(int64_t)r * ld + col0 + c];
}

size_t fc_workspace_bytes(int C, int n_hidden, int H, int W)
{
	const size_t HW = (size_t)H * W;
	size_t b = 2 * HW * FC_NH * sizeof(float);                         // PL, PR
	b += ((size_t)2 * C * FC_NH + (size_t)n_hidden * FC_NH * FC_NH) * sizeof(float);  // transposed weights
	return (b + 255) & ~(size_t)255;
}

// weights[l]: (out, in) row-major as nn.SpatialConvolution1_fw stores them; layer 0: (NH, 2C); 1..n_hidden: (NH, NH);
// last: (1, NH).
int fc_stack(const float *featL, const float *featR, int C, int H, int W, int D, const float *const *weights,
             const float *const *biases, int n_layers, float *volL, float *volR, void *workspace, hipStream_t st)
{
	const int n_hidden = n_layers - 2;
	const int64_t HW = (int64_t)H * W;
	float *PL = (float *)workspace, *PR = PL + HW * FC_NH;
	float *W1Lt = PR + HW * FC_NH, *W1Rt = W1Lt + (size_t)C * FC_NH, *Wh = W1Rt + (size_t)C * FC_NH;
	hipLaunchKernelGGL(fc_transpose_kernel, dim3(cdiv((int64_t)FC_NH * C, 256)), dim3(256), 0, st, weights[0], W1Lt, FC_NH, 2 * C, 0, C);
	hipLaunchKernelGGL(fc_transpose_kernel, dim3(cdiv((int64_t)FC_NH * C, 256)), dim3(256), 0, st, weights[0], W1Rt, FC_NH, 2 * C, C, C);
	FcArgs A;
	for (int l = 0; l < n_hidden; ++l) {
		float *wt = Wh + (size_t)l * FC_NH * FC_NH;
		hipLaunchKernelGGL(fc_transpose_kernel, dim3(cdiv((int64_t)FC_NH * FC_NH, 256)), dim3(256), 0, st, weights[1 + l], wt, FC_NH, FC_NH,
		                   0, FC_NH);
		A.Wt[l] = wt;
		A.bh[l] = biases[1 + l];
	}
	hipLaunchKernelGGL(fc_project_kernel, dim3(cdiv(HW, 32)), dim3(256), 0, st, featL, W1Lt, PL, C, HW);
	hipLaunchKernelGGL(fc_project_kernel, dim3(cdiv(HW, 32)), dim3(256), 0, st, featR, W1Rt, PR, C, HW);
	A.PL = PL; A.PR = PR; A.b1 = biases[0];
	A.n_hidden = n_hidden;
	A.wlast = weights[n_layers - 1];
	A.blast = biases[n_layers - 1];
	A.volL = volL; A.volR = volR;
	A.D = D; A.H = H; A.W = W;
	const size_t lds = (size_t)FC_NH * FC_LD * sizeof(float);
	static bool raised = false;
	if (!raised) {
		const hipError_t e = hipFuncSetAttribute((const void *)fc_stack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) {
			set_error("fc_stack: hipFuncSetAttribute: %s", hipGetErrorString(e));
			return (int)e;
		}
		raised = true;
	}
	const int tiles_x = (W + FC_M - 1) / FC_M;
	const int64_t blocks = (int64_t)((H + 7) / 8) * tiles_x * D * 8;
	hipLaunchKernelGGL(fc_stack_kernel, dim3((unsigned)blocks), dim3(256), lds, st, A);
	return check_launch("fc_stack");
}

}  // namespace mc
