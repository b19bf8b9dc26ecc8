// Feature front end of the matching-cost network (gfx950): 3x3 convolution, stride 1, zero padding 1, + bias (+ ReLU), fp32.
//
// Replaces the cudnn.SpatialConvolution(n_in, fm, 3, 3, 1, 1, 1, 1) [+ cudnn.ReLU] layers of net_te
// (main.lua:681-686 arch slow: ReLU after every layer; main.lua:727-746 arch fast: ReLU between the layers, padding set
// to 1 for testing) on the matrix cores: an implicit GEMM  out[co, p] = b[co] + sum_{ky,kx,ci} W[co,ci,ky,kx] * in[ci, p + (ky-1, kx-1)]
// evaluated with v_mfma_f32_32x32x2_f32 -- A = 32 output channels x 2 input channels of one filter tap, B = 2 input
// channels x 32 pixels of one image row, so that a lane ends up with 16 output channels of ONE pixel column and the 32
// lanes of a half-wave store 32 consecutive pixels of a channel (128-byte lines).
//
// A block of 4 waves owns a tile of 4 output rows x 32 columns of one image and ALL output channels (up to 128):
//   * input channels are processed in chunks of 16: the 6 x 34 halo tile of the chunk is staged in LDS once (zero padding
//     materialised there), the 9 filter taps of the chunk one after the other as a [16 ci][Cout] slab, from weights
//     re-laid as [ky][kx][ci][co] by conv_prep_kernel so that a slab is one contiguous run;
//   * the next slab is fetched into registers while the current one is multiplied (16 K-steps x Cout/32 MFMAs per wave).
// The feature maps feeding the pipeline come from cuDNN in the reference, whose algorithm choice is not deterministic
// (cudnn.benchmark = true, main.lua:330): parity is by tolerance against a plain fp32 convolution, not bit-exact.
#include "mc_common.h"

namespace mc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int CV_TH = 4;       // output rows per block (one per wave)
constexpr int CV_TW = 32;      // output columns per block
constexpr int CV_CK = 16;      // input channels per chunk
constexpr int CV_LW = 36;      // LDS row pitch of the input tile (34 used)

// W (Cout, Cin, 3, 3) -> Wt [ky][kx][ci_pad][co_pad], zero padded
__global__ void __launch_bounds__(256) conv_prep_kernel(const float *__restrict__ w, float *__restrict__ wt, int Cin, int Cout, int cip, int cop)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t total = (int64_t)9 * cip * cop;
	if (id >= total) return;
	const int co = (int)(id % cop);
	const int ci = (int)((id / cop) % cip);
	const int tap = (int)(id / ((int64_t)cop * cip));
	wt[id] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.0f;
}

template <int NT, bool RELU>
__global__ void __launch_bounds__(256) conv3x3_kernel(const float *__restrict__ in, const float *__restrict__ wt, const float *__restrict__ bias,
                                                      float *__restrict__ out, int Cin, int Cout, int cip, int H, int W)
{
	constexpr int COP = NT * 32;
	__shared__ float Xs[CV_CK][CV_TH + 2][CV_LW];
	__shared__ float Ws[2][CV_CK][COP];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int x0 = blockIdx.x * CV_TW, y0 = blockIdx.y * CV_TH, n = blockIdx.z;
	const int nl = lane & 31, kh = lane >> 5;
	const int64_t HW = (int64_t)H * W;
	const float *__restrict__ inn = in + (int64_t)n * Cin * HW;

	floatx16 acc[NT];
#pragma unroll
	for (int t = 0; t < NT; ++t) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int co = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
			acc[t][i] = co < Cout ? bias[co] : 0.0f;   // the bias is the start value of the sum
		}
	}

	constexpr int WSLAB = CV_CK * COP;                 // floats per weight slab
	constexpr int WPT = (WSLAB + 255) / 256;           // slab floats per thread
	float wreg[WPT];
	auto fetch_w = [&](int chunk, int tap) {           // slab (tap, chunk) -> registers
		const float *src = wt + ((int64_t)tap * cip + chunk * CV_CK) * COP;
#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			const int e = tid + 256 * k;
			wreg[k] = e < WSLAB ? src[e] : 0.0f;
		}
	};
	auto store_w = [&](int buf) {
#pragma unroll
		for (int k = 0; k < WPT; ++k) {
			const int e = tid + 256 * k;
			if (e < WSLAB) (&Ws[buf][0][0])[e] = wreg[k];
		}
	};

	const int nchunks = cip / CV_CK;
	fetch_w(0, 0);
	for (int ch = 0; ch < nchunks; ++ch) {
		__syncthreads();   // the previous chunk's input tile and weight slabs are no longer read
		// input tile of this chunk: channels ch*16 .. +15, rows y0-1 .. y0+4, columns x0-1 .. x0+32 (zero outside the image)
		for (int e = tid; e < CV_CK * (CV_TH + 2) * (CV_TW + 2); e += 256) {
			const int xx = e % (CV_TW + 2);
			const int yy = (e / (CV_TW + 2)) % (CV_TH + 2);
			const int c = e / ((CV_TW + 2) * (CV_TH + 2));
			const int gx = x0 - 1 + xx, gy = y0 - 1 + yy, gc = ch * CV_CK + c;
			Xs[c][yy][xx] = (gc < Cin && gx >= 0 && gx < W && gy >= 0 && gy < H) ? inn[(int64_t)gc * HW + (int64_t)gy * W + gx] : 0.0f;
		}
		store_w(0);
		__syncthreads();
#pragma unroll 1
		for (int tap = 0; tap < 9; ++tap) {
			const int buf = tap & 1;
			// fetch the next slab (next tap of this chunk, or tap 0 of the next chunk) while this one is multiplied
			const bool more = tap < 8 || ch + 1 < nchunks;
			if (more) fetch_w(tap < 8 ? ch : ch + 1, tap < 8 ? tap + 1 : 0);
			const int ky = tap / 3, kx = tap % 3;
#pragma unroll
			for (int kk = 0; kk < CV_CK / 2; ++kk) {
				const float b = Xs[2 * kk + kh][wv + ky][nl + kx];          // B: 2 input channels x 32 pixels
#pragma unroll
				for (int t = 0; t < NT; ++t) {
					const float a = Ws[buf][2 * kk + kh][t * 32 + nl];       // A: 32 output channels x 2 input channels
					acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
				}
			}
			if (tap < 8) {
				store_w(buf ^ 1);      // the other buffer was last read one tap ago: every wave is past it after this barrier
				__syncthreads();
			}
		}
	}
	// a lane holds pixel x0 + nl of row y0 + wv for output channels t*32 + (i&3) + 8*(i>>2) + 4*kh
	const int oy = y0 + wv, ox = x0 + nl;
	if (oy < H && ox < W) {
		float *__restrict__ o = out + (int64_t)n * Cout * HW + (int64_t)oy * W + ox;
#pragma unroll
		for (int t = 0; t < NT; ++t) {
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int co = t * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
				if (co < Cout) {
					const float v = acc[t][i];
					o[(int64_t)co * HW] = RELU ? fmaxf(v, 0.0f) : v;
				}
			}
		}
	}
}

size_t conv3x3_workspace_bytes(int Cin, int Cout)
{
	const int cip = (Cin + CV_CK - 1) / CV_CK * CV_CK, cop = (Cout + 31) / 32 * 32;
	return ((size_t)9 * cip * cop * sizeof(float) + 255) & ~(size_t)255;
}

int conv3x3(const float *in, const float *w, const float *bias, float *out, int N, int Cin, int Cout, int H, int W, int relu,
            void *workspace, hipStream_t st)
{
	const int cip = (Cin + CV_CK - 1) / CV_CK * CV_CK, cop = (Cout + 31) / 32 * 32;
	float *wt = (float *)workspace;
	const int64_t total = (int64_t)9 * cip * cop;
	hipLaunchKernelGGL(conv_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, wt, Cin, Cout, cip, cop);
	const dim3 grid(cdiv(W, CV_TW), cdiv(H, CV_TH), N), block(256);
#define MC_CONV_GO(NT_)                                                                                                      \
	do {                                                                                                                     \
		if (relu) hipLaunchKernelGGL((conv3x3_kernel<NT_, true>), grid, block, 0, st, in, wt, bias, out, Cin, Cout, cip, H, W);  \
		else hipLaunchKernelGGL((conv3x3_kernel<NT_, false>), grid, block, 0, st, in, wt, bias, out, Cin, Cout, cip, H, W);      \
	} while (0)
	if (cop == 32) MC_CONV_GO(1);
	else if (cop == 64) MC_CONV_GO(2);
	else if (cop == 96) MC_CONV_GO(3);
	else MC_CONV_GO(4);
#undef MC_CONV_GO
	return check_launch("conv3x3");
}

}  // namespace mc
