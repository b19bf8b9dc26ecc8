// Microbenchmark (tuning aid, not product): what each ingredient of cbca_lean2_kernel's access pattern costs against the one-element-per-
// thread copy (6.3 / 6.7 TB/s at 2 x 2 GB, bw_sizes.hip).  Volume D x H x W floats (256 x 1000 x 1500); a wave = R rows x 256 columns
// (16 bytes per lane and row), grid (blocks of one plane, planes), strips fastest.
//   MODE 0  copy: R loads, R stores                      MODE 1  + the two halo rows (R + 2 loads)
//   MODE 2  + the strip's outer columns (one dword per row in lanes 0 / 63)   MODE 3  + the 3 x 3 sums (36 adds, 4 divides per row)
//   hipcc --offload-arch=gfx950 -O3 bw_lean.hip -o bw_lean.bin && ./bw_lean.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int R, int MODE, int NT, int BAND>
__global__ void __launch_bounds__(256) k(const float *__restrict__ in, float *__restrict__ out, int H, int W, int gx, int gy, int gyb)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int d = blockIdx.y;
	const unsigned lw = (BAND ? (blockIdx.x >> 3) : blockIdx.x) * 4u + wv;
	const int strip = lw % gx;
	int chunk = lw / gx;
	if (BAND) { if (chunk >= gyb) return; chunk += (blockIdx.x & 7) * gyb; }
	if (chunk >= gy) return;
	const int y0 = chunk * R, xs = strip * 256 + 4 * lane;
	const int HW = H * W;
	const unsigned OOB = 0x80000000u;
	const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)d * HW), 0, HW * 4, 0x00020000);
	const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)d * HW), 0, HW * 4, 0x00020000);
	constexpr int NR = MODE >= 1 ? R + 2 : R;
	const int r0 = MODE >= 1 ? y0 - 1 : y0;
	const bool lane_in = xs < W;
	const int ecol = lane == 0 ? xs - 1 : (lane == 63 ? xs + 4 : -1);
	const bool eok = ecol >= 0 && ecol < W;
	u4 v[NR];
	unsigned e[NR];
#pragma unroll
	for (int q = 0; q < NR; ++q) {
		const int r = r0 + q;
		const bool rok = (unsigned)r < (unsigned)H;
		v[q] = __builtin_amdgcn_raw_buffer_load_b128(rv, (rok & lane_in) ? (unsigned)(r * W + xs) * 4u : OOB, 0, NT ? 2 : 0);
		e[q] = MODE >= 2 ? __builtin_amdgcn_raw_buffer_load_b32(rv, (rok & eok) ? (unsigned)(r * W + ecol) * 4u : OOB, 0, 0) : 0u;
	}
#pragma unroll
	for (int q = 0; q < R; ++q) {
		const int yo = y0 + q;
		u4 o = v[MODE >= 1 ? q + 1 : q];
		if (MODE >= 2) o.x ^= e[q] & 1u;
		if (MODE >= 3) {
			float res[4];
			const float *a = (const float *)&v[q], *b = (const float *)&v[q + 1], *c = (const float *)&v[q + 2];
			float ra[6], rb[6], rc[6];
			for (int t = 0; t < 4; ++t) { ra[t + 1] = a[t]; rb[t + 1] = b[t]; rc[t + 1] = c[t]; }
			ra[0] = __shfl_up(ra[4], 1); rb[0] = __shfl_up(rb[4], 1); rc[0] = __shfl_up(rc[4], 1);
			ra[5] = __shfl_down(ra[1], 1); rb[5] = __shfl_down(rb[1], 1); rc[5] = __shfl_down(rc[1], 1);
			for (int j = 0; j < 4; ++j) {
				float s = 0;
				s += ra[j]; s += ra[j + 1]; s += ra[j + 2]; s += rb[j]; s += rb[j + 1]; s += rb[j + 2]; s += rc[j]; s += rc[j + 1]; s += rc[j + 2];
				res[j] = s / 9.0f;
			}
			o = u4{__float_as_uint(res[0]), __float_as_uint(res[1]), __float_as_uint(res[2]), __float_as_uint(res[3])};
		}
		__builtin_amdgcn_raw_buffer_store_b128(o, ro, (yo < H && lane_in) ? (unsigned)(yo * W + xs) * 4u : OOB, 0, NT ? 2 : 0);
	}
}

template <int R, int MODE, int NT, int BAND>
void run(const float *a, float *b, int D, int H, int W, const char *tag)
{
	const int gx = (W + 255) / 256, gy = (H + R - 1) / R, gyb = (gy + 7) / 8;
	const dim3 grid(BAND ? 8 * ((gx * gyb + 3) / 4) : (gx * gy + 3) / 4, D);
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<R, MODE, NT, BAND>), grid, dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? (float *)a : b, H, W, gx, gy, gyb);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<R, MODE, NT, BAND>), grid, dim3(256), 0, 0, (i & 1) ? b : a, (i & 1) ? (float *)a : b, H, W, gx, gy, gyb);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms; CK(hipEventElapsedTime(&ms, e0, e1));
	printf("R=%d mode %d %-28s %s %s: %.3f ms, %.2f TB/s of 2 V\n", R, MODE, tag, NT ? "nt   " : "plain", BAND ? "band/XCD" : "linear  ", ms / 10,
	       2.0 * D * H * W * 4 / 1e12 / (ms / 10 * 1e-3));
}

int main()
{
	const int D = 256, H = 1000, W = 1500;
	const size_t bytes = (size_t)D * H * W * 4;
	float *a, *b;
	CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
	CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
#define ALL(R, MODE, TAG) run<R, MODE, 1, 0>(a, b, D, H, W, TAG); run<R, MODE, 1, 1>(a, b, D, H, W, TAG); run<R, MODE, 0, 0>(a, b, D, H, W, TAG); run<R, MODE, 0, 1>(a, b, D, H, W, TAG);
	if (getenv("BW_LEAN_FULL")) {
	ALL(8, 0, "copy")
	ALL(8, 1, "+ halo rows")
	ALL(8, 2, "+ outer columns")
	ALL(8, 3, "+ 3x3 sums")
	ALL(4, 0, "copy")
	ALL(16, 0, "copy")
	ALL(16, 3, "+ 3x3 sums")
	ALL(1, 0, "copy")
	ALL(2, 0, "copy")
	}
	ALL(1, 3, "+ 3x3 sums")
	ALL(2, 3, "+ 3x3 sums")
	ALL(3, 3, "+ 3x3 sums")
	ALL(4, 3, "+ 3x3 sums")
	ALL(6, 3, "+ 3x3 sums")
	ALL(1, 1, "+ halo rows")
	ALL(2, 1, "+ halo rows")
	return 0;
}
