// Cross-based cost aggregation (gfx950): support arms (`cross`) and region mean (`cbca`).
//
// Parity rule: the region sum keeps the reference's order -- rows yy
// ascending, inside a row xx ascending, ONE fp32 accumulator, then an IEEE divide by the
// integer count (adcensus.cu:356-373) -- so that costs, and therefore arg-min disparities,
// are bit-identical.  No separable / prefix-sum shortcut is taken on this path.
#include "mc_common.h"

namespace mc {

// ---- cross, adcensus.cu:280-322 --------------------------------------------------------------
__global__ void __launch_bounds__(256) cross_kernel(const float *__restrict__ img, float *__restrict__ out, int64_t size, int H, int W,
                                                    int L1, float tau1)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= size) return;
	int64_t t = id;
	const int x = (int)(t % W);
	t /= W;
	const int y = (int)(t % H);
	t /= H;
	const int dir = (int)t;
	const int dx = dir == 0 ? -1 : (dir == 1 ? 1 : 0);
	const int dy = dir == 2 ? -1 : (dir == 3 ? 1 : 0);
	const float c = img[y * W + x];
	int xx, yy;
	for (xx = x + dx, yy = y + dy;; xx += dx, yy += dy) {
		if (xx < 0 || xx >= W || yy < 0 || yy >= H) break;
		const int dist = max(abs(xx - x), abs(yy - y));
		if (dist == 1) continue;
		if (fabsf(c - img[yy * W + xx]) >= tau1) break;  // rule 1
		if (dist >= L1) break;                           // rule 2
	}
	out[id] = (float)(dir <= 1 ? xx : yy);
}

int cross(const float *img, float *arms, int H, int W, int L1, float tau1, hipStream_t st)
{
	const int64_t size = (int64_t)4 * H * W;
	hipLaunchKernelGGL(cross_kernel, dim3(cdiv(size, 256)), dim3(256), 0, st, img, arms, size, H, W, L1, tau1);
	return check_launch("cross");
}

// ---- cbca v1, adcensus.cu:343-377: one thread per voxel, (D,H,W), reads through L1/L2 -------------
__global__ void __launch_bounds__(256) cbca_direct_kernel(const float *__restrict__ x0c, const float *__restrict__ x1c,
                                                          const float *__restrict__ vol, float *__restrict__ out, int D, int H, int W,
                                                          int direction)
{
	const int x = blockIdx.x * 64 + (threadIdx.x & 63);
	const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
	const int d = blockIdx.z;
	if (x >= W || y >= H) return;
	const int64_t HW = (int64_t)H * W;
	const int64_t id = (int64_t)d * HW + (int64_t)y * W + x;
	const int xp = x + d * direction;
	if (xp < 0 || xp >= W) {
		out[id] = vol[id];
		return;
	}
	const float *__restrict__ a0 = x0c, *__restrict__ a1 = x1c;
	const float sh = (float)(d * direction);
	const int yy_s = (int)fmaxf(a0[2 * HW + y * W + x], a1[2 * HW + y * W + xp]);
	const int yy_t = (int)fminf(a0[3 * HW + y * W + x], a1[3 * HW + y * W + xp]);
	float sum = 0;
	int cnt = 0;
	const float *__restrict__ vd = vol + (int64_t)d * HW;
	for (int yy = yy_s + 1; yy < yy_t; yy++) {
		const int xx_s = (int)fmaxf(a0[0 * HW + yy * W + x], a1[0 * HW + yy * W + xp] - sh);
		const int xx_t = (int)fminf(a0[1 * HW + yy * W + x], a1[1 * HW + yy * W + xp] - sh);
		for (int xx = xx_s + 1; xx < xx_t; xx++) {
			sum += vd[yy * W + xx];
			cnt++;
		}
	}
	out[id] = sum / (float)cnt;
}

int cbca(const float *x0c, const float *x1c, const float *vin, float *vout, int D, int H, int W, int direction, hipStream_t st)
{
	hipLaunchKernelGGL(cbca_direct_kernel, dim3(cdiv(W, 64), cdiv(H, 4), D), dim3(256), 0, st, x0c, x1c, vin, vout, D, H, W,
	                   direction);
	return check_launch("cbca");
}

}  // namespace mc
