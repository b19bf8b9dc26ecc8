// Feature front end of the matching-cost network (gfx950): 3x3 convolution, stride 1, zero padding 1, + bias (+ ReLU), fp32.
//
// Replaces the cudnn.SpatialConvolution(n_in, fm, 3, 3, 1, 1, 1, 1) [+ cudnn.ReLU] layers of net_te
// (main.lua:681-686 arch slow: ReLU after every layer; main.lua:727-746 arch fast: ReLU between the layers, padding set
// to 1 for testing) on the matrix cores: an implicit GEMM  out[co, p] = b[co] + sum_{ci,ky,kx} W[co,ci,ky,kx] * in[ci, p + (ky-1, kx-1)]
// evaluated with v_mfma_f32_32x32x2_f32 -- A = 32 output channels x 2 input channels of one filter tap, B = 2 input
// channels x 32 pixels of one image row, so that a lane ends up with 16 output channels of ONE pixel column and the 32
// lanes of a half-wave store 32 consecutive pixels of a channel (128-byte lines).
//
// The kernel is bound by the matrix pipe (64 cycles per instruction and SIMD) and everything else is arranged so that
// the pipe never waits:
//   * ONE block of four waves per CU, one wave per SIMD, resident for the whole launch.  The block's LDS holds the filter
//     bank of the layer -- 64 -> 64 channels: 9 * 64 * 64 * 4 B = 144 KB of the CU's 160 KB -- copied once, in the order
//     the matrix instruction wants its A operand: [ci pair][lane][tap][co tile], so that a lane fetches the 9 taps of a
//     pair of input channels with a few wide, conflict-free ds_reads.  No barrier after that copy.  A layer whose bank
//     is larger (arch slow, 112 -> 112: 504 KB) is split by OUTPUT channels into groups of 64 or 32 whose banks fit
//     (112 -> 32: 126 KB); a group is served by its own share of the CUs and reads the whole input (cheap: the input
//     comes out of the L2 / the Infinity Cache, and the kernel is bound by the matrix pipe, not by memory).
//   * a wave owns a run of (row, 32-column strip) units, cut from the N x strips x H sequence in equal parts (one part
//     per wave of the group: no tail of idle CUs beyond one row), and walks it in tiles of R rows.  The B operand of a
//     tile -- R + 2 input rows x 3 column shifts of a channel pair -- comes STRAIGHT from global memory into registers,
//     one dword per lane and load (32 consecutive pixels per half-wave), fetched one channel pair ahead of the
//     multiplications that use it: 18 loads and 9 ds_reads beside 72 matrix instructions (R = 4, 64 output channels).
//     The last pair of a tile fetches the first pair of the NEXT tile, so that the tile's stores and the next tile's set-up
//     run in the shadow of those requests.
//   * zero padding = the hardware's range check: a position outside the image gets an offset beyond the buffer.
// A bank that does not fit even for 32 output channels (Cin > 140) is processed in chunks of input channels; the block then
// reloads the LDS between chunks (two barriers per chunk and tile).
// The feature maps feeding the pipeline come from cuDNN in the reference, whose algorithm choice is not deterministic
// (cudnn.benchmark = true, main.lua:330): parity is by tolerance against a plain fp32 convolution, not bit-exact.
#include "mc_common.h"
#include <type_traits>

namespace mc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int CV_WAVES = 4;                 // waves per block = SIMDs per CU
constexpr int CV_LDS_BYTES = 159 * 1024;    // the LDS of a CU (160 KB) less the block's few static bytes (__syncthreads_or)
constexpr uint32_t CV_OOB = 0xFFFFFFF0u;    // a byte offset beyond every buffer: loads return 0, stores are dropped

// rows per tile: R * NT accumulator tiles of 16 registers each, and R * 9 * NT matrix instructions (4608 cycles) per
// channel pair to cover the latency of the next pair's requests
__host__ __device__ constexpr int cv_rows(int NT) { return NT == 1 ? 8 : 4; }

// A raw buffer over `bytes` bytes at p, both wave-uniform -- and passed through readfirstlane so that the compiler knows it:
// a descriptor it cannot prove uniform gets every access wrapped in a loop over its distinct values.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float *p, uint32_t bytes)
{
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
	return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// ReLU as ONE v_max_f32: through fmaxf the compiler first quiets an operand it cannot prove quiet (a v_max x, x each).  A NaN sum becomes 0, as with fmaxf.
__device__ __forceinline__ float relu1(float v)
{
	float r;
	asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
	return r;
}

// W (Cout, Cin, 3, 3) -> Wt [group][ci pair][lane = kh * 32 + nl][tap][t] = W[co = 32 (group NT + t) + nl][ci = 2 pair + kh][tap],
// zero padded (channels beyond Cout / Cin, and the pair that makes the pair count even)
__global__ void __launch_bounds__(256) conv_prep_kernel(const float *__restrict__ w, float *__restrict__ wt, int Cin, int Cout, int npairs, int NT, int groups)
{
	const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t total = (int64_t)groups * npairs * 64 * 9 * NT;
	if (id >= total) return;
	const int t = (int)(id % NT);
	const int tap = (int)((id / NT) % 9);
	const int lane = (int)((id / (9 * NT)) % 64);
	const int pair = (int)((id / ((int64_t)9 * NT * 64)) % npairs);
	const int g = (int)(id / ((int64_t)9 * NT * 64 * npairs));
	const int co = 32 * (g * NT + t) + (lane & 31), ci = 2 * pair + (lane >> 5);
	wt[id] = (co < Cout && ci < Cin) ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.0f;
}

// npairs and chunk_pairs are even.  units / units_rem: a wave's share of the N x strips x H units (the first units_rem waves
// of a group take one more).  bpg: blocks per output-channel group.
template <int NT, bool RELU, bool CHUNKED>
__global__ void __launch_bounds__(256) conv3x3_kernel(const float *__restrict__ in, const float *__restrict__ wt, const float *__restrict__ bias,
                                                      float *__restrict__ out, int Cin, int Cout, int npairs, int chunk_pairs, int H, int W,
                                                      int units, int units_rem, int bpg)
{
	constexpr int R = cv_rows(NT);
	constexpr int AV = 9 * NT;                       // A operands of a lane per channel pair
	extern __shared__ __attribute__((aligned(16))) float Wl[];   // [chunk_pairs][64][AV]
	const int tid = threadIdx.x, lane = tid & 63;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform, and the compiler has to know: everything derived from it
	                                                             // (the tile's position, the buffer descriptors) then lives in SGPRs
	const int nl = lane & 31, kh = lane >> 5;
	const int S = (W + 31) >> 5;
	const int64_t HW = (int64_t)H * W;
	const uint32_t plane = (uint32_t)HW * 4u;        // bytes of one channel
	const int grp = blockIdx.x / bpg, bl = blockIdx.x - grp * bpg;
	const int co0 = grp * NT * 32;                   // the group's first output channel
	const float *__restrict__ wtg = wt + (int64_t)grp * npairs * 64 * AV;

	auto load_bank = [&](int p0, int np) {           // LDS <- pairs [p0, p0 + np) of the group's re-laid bank, a straight copy,
		const float4 *src = (const float4 *)(wtg + (int64_t)p0 * 64 * AV);   // twelve 16-byte requests per thread in flight
		float4 *dst = (float4 *)Wl;
		const int n4 = np * 64 * AV / 4;
		for (int e0 = 0; e0 < n4; e0 += 12 * 256) {
			float4 v[12];
#pragma unroll
			for (int j = 0; j < 12; ++j) {
				const int e = e0 + j * 256 + tid;
				v[j] = src[min(e, n4 - 1)];
			}
#pragma unroll
			for (int j = 0; j < 12; ++j) {
				const int e = e0 + j * 256 + tid;
				if (e < n4) dst[e] = v[j];
			}
		}
	};
	if (!CHUNKED) {
		load_bank(0, npairs);
		__syncthreads();
	}

	float bv[NT][16];                                // start value of a sum: the bias of the lane's 16 * NT output channels
#pragma unroll
	for (int t = 0; t < NT; ++t) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int co = co0 + t * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
			const float bco = bias[min(co, Cout - 1)];
			bv[t][i] = co < Cout ? bco : 0.0f;
		}
	}

	// this wave's run of (image, strip, row) units
	const int gw = bl * CV_WAVES + wv;
	int cur = gw * units + min(gw, units_rem);
	const int end = cur + units + (gw < units_rem ? 1 : 0);

	// the tile at unit `at`: image, first column, first row, rows.  What is left of the run in this column is cut into its
	// number of tiles EVENLY (29 rows = 4 4 4 4 4 3 3 3, not seven fours and a one): a tile of one row has 18 matrix
	// instructions per channel pair, too few to cover the next pair's requests
	// `want` > 0 (the first one or two tiles of a run): that many rows instead -- the four waves of a block start with tiles of different
	// heights, so that their tile boundaries, i.e. their bursts of R * NT * 16 stores, fall at different quarters of a tile's time for
	// the rest of the launch.  With every wave of the chip in step the bursts (32 MB together) came at once and the next tile's second
	// channel pair waited for them to drain.
	auto place = [&](int at, int &n, int &x0, int &y0, int &rows, int want = 0) {
		const int col = at / H;
		const int y = at - col * H, nn = col / S;
		const int left = min(H - y, end - at), tiles = (left + R - 1) / R;
		const int even = (left + tiles - 1) / max(tiles, 1);
		n = __builtin_amdgcn_readfirstlane(nn), x0 = __builtin_amdgcn_readfirstlane((col - nn * S) * 32);
		y0 = __builtin_amdgcn_readfirstlane(y), rows = __builtin_amdgcn_readfirstlane((want > 0 && left >= 2 * R) ? want : even);
	};
	// heights of a run's first two tiles for the block's wave wv: the boundaries of wave wv lie wv * R / 4 rows ahead of wave 0's
	// (never a tile of one row: 3 quarters ahead = a tile of 2 rows, then one of R - 1)
	auto phase_rows = [&](int tile) -> int {
		const int first = R - wv * (R / 4);
		if (wv == 0 || tile > 1) return 0;
		if (first >= 2) return tile == 0 ? first : 0;
		return tile == 0 ? 2 : R - 1;
	};
	// byte offsets, within a pair of channel planes, of the lane's pixel in the R + 2 input rows x 3 column shifts; a
	// position outside the image gets an offset beyond the buffer, for which the hardware returns 0: the zero padding
	auto offsets = [&](int x0, int y0, uint32_t (&voff)[R + 2][3]) {
#pragma unroll
		for (int r = 0; r < R + 2; ++r) {
			const int yy = y0 - 1 + r;
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				const int xx = x0 + nl + k - 1;
				const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
				voff[r][k] = ok ? (uint32_t)kh * plane + (uint32_t)(yy * W + xx) * 4u : CV_OOB;
			}
		}
	};
	// operands of channel pair p of image n (pl = the pair's index within the LDS-resident part of the bank).  B: channel 2p
	// for lanes 0-31, 2p + 1 for lanes 32-63, through a buffer of exactly the planes that exist (an odd Cin's last pair has
	// one and the upper half reads 0; the pair that pads the count to even has none)
	auto fetch = [&](int n, const uint32_t (&voff)[R + 2][3], int p, int pl, float (&b)[R + 2][3], float (&a)[AV]) {
		const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(in + ((int64_t)n * Cin + min(2 * p, Cin - 1)) * HW, max(0, min(2, Cin - 2 * p)) * plane);
#pragma unroll
		for (int r = 0; r < R + 2; ++r)
#pragma unroll
			for (int k = 0; k < 3; ++k) b[r][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[r][k], 0, 0));
		const float *src = Wl + (pl * 64 + lane) * AV;
		if constexpr (NT == 2) {
#pragma unroll
			for (int j = 0; j < 9; ++j) {
				const float2 v = ((const float2 *)src)[j];
				a[2 * j] = v.x; a[2 * j + 1] = v.y;
			}
		} else {
#pragma unroll
			for (int j = 0; j < AV; ++j) a[j] = src[j];
		}
	};

	floatx16 acc[R][NT];
	float b0[R + 2][3], b1[R + 2][3];
	float a0[AV], a1[AV];
	int rows = 0;                                    // of the tile being computed

	auto row = [&](int r, const float (&b)[R + 2][3], const float (&a)[AV]) {
#pragma unroll
		for (int ky = 0; ky < 3; ++ky)
#pragma unroll
			for (int kx = 0; kx < 3; ++kx)
#pragma unroll
				for (int t = 0; t < NT; ++t)
					acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(ky * 3 + kx) * NT + t], b[r + ky][kx], acc[r][t], 0, 0, 0);
	};
	// One pair: R rows x 9 taps x NT channel tiles of matrix instructions on one register set (b, a), and the requests for
	// the operands of pair pn of image nn into the other set.  The requests are dealt out a few per matrix instruction of the
	// tile's first row (the pipe takes a new instruction every 64 cycles: a request issued beside it is free, a burst of 18 + 9
	// in front of it is not); the other rows are a basic block each, skipped by a short tile (the end of a column or of
	// the wave's run).  The step opens with vmcnt(0) -- everything requested a step ago has had a step's worth of matrix
	// instructions to arrive; the compiler's own count would also wait for most of the requests just made, because its
	// scheduler does not request the two register sets in one order.  (wait = false: the caller has waited already.)
	auto step = [&](bool wait, const float (&b)[R + 2][3], const float (&a)[AV], int nn, const uint32_t (&voffn)[R + 2][3], int pn, int pln,
	                float (&bn)[R + 2][3], float (&an)[AV]) {
		if (wait) __builtin_amdgcn_s_waitcnt(0x0F70);
		__builtin_amdgcn_sched_barrier(0);
		fetch(nn, voffn, pn, pln, bn, an);
		row(0, b, a);
		constexpr int VM = (3 * (R + 2) + 9 * NT - 1) / (9 * NT);       // buffer loads per matrix instruction of the first row
#pragma unroll
		for (int i = 0; i < 9 * NT; ++i) {
			__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one matrix instruction
			__builtin_amdgcn_sched_group_barrier(0x020, VM, 0);          // buffer loads
			__builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // an LDS read
		}
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int r = 1; r < R; ++r)
			if (r < rows) row(r, b, a);
	};
	// a lane holds pixel x0 + nl of rows y0 .. y0 + rows - 1 for output channels co0 + t*32 + (i&3) + 8*(i>>2) + 4*kh
	auto store_tile = [&](int n, int x0, int y0) {
		const __amdgpu_buffer_rsrc_t ro = uniform_rsrc(out + (int64_t)n * Cout * HW, (uint32_t)Cout * plane);
		const int ox = x0 + nl;
		if ((Cout & 7) == 0) {
			// the eight channels co .. co + 7 of a register quartet exist or not as one: a wave-uniform test, no range check
			const bool whole = co0 + NT * 32 <= Cout;   // every channel of the group exists (64 of 64; the last group of 112 has 16 of 32)
			if (ox < W) {
#pragma unroll
				for (int r = 0; r < R; ++r) {
					if (r >= rows) break;
					__builtin_amdgcn_sched_barrier(0);   // a row's results at a time: hoisting all R * NT * 16 reads of the sums ahead of the stores would spill
					const uint32_t o0 = (uint32_t)(4 * kh) * plane + (uint32_t)((y0 + r) * W + ox) * 4u;
#pragma unroll
					for (int t = 0; t < NT; ++t) {
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							const int cq = co0 + t * 32 + 8 * q;
							if (whole || cq < Cout) {
#pragma unroll
								for (int j = 0; j < 4; ++j) {
									const float v = acc[r][t][4 * q + j];
									__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(RELU ? relu1(v) : v), ro, o0, (uint32_t)(cq + j) * plane, 0);
								}
							}
						}
					}
				}
			}
		} else {
#pragma unroll
			for (int r = 0; r < R; ++r) {
				if (r >= rows) break;
				__builtin_amdgcn_sched_barrier(0);
				const uint32_t o0 = ox < W ? (uint32_t)(4 * kh) * plane + (uint32_t)((y0 + r) * W + ox) * 4u : CV_OOB;
#pragma unroll
				for (int t = 0; t < NT; ++t) {
#pragma unroll
					for (int i = 0; i < 16; ++i) {
						const uint32_t cpl = (uint32_t)(co0 + t * 32 + (i & 3) + 8 * (i >> 2)) * plane;   // wave-uniform
						const float v = acc[r][t][i];
						// the buffer ends with channel Cout - 1, which drops the padded channels; o0 + cpl stays below 2^32 for every
						// lane that must store (api.hip checks Cout * plane), and a lane holding CV_OOB must stay out of range:
						// saturate instead of wrapping
						const uint32_t off = o0 > CV_OOB - cpl ? CV_OOB : o0 + cpl;
						__builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(RELU ? relu1(v) : v), ro, off, 0, 0);
					}
				}
			}
		}
	};
	auto start_sums = [&]() {
#pragma unroll
		for (int r = 0; r < R; ++r)
#pragma unroll
			for (int t = 0; t < NT; ++t)
#pragma unroll
				for (int i = 0; i < 16; ++i) acc[r][t][i] = bv[t][i];
	};

	if constexpr (!CHUNKED) {
		// The wave's run as ONE sequence of steps: the last pair of a tile requests pair 0 of the next tile.
		if (cur >= end) return;
		int n, x0, y0;
		uint32_t voff[R + 2][3];
		int tile = 0;
		place(cur, n, x0, y0, rows, phase_rows(0));
		offsets(x0, y0, voff);
		fetch(n, voff, 0, 0, b0, a0);
		__builtin_amdgcn_s_waitcnt(0x0F70);            // the one request of the run that nothing covers
		const int np = npairs;
		for (;;) {
			start_sums();
			const int nxt = cur + rows;
			const bool more = nxt < end;
			int n2, x2, y2, rows2;
			uint32_t voff2[R + 2][3];
			++tile;
			place(more ? nxt : cur, n2, x2, y2, rows2, more ? phase_rows(tile) : 0);   // a finished run requests its last tile again, and never uses it
			offsets(x2, y2, voff2);
			// two register sets of operands, used in turn.  The requests are straight-line code (one whose only consumer sits
			// behind a branch gets sunk into that branch by the compiler, in front of its own wait): the tile's last step
			// SELECTS the next tile's image and offsets instead of branching to another request.
#pragma unroll 1
			for (int pl = 0; pl < np; pl += 2) {
				step(pl > 0, b0, a0, n, voff, pl + 1, pl + 1, b1, a1);
				const bool last = pl + 2 >= np;
				uint32_t voffn[R + 2][3];
#pragma unroll
				for (int r = 0; r < R + 2; ++r)
#pragma unroll
					for (int k = 0; k < 3; ++k) voffn[r][k] = last ? voff2[r][k] : voff[r][k];
				step(true, b1, a1, last ? n2 : n, voffn, last ? 0 : pl + 2, last ? 0 : pl + 2, b0, a0);
			}
			__builtin_amdgcn_s_waitcnt(0x0F70);        // the next tile's pair 0 has had a step to arrive; waiting here, ahead of the
			store_tile(n, x0, y0);                     // stores, lets the next tile start without waiting for THEM
			if (!more) break;
			cur = nxt, n = n2, x0 = x2, y0 = y2, rows = rows2;
#pragma unroll
			for (int r = 0; r < R + 2; ++r)
#pragma unroll
				for (int k = 0; k < 3; ++k) voff[r][k] = voff2[r][k];
		}
	} else {
		const int nchunks = (npairs + chunk_pairs - 1) / chunk_pairs;
		for (;;) {
			const bool has = cur < end;
			if (!__syncthreads_or(has)) break;
			int n = 0, x0 = 0, y0 = 0;
			uint32_t voff[R + 2][3];
			rows = 0;
			if (has) place(cur, n, x0, y0, rows);
			offsets(x0, y0, voff);
			start_sums();
			for (int ch = 0; ch < nchunks; ++ch) {
				const int p0 = ch * chunk_pairs, np = min(chunk_pairs, npairs - p0);
				__syncthreads();           // every wave is done with the previous chunk's bank
				load_bank(p0, np);
				__syncthreads();
				if (!has) continue;
				fetch(n, voff, p0, 0, b0, a0);
#pragma unroll 1
				for (int pl = 0; pl < np; pl += 2) {       // the request past the chunk's last pair repeats that pair and is never used
					const int l2 = min(pl + 2, np - 1);
					step(true, b0, a0, n, voff, p0 + pl + 1, pl + 1, b1, a1);
					step(true, b1, a1, n, voff, p0 + l2, l2, b0, a0);
				}
			}
			if (has) {
				store_tile(n, x0, y0);
				cur += rows;
			}
		}
	}
}

// the split of a layer: NT tiles of 32 output channels per group, `groups` groups, pairs (even) per LDS-resident chunk
struct ConvPlan { int NT, groups, npairs, chunk_pairs; bool chunked; };

static ConvPlan conv_plan(int Cin, int Cout)
{
	ConvPlan p;
	p.npairs = ((Cin + 1) / 2 + 1) & ~1;
	const int tiles = (Cout + 31) / 32;
	auto bank = [&](int NT) { return (int64_t)p.npairs * 64 * 9 * NT * (int64_t)sizeof(float); };
	p.NT = (tiles % 2 == 0 && bank(2) <= CV_LDS_BYTES) ? 2 : 1;
	p.groups = tiles / p.NT;
	p.chunked = bank(p.NT) > CV_LDS_BYTES;
	if (p.chunked) {
		const int max_pairs = (CV_LDS_BYTES / (64 * 9 * p.NT * (int)sizeof(float))) & ~1;
		const int nchunks = (p.npairs + max_pairs - 1) / max_pairs;
		p.chunk_pairs = ((p.npairs + nchunks - 1) / nchunks + 1) & ~1;
	} else {
		p.chunk_pairs = p.npairs;
	}
	return p;
}

size_t conv3x3_workspace_bytes(int Cin, int Cout)
{
	const ConvPlan p = conv_plan(Cin, Cout);
	return ((size_t)p.groups * p.npairs * 64 * 9 * p.NT * sizeof(float) + 255) & ~(size_t)255;
}

static int conv_cus(void)   // CUs of the current device (256 on MI355X); the launch is one resident block per CU
{
	static int cus[64];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
	if (!cus[dev]) {
		int v = 0;
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
		cus[dev] = v;
	}
	return cus[dev];
}

template <int NT, bool RELU, bool CHUNKED>
static int conv_launch(const ConvPlan &p, const float *in, const float *wt, const float *bias, float *out, int N, int Cin, int Cout, int H, int W,
                       hipStream_t st)
{
	constexpr int R = cv_rows(NT);
	const int lds = p.chunk_pairs * 64 * 9 * NT * (int)sizeof(float);
	static bool raised[64];                          // the attribute belongs to (function, device)
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	if (!raised[dev]) {
		const hipError_t e = hipFuncSetAttribute((const void *)conv3x3_kernel<NT, RELU, CHUNKED>, hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES);
		if (e != hipSuccess) {
			set_error("conv3x3: hipFuncSetAttribute(%d bytes of LDS): %s", CV_LDS_BYTES, hipGetErrorString(e));
			return (int)e;
		}
		raised[dev] = true;
	}
	const int64_t T = (int64_t)N * ((W + 31) / 32) * H;                     // (image, strip, row) units
	const int64_t want = (T + (int64_t)CV_WAVES * R - 1) / ((int64_t)CV_WAVES * R);   // blocks that give every wave a full tile
	const int bpg = (int)max((int64_t)1, min((int64_t)(conv_cus() / p.groups), want));   // blocks per output-channel group
	const int64_t nw = (int64_t)bpg * CV_WAVES;
	hipLaunchKernelGGL((conv3x3_kernel<NT, RELU, CHUNKED>), dim3((unsigned)(bpg * p.groups)), dim3(256), lds, st, in, wt, bias, out, Cin, Cout, p.npairs,
	                   p.chunk_pairs, H, W, (int)(T / nw), (int)(T % nw), bpg);
	return 0;
}

int conv3x3(const float *in, const float *w, const float *bias, float *out, int N, int Cin, int Cout, int H, int W, int relu,
            void *workspace, hipStream_t st)
{
	const ConvPlan p = conv_plan(Cin, Cout);
	float *wt = (float *)workspace;
	const int64_t total = (int64_t)p.groups * p.npairs * 64 * 9 * p.NT;
	hipLaunchKernelGGL(conv_prep_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, wt, Cin, Cout, p.npairs, p.NT, p.groups);
	int rc = 0;
#define MC_CONV_GO(NT_, CH_) rc = relu ? conv_launch<NT_, true, CH_>(p, in, wt, bias, out, N, Cin, Cout, H, W, st) : conv_launch<NT_, false, CH_>(p, in, wt, bias, out, N, Cin, Cout, H, W, st)
	if (p.NT == 2) MC_CONV_GO(2, false);        // a 64-channel bank that fits is never chunked: conv_plan falls back to NT = 1 first
	else if (!p.chunked) MC_CONV_GO(1, false);
	else MC_CONV_GO(1, true);
#undef MC_CONV_GO
	if (rc) return rc;
	return check_launch("conv3x3");
}

}  // namespace mc
