// extern "C" surface of libmcadcensus.so (see include/mc_adcensus.h) and the fused
// stereo_predict pipeline (main.lua:929-1082).
#include "cbca_common.h"

#include <stdarg.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

namespace mc {

// kernels.hip units
int fill_nan(float *p, int64_t n, hipStream_t st);
int scale(const float *in, float *out, int64_t n, float s, hipStream_t st);
int transpose(const float *in, float *out, int64_t R, int64_t Cn, int64_t ldin, int64_t ldout, float s, hipStream_t st, int nt = -1);
int fix_border(float *vol, int D, int H, int W, int n, int direction, hipStream_t st);
int argmin_dhw(const float *vol, float *out, int D, int H, int W, int base1, hipStream_t st);
int argmin_hwd(const float *vol, float *out, int D, int ds, int H, int W, hipStream_t st);
int outlier_detection(const float *d0, const float *d1, float *outlier, int H, int W, int disp_max, hipStream_t st);
int interpolate_occlusion(const float *d0, const float *outlier, float *out, int H, int W, hipStream_t st);
int interpolate_mismatch(const float *d0, const float *outlier, float *out, int H, int W, hipStream_t st);
int subpixel(const float *d0, const float *vol, float *out, int D, int H, int W, int64_t sd, int64_t sp, hipStream_t st);
int median2d(const float *img, float *out, int H, int W, int k, hipStream_t st);
int mean2d(const float *img, const float *kernel, float *out, int H, int W, int ks, float alpha2, hipStream_t st);
int normalize_forward(const float *in, float *norm, float *out, int N, int C, int H, int W, hipStream_t st);
int stereo_join_dhw(const float *fL, const float *fR, float *volL, float *volR, int C, int D, int H, int W, hipStream_t st);
int stereo_join_hwd(const float *fL, const float *fR, float *volL, float *volR, int C, int D, int ds, int H, int W, int n,
                    hipStream_t st);
int ad_tiled(const float *x0, const float *x1, float *vol, int D, int H, int W, int direction, hipStream_t st);
size_t census_scratch_bytes(int Cimg, int H, int W);
int census_sig(const float *x0, const float *x1, float *vol, void *scratch, int Cimg, int D, int H, int W, int direction,
               hipStream_t st);
int cross(const float *img, float *arms, int H, int W, int L1, float tau1, hipStream_t st);
int cbca(const float *x0c, const float *x1c, const float *vin, float *vout, int D, int H, int W, int direction, hipStream_t st);
size_t cbca_scratch_bytes(int H, int W);
int cbca_pack(const float *x0c, const float *x1c, void *scratch, int H, int W, hipStream_t st);
int cbca_if_overflow(const float *x0c, const float *x1c, const void *packed, const float *vin, float *vout, int D, int H, int W,
                     int direction, hipStream_t st);
int cbca_strips(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int route,
               hipStream_t st, const CbcaCfg &cfg = CbcaCfg());
size_t cbca_plan_bytes(int D, int H, int W);
int cbca_tiles(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int arm_class, int route,
               hipStream_t st, const CbcaCfg &cfg = CbcaCfg());
bool cbca_lean_fits(int D, int H, int W, size_t plan_bytes, bool two_pass = false, int rb = 0);
int cbca_classify(const void *packed, void *plan, size_t plan_bytes, int D, int H, int W, int direction, int route, int rb, int cap_limit,
                  hipStream_t st, bool two_pass = false, float cost_limit = 0);
int cbca_lean2x(const void *packed, const void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
                int route, hipStream_t st, const CbcaCfg &cfg);
int cbca_lean(const void *packed, const void *plan, size_t plan_bytes, const float *vin, float *vout, int D, int H, int W, int direction,
              int route, hipStream_t st, const CbcaCfg &cfg);
size_t conv3x3_workspace_bytes(int Cin, int Cout);
int conv3x3(const float *in, const float *w, const float *bias, float *out, int N, int Cin, int Cout, int H, int W, int relu,
            void *workspace, hipStream_t st);
size_t fc_workspace_bytes(int C, int n_hidden, int H, int W);
int fc_stack(const float *featL, const float *featR, int C, int H, int W, int D, const float *const *weights,
             const float *const *biases, int n_layers, float *volL, float *volR, void *workspace, hipStream_t st);
size_t sgm_maps_bytes(int H, int W);
int sgm_prep(const float *x0, const float *x1, void *maps, int H, int W, float tau_so, hipStream_t st);
int sgm_contract_violations(const float *vol, int H, int W, int D, unsigned *count, hipStream_t st);
int sgm_sweeps(const float *const C[2], float *const out[2], float *const out2[2], float *const disp[2],
               const int direction[2], int nvol, int H, int W, int D, int ds, const void *maps, float pi1, float pi2,
               float alpha1, float q1, float q2, bool fused, hipStream_t st);

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

int check_launch(const char *what)
{
	const hipError_t e = hipPeekAtLastError();
	if (e != hipSuccess) {
		(void)hipGetLastError();
		set_error("%s: %s", what, hipGetErrorString(e));
		return (int)e;
	}
	return 0;
}

static bool dims_ok(int D, int H, int W) { return D >= 1 && H >= 1 && W >= 1 && (int64_t)D * H * W < ((int64_t)1 << 40); }

// gaussian(sigma), main.lua:528-540, double on the host.  Cached per sigma in PINNED host memory (allocated once, never
// freed), so that the per-call upload into the workspace is a true asynchronous copy that never sees its source disappear.
static void gaussian_fill(double sigma, float *k)
{
	const int kr = (int)ceil(sigma * 3);
	const int ks = kr * 2 + 1;
	for (int i = 1; i <= ks; ++i) {
		for (int j = 1; j <= ks; ++j) {
			const double y = (i - 1) - kr;
			const double x = (j - 1) - kr;
			k[(size_t)(i - 1) * ks + (j - 1)] = (float)exp(-(x * x + y * y) / (2 * sigma * sigma));
		}
	}
}
struct GaussianK { const float *data; size_t n; };
static int gaussian_cached(double sigma, GaussianK &out)
{
	static std::mutex mu;
	static std::map<double, GaussianK> cache;
	std::lock_guard<std::mutex> lk(mu);
	auto it = cache.find(sigma);
	if (it != cache.end()) { out = it->second; return 0; }
	const int kr = (int)ceil(sigma * 3);
	const int ks = kr * 2 + 1;
	float *k = nullptr;
	const hipError_t e = hipHostMalloc((void **)&k, (size_t)ks * ks * sizeof(float), hipHostMallocDefault);
	if (e != hipSuccess) {
		set_error("gaussian: hipHostMalloc: %s", hipGetErrorString(e));
		return (int)e;
	}
	gaussian_fill(sigma, k);
	out = GaussianK{k, (size_t)ks * ks};
	cache[sigma] = out;
	return 0;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// shapes the packed-length kernels (strip / tile / lean) take: 32-bit byte offsets inside a plane, and row indices that the tile
// kernel multiplies with 24-bit multiplies (an image of 2 x 10 000 000 pixels passes the first test only)
static inline bool packed_dims_ok(int H, int W)
{
	return (int64_t)H * W < ((int64_t)1 << 29) - 4096 && H < (1 << 23) - 64 && W < (1 << 23) - 256;
}

// One aggregation pass over packed arm lengths (cbca_pack).  max_arm = the largest arm that can occur (L1 - 1 where L1 is
// known, < 0 where it is not: adcensus.cbca).  Arms <= 4: the tile kernel's short-arm instance.  Otherwise the pair's
// route word (cbca_pack: arm classes actually present, share of pixels with unit arms) decides on the device between the
// tile kernel's two instances and the strip kernel -- the launches that are not the pair's stand down at their first
// instruction; nothing is read back by the host.  cfg.lean (mc_predict, from the first pass of a direction on): pairs of the
// strip kernel's route with arms <= 13 (textures) are served by the lean + list kernels out of the list cbca_classify wrote.
static int cbca_by_arms(const void *packed, const float *vin, float *vout, int D, int H, int W, int direction, int max_arm, hipStream_t st,
                        const CbcaCfg &cfg = CbcaCfg())
{
	if (max_arm >= 0 && max_arm <= 4) return cbca_tiles(packed, vin, vout, D, H, W, direction, 4, -1, st, cfg);
	if (cfg.planned) {
		// mc_predict with a plan area and 5 <= L1 <= 14: ONE candidate per pass beside the texture route's own kernel (round 4: three -- both tile
		// instances and the strip kernel, ~6 us each to stand down).  Arms <= 4 only: served by the long-arm instance; a texture whose list is
		// unusable (flat regions next to it): the tile kernel as well, with its plan -- not the strip kernel, whose compaction passes run at the pace of
		// the largest support
		int rc = cbca_tiles(packed, vin, vout, D, H, W, direction, 13, CR_PLANNED_TILE13, st, cfg);
		if (rc || cfg.lean_two_pass) return rc;   // (pairs of passes on a texture: cbca_lean2x, launched by the caller)
		return cbca_strips(packed, vin, vout, D, H, W, direction, CR_STRIP_IF_LIST, st, cfg);
	}
	int rc = cbca_tiles(packed, vin, vout, D, H, W, direction, 4, CR_TILE4, st, cfg);
	if (rc) return rc;
	if (max_arm < 0 || max_arm <= 13) {
		rc = cbca_tiles(packed, vin, vout, D, H, W, direction, 13, CR_TILE13, st, cfg);
		if (rc) return rc;
	}
	if (cfg.lean) {   // textures (route CR_STRIP) out of the pair's list: the lean + list kernels, the strip kernel only if the list is unusable
		if (!cfg.lean_two_pass) {   // (two passes per launch: cbca_lean2x, launched by the caller for a pair of passes)
			rc = cbca_lean(packed, cfg.plan, cfg.plan_bytes, vin, vout, D, H, W, direction, CR_STRIP, st, cfg);
			if (rc) return rc;
		}
		return cbca_strips(packed, vin, vout, D, H, W, direction, CR_STRIP_IF_NO_LIST, st, cfg);
	}
	return cbca_strips(packed, vin, vout, D, H, W, direction, max_arm > 13 ? CR_STRIP_OR_TILE13 : CR_STRIP, st, cfg);
}

struct Plan {
	int Dp;                 // padded pixel stride of the (H,W,Dp) volumes
	size_t maps, arms, pack, vol, img, gk;
	size_t cplan;           // per direction: the tile kernel's plan (cbca_tile.hip), 0 where it would not be reused
	int nplan;              // directions that get one
	size_t total;
};

// -sm_terminate / -sm_skip (main.lua:956,988-1040) are identical for both directions, so for the three volume stages they
// reduce to effective iteration counts
struct StageCounts { int cbca1, sgm, cbca2; bool active_after; };
static StageCounts stage_counts(const mc_params *p)
{
	StageCounts c;
	bool sm_active = p->sm_terminate != MC_SM_CNN;
	c.cbca1 = (sm_active && p->sm_skip != MC_SKIP_CBCA) ? p->cbca_i1 : 0;
	sm_active = sm_active && p->sm_terminate != MC_SM_CBCA1;
	c.sgm = (sm_active && p->sm_skip != MC_SKIP_SGM) ? p->sgm_i : 0;
	sm_active = sm_active && p->sm_terminate != MC_SM_SGM;
	c.cbca2 = (sm_active && p->sm_skip != MC_SKIP_CBCA) ? p->cbca_i2 : 0;
	sm_active = sm_active && p->sm_terminate != MC_SM_CBCA2;
	c.active_after = sm_active;
	return c;
}

static Plan make_plan(const mc_params *p, int D, int H, int W)
{
	Plan pl;
	// pixel stride of the (H,W,ds) volumes: D rounded up to 4 (16-byte runs).  Rounding up to 32 (every run on whole 128-byte
	// lines) was measured on one box at 370x1226x228 (ds 228 -> 256): the transposes 0.465 -> 0.377 ms, the sweeps 2.016 ->
	// 2.105 ms (12 % more bytes), StereoJoin unchanged; 6.000 -> 5.998 ms with the aggregation, 2.87 -> 2.97 ms without: not adopted.
	pl.Dp = (D + 3) / 4 * 4;
	const size_t HW = (size_t)H * W;
	pl.maps = align_up(sgm_maps_bytes(H, W), 256);
	pl.arms = align_up(8 * HW * sizeof(float), 256);
	pl.pack = cbca_scratch_bytes(H, W);
	pl.vol = align_up((size_t)pl.Dp * HW * sizeof(float), 256);
	pl.img = align_up(HW * sizeof(float), 256);
	const int kr = (int)ceil(p->blur_sigma * 3);
	const int ks = 2 * kr + 1;
	pl.gk = align_up((size_t)ks * ks * sizeof(float), 256);
	// the tile kernel's bookkeeping is the same in every aggregation pass over the pair and direction: kept from the first pass
	// on (cbca_plan_bytes: ~3.6 bytes per voxel) -- where a second pass exists to read it (after -sm_skip / -sm_terminate),
	// and per direction that is computed (left_only without the LR check: the left volume only; a caller who then asks for
	// right-side outputs gets that direction without a plan)
	const StageCounts sc = stage_counts(p);
	pl.cplan = (sc.cbca1 + sc.cbca2 >= 2 && p->L1 - 1 <= 13) ? align_up(cbca_plan_bytes(D, H, W), 256) : 0;
	pl.nplan = (p->left_only && !p->lr_check) ? 1 : 2;
	constexpr int NVOLS = 6;   // ping-pong per side (4) + the left sweep's partial sums per side
	pl.total = pl.maps + pl.arms + pl.pack + NVOLS * pl.vol + 6 * pl.img + pl.gk + pl.nplan * pl.cplan;
	return pl;
}

struct StageTimer {
	bool on = false;
	hipStream_t st;
	std::vector<hipEvent_t> ev;
	std::vector<int> tag;  // stage id the interval ENDING at this event belongs to
	void mark(int stage)
	{
		if (!on) return;
		hipEvent_t e;
		(void)hipEventCreate(&e);
		(void)hipEventRecord(e, st);
		ev.push_back(e);
		tag.push_back(stage);
	}
	void collect(float out[MC_N_STAGES])
	{
		for (int i = 0; i < MC_N_STAGES; ++i) out[i] = 0;
		if (!on || ev.empty()) return;
		(void)hipEventSynchronize(ev.back());
		for (size_t i = 1; i < ev.size(); ++i) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
			if (tag[i] >= 0 && tag[i] < MC_N_STAGES) out[tag[i]] += ms;
		}
		for (auto e : ev) (void)hipEventDestroy(e);
		ev.clear();
	}
};
enum { ST_PREP = MC_STAGE_PREP, ST_JOIN = MC_STAGE_JOIN, ST_CBCA = MC_STAGE_CBCA, ST_LAYOUT = MC_STAGE_LAYOUT,
       ST_SGM = MC_STAGE_SGM, ST_ARGMIN = MC_STAGE_ARGMIN, ST_POST = MC_STAGE_POST };

static int predict_impl(const mc_params *p, const float *x0, const float *x1, const float *featL, const float *featR, int C,
                        const float *rawL, const float *rawR, int D, int H, int W, void *workspace, size_t workspace_bytes,
                        float *volL_out, float *volR_out, float *dispL0_out, float *dispR0_out, float *disp_out, hipStream_t st,
                        StageTimer &tm)
{
	MC_REQUIRE(p && x0 && x1 && disp_out && workspace, "mc_predict: null argument");
	MC_REQUIRE(dims_ok(D, H, W), "mc_predict: bad dims D=%d H=%d W=%d", D, H, W);
	MC_REQUIRE(D <= MC_SGM_MAX_D, "mc_predict: D=%d exceeds %d", D, MC_SGM_MAX_D);
	MC_REQUIRE((featL && featR && C >= 1) || (rawL && rawR), "mc_predict: need features or raw volumes");
	MC_REQUIRE(p->cbca_i1 >= 0 && p->cbca_i2 >= 0 && p->sgm_i >= 0, "mc_predict: negative iteration count");
	MC_REQUIRE(p->median_k % 2 == 1 && p->median_k <= 11, "mc_predict: median_k must be odd and <= 11");
	MC_REQUIRE(p->blur_sigma > 0, "mc_predict: blur_sigma must be > 0");
	MC_REQUIRE(p->sm_terminate >= 0 && p->sm_terminate <= MC_SM_BILATERAL && p->sm_skip >= 0 && p->sm_skip <= MC_SKIP_BILATERAL,
	           "mc_predict: bad sm_terminate / sm_skip");
	const bool from_feat = featL != nullptr;
	if (from_feat) MC_REQUIRE(p->border_n >= 0 && p->border_n < W, "mc_predict: border_n=%d out of range", p->border_n);
	if (from_feat) MC_REQUIRE(C <= MC_JOIN_MAX_C, "mc_predict: C=%d exceeds %d (adcensus.cu:1460)", C, MC_JOIN_MAX_C);
	const Plan pl = make_plan(p, D, H, W);
	MC_REQUIRE(workspace_bytes >= pl.total, "mc_predict: workspace %zu < %zu bytes", workspace_bytes, pl.total);
	MC_REQUIRE((uintptr_t)workspace % 256 == 0, "mc_predict: workspace must be 256-byte aligned");

	const int64_t HW = (int64_t)H * W;
	const int64_t V = (int64_t)D * HW;
	char *w = (char *)workspace;
	void *maps = w; w += pl.maps;
	float *x0c = (float *)w; float *x1c = x0c + 4 * HW; w += pl.arms;
	void *packed = w; w += pl.pack;
	float *bufA[2], *bufB[2];
	bufA[0] = (float *)w; w += pl.vol;
	bufA[1] = (float *)w; w += pl.vol;
	bufB[0] = (float *)w; w += pl.vol;
	bufB[1] = (float *)w; w += pl.vol;
	float *bufC[2];  // scratch of the SGM's concurrent second direction
	bufC[0] = (float *)w; w += pl.vol;
	bufC[1] = (float *)w; w += pl.vol;
	float *img[6];
	for (int i = 0; i < 6; ++i) { img[i] = (float *)w; w += pl.img; }
	float *gk = (float *)w; w += pl.gk;
	void *cplan[2] = {pl.cplan ? (void *)w : nullptr, (pl.cplan && pl.nplan > 1) ? (void *)(w + pl.cplan) : nullptr};
	int cplan_passes[2] = {0, 0};   // aggregation passes so far: the first one writes the plan, the others read it
	bool cplan_listed[2] = {false, false};   // ... the texture route's list has been written
	const int Dp = pl.Dp;
	int rc;
#define RUN(call) do { rc = (call); if (rc) return rc; } while (0)

	// index 0 = left volume (direction -1), 1 = right volume (direction +1)  (main.lua:986)
	const int direction[2] = {-1, 1};
	const StageCounts sc = stage_counts(p);   // -sm_terminate / -sm_skip as effective iteration counts
	const int n_cbca1 = sc.cbca1, n_sgm = sc.sgm, n_cbca2 = sc.cbca2;
	bool sm_active = sc.active_after;
	const bool use_cbca = (n_cbca1 + n_cbca2) > 0;
	tm.mark(-1);

	if (n_sgm > 0) RUN(sgm_prep(x0, x1, maps, H, W, p->tau_so, st));
	if (use_cbca) {  // main.lua:993-996: x0c from the LEFT image, x1c from the RIGHT, for both directions
		RUN(cross(x0, x0c, H, W, p->L1, p->tau1, st));
		RUN(cross(x1, x1c, H, W, p->L1, p->tau1, st));
		RUN(cbca_pack(x0c, x1c, packed, H, W, st));
	}
	const int cbca_cap = p->L1 - 1;  // cross(): an arm never exceeds L1-1 pixels (adcensus.cu:314)
	tm.mark(ST_PREP);

	// ---- (A) cost volumes + CBCA-1; ends with cur[v] and its layout ----
	// Internal (H,W,ds) volumes use the padded pixel stride ds = Dp (multiple of 4) so that
	// every lane's run of 4 disparities is one aligned 16-byte access.
	const int ds = Dp;
	const float *cur[2];
	auto other = [&](int v) -> float * { return cur[v] == bufA[v] ? bufB[v] : bufA[v]; };
	bool hwd;  // layout of cur[]
	// direction +1 (the right volume) is skipped where the reference skips it: dataset mb outside `-a predict`
	// (mb_directions, main.lua:953-955) -- only when nothing of it is asked for
	const int nvol = (p->left_only && !p->lr_check && !volR_out && !dispR0_out) ? 1 : 2;
	// n CBCA iterations on the (D,H,W) volumes, ping-pong between the two buffers of each side (instead of vol:copy(tmp)).
	// Where the pair's route may be the texture one (cfg.lean), the iterations go in PAIRS: cbca_lean2x runs two passes in one launch
	// (cur -> dst) out of a list written once per pair and direction; the kernels of the other routes -- which one runs is decided on
	// the device by the route word, the others stand down at their first instruction -- take the same two passes through the SGM's
	// scratch volume (cur -> bufC -> dst), so that every route ends in the same buffer.  An odd last iteration is a single pass.
	auto cbca_iterations = [&](int n) -> int {
		const bool packed_ok = cbca_cap <= 254 && packed_dims_ok(H, W);  // packed lengths saturate at 255
		for (int v = 0; v < nvol; ++v) {
			// the plan area serves whichever kernel the pair's route word picks: the tile kernel's plan (written by its first pass) or, on
			// textures, the records of the outputs whose support is not the minimal 3 x 3 -- written HERE, before the direction's first pass
			// (route-gated, on the device), so that every later launch finds the list's state final: usable -> cbca_lean2x per pair of passes
			// and the strip kernel for a single one; unusable (flat regions next to the texture) -> the tile kernel
			const bool planned = packed_ok && cplan[v] && cbca_cap > 4 && cbca_cap <= 13 && cbca_lean_fits(D, H, W, pl.cplan, true);
			if (planned && n > 0 && !cplan_listed[v]) {
				const int rc1 = cbca_classify(packed, cplan[v], pl.cplan, D, H, W, direction[v], CR_STRIP, 0, 0, st, true);
				if (rc1) return rc1;
				cplan_listed[v] = true;
			}
			for (int i = 0; i < n;) {
				float *dst = other(v);
				CbcaCfg cfg;
				cfg.plan = cplan[v];
				cfg.plan_bytes = pl.cplan;
				cfg.planned = planned;
				const bool two = planned && i + 1 < n;
				cfg.lean = cfg.lean_two_pass = two;   // (a single pass: the strip kernel serves the texture route itself)
				float *mid = two ? bufC[v] : dst;
				for (int half = 0; half < (two ? 2 : 1); ++half) {   // the routes that take one pass per launch
					cfg.plan_mode = cplan[v] ? (cplan_passes[v]++ == 0 ? 1 : 2) : 0;
					const float *src = half == 0 ? cur[v] : mid;
					float *to = (two && half == 0) ? mid : dst;
					const int rc2 = packed_ok ? cbca_by_arms(packed, src, to, D, H, W, direction[v], cbca_cap, st, cfg)
					                          : cbca(x0c, x1c, src, to, D, H, W, direction[v], st);
					if (rc2) return rc2;
				}
				if (two) {
					const int rc3 = cbca_lean2x(packed, cplan[v], pl.cplan, cur[v], dst, D, H, W, direction[v], CR_STRIP, st, cfg);
					if (rc3) return rc3;
				}
				cur[v] = dst;
				i += two ? 2 : 1;
			}
		}
		return 0;
	};
	// (C + 64: the kernels pad the channel count to their k-step sizes and add channel offsets to range-checked 32-bit byte
	// offsets -- no padded channel's offset may wrap)
	const bool join_fits = (int64_t)W * ((D + 3) / 4 * 4) * 4 < ((int64_t)1 << 31) && ((int64_t)(C + 64) * HW + W) * 4 < ((int64_t)1 << 31);
	if (from_feat && n_cbca1 == 0 && n_sgm > 0 && join_fits) {
		// fast path: StereoJoin straight into (H,W,ds) with NaN fill and fix_border folded in
		RUN(stereo_join_hwd(featL, featR, bufA[0], bufA[1], C, D, ds, H, W, p->border_n, st));
		cur[0] = bufA[0]; cur[1] = bufA[1];
		hwd = true;
		tm.mark(ST_JOIN);
	} else {
		if (from_feat) {  // main.lua:946-949
			RUN(fill_nan(bufA[0], V, st));
			RUN(fill_nan(bufA[1], V, st));
			RUN(stereo_join_dhw(featL, featR, bufA[0], bufA[1], C, D, H, W, st));
			RUN(fix_border(bufA[0], D, H, W, p->border_n, -1, st));
			RUN(fix_border(bufA[1], D, H, W, p->border_n, 1, st));
			cur[0] = bufA[0]; cur[1] = bufA[1];
		} else {
			cur[0] = rawL; cur[1] = rawR;
		}
		hwd = false;
		tm.mark(ST_JOIN);
		RUN(cbca_iterations(n_cbca1));  // main.lua:998-1001 (ping-pong instead of vol:copy(tmp))
		tm.mark(ST_CBCA);
	}

	// ---- (B) SGM, main.lua:1007-1030 ----
	float *dispv[2] = {img[0], img[1]};  // [0] = left disparity (disp[2] in Lua), [1] = right
	bool have_disp = false;
	if (n_sgm > 0) {
		if (!hwd) {  // vol:transpose(2,3):transpose(3,4):clone(), main.lua:1008
			for (int v = 0; v < nvol; ++v) {
				float *dst = other(v);
				RUN(transpose(cur[v], dst, D, HW, HW, ds, 1.0f, st));
				cur[v] = dst;
			}
			hwd = true;
			tm.mark(ST_LAYOUT);
		}
		for (int it = 0; it < n_sgm; ++it) {
			// out:zero(); sgm2(...); vol:copy(out):div(4)  (main.lua:1013-1018): the zero is folded
			// into the first sweep (0 + L_0) and the /4 into the last
			const float *Cv[2] = {cur[0], cur[1]};
			float *outv[2] = {other(0), other(1)};
			const bool am = (it == n_sgm - 1) && n_cbca2 == 0;
			RUN(sgm_sweeps(Cv, outv, bufC, am ? dispv : nullptr, direction, nvol, H, W, D, ds, maps, p->pi1, p->pi2, p->alpha1,
			               p->sgm_q1, p->sgm_q2, true, st));
			have_disp = am;
			cur[0] = outv[0]; cur[1] = outv[1];
		}
		tm.mark(ST_SGM);
		if (n_cbca2 > 0) {  // back to (D,H,W): vol:copy(out:transpose(3,4):transpose(2,3)), main.lua:1019-1020
			for (int v = 0; v < nvol; ++v) {
				float *dst = other(v);
				RUN(transpose(cur[v], dst, HW, D, ds, HW, 1.0f, st));
				cur[v] = dst;
			}
			hwd = false;
			tm.mark(ST_LAYOUT);
		}
	}
	if (!hwd) {  // CBCA-2, main.lua:1033-1039
		RUN(cbca_iterations(n_cbca2));
		tm.mark(ST_CBCA);
	}

	// ---- argmin, main.lua:1049-1050 ----
	if (!have_disp) {
		for (int v = 0; v < nvol; ++v) {
			if (hwd) RUN(argmin_hwd(cur[v], dispv[v], D, ds, H, W, st));
			else RUN(argmin_dhw(cur[v], dispv[v], D, H, W, 0, st));
		}
	}
	// ---- left.bin / right.bin contents, main.lua:1042-1047 ----
	float *vout[2] = {volL_out, volR_out};
	for (int v = 0; v < nvol; ++v) {
		if (!vout[v]) continue;
		if (hwd) RUN(transpose(cur[v], vout[v], HW, D, ds, HW, 1.0f, st));
		else RUN(scale(cur[v], vout[v], V, 1.0f, st));
	}
	if (dispL0_out) RUN(scale(dispv[0], dispL0_out, HW, 1.0f, st));
	if (dispR0_out) RUN(scale(dispv[1], dispR0_out, HW, 1.0f, st));
	tm.mark(ST_ARGMIN);

	// ---- (C) post-processing on the LEFT disparity, main.lua:1054-1081 ----
	// every stage reads `d` and writes the next free image; -sm_skip / -sm_terminate drop stages (main.lua:1057-1079)
	float *d = dispv[0];
	float *outl = img[4];
	int nfree = 0;
	float *freeimg[3] = {img[2], img[3], img[5]};
	auto next = [&]() -> float * {
		float *r = freeimg[nfree % 3];
		if (r == d) r = freeimg[++nfree % 3];
		++nfree;
		return r;
	};
	if (p->lr_check) {
		RUN(outlier_detection(d, dispv[1], outl, H, W, D, st));
		if (sm_active && p->sm_skip != MC_SKIP_OCCLUSION) {
			float *o = next();
			RUN(interpolate_occlusion(d, outl, o, H, W, st));
			d = o;
		}
		sm_active = sm_active && p->sm_terminate != MC_SM_OCCLUSION;
		if (sm_active && p->sm_skip != MC_SKIP_OCCLUSION) {
			float *o = next();
			RUN(interpolate_mismatch(d, outl, o, H, W, st));
			d = o;
		}
		sm_active = sm_active && p->sm_terminate != MC_SM_MISMATCH;
	}
	if (sm_active && p->sm_skip != MC_SKIP_SUBPIXEL) {
		// subpixel on the LEFT volume (vol of the last loop iteration, main.lua:1068)
		float *o = next();
		if (hwd) RUN(subpixel(d, cur[0], o, D, H, W, 1, ds, st));
		else RUN(subpixel(d, cur[0], o, D, H, W, HW, 1, st));
		d = o;
	}
	sm_active = sm_active && p->sm_terminate != MC_SM_SUBPIXEL;
	if (sm_active && p->sm_skip != MC_SKIP_MEDIAN) {
		float *o = next();
		RUN(median2d(d, o, H, W, p->median_k, st));
		d = o;
	}
	sm_active = sm_active && p->sm_terminate != MC_SM_MEDIAN;
	if (sm_active && p->sm_skip != MC_SKIP_BILATERAL) {
		GaussianK k;
		RUN(gaussian_cached(p->blur_sigma, k));
		const int ks = 2 * (int)ceil(p->blur_sigma * 3) + 1;
		const hipError_t e = hipMemcpyAsync(gk, k.data, k.n * sizeof(float), hipMemcpyHostToDevice, st);  // pinned source
		if (e != hipSuccess) {
			set_error("mc_predict: kernel upload: %s", hipGetErrorString(e));
			return (int)e;
		}
		RUN(mean2d(d, gk, disp_out, H, W, ks, p->blur_t, st));
	} else {
		RUN(scale(d, disp_out, HW, 1.0f, st));  // the last stage that ran is the result
	}
	tm.mark(ST_POST);
#undef RUN
	return 0;
}

}  // namespace mc

using namespace mc;

extern "C" {

int mc_version(void) { return MC_ABI_VERSION; }
const char *mc_last_error(void) { return g_err; }

int mc_fill_nan(float *p, int64_t n, void *stream)
{
	MC_REQUIRE(p || n == 0, "mc_fill_nan: null pointer");
	MC_REQUIRE(n >= 0, "mc_fill_nan: negative size");
	return fill_nan(p, n, as_stream(stream));
}

int mc_stereo_join(const float *featL, const float *featR, float *volL, float *volR, int C, int D, int H, int W, void *stream)
{
	MC_REQUIRE(featL && featR && volL && volR, "mc_stereo_join: null pointer");
	MC_REQUIRE(dims_ok(D, H, W) && C >= 1, "mc_stereo_join: bad dims C=%d D=%d H=%d W=%d", C, D, H, W);
	MC_REQUIRE(C <= MC_JOIN_MAX_C, "mc_stereo_join: C=%d exceeds %d (adcensus.cu:1460)", C, MC_JOIN_MAX_C);
	MC_REQUIRE(D <= 65535, "mc_stereo_join: D too large");
	return stereo_join_dhw(featL, featR, volL, volR, C, D, H, W, as_stream(stream));
}

int mc_ad(const float *x0, const float *x1, float *vol, int D, int H, int W, int direction, void *stream)
{
	MC_REQUIRE(x0 && x1 && vol, "mc_ad: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_ad: bad dims");
	MC_REQUIRE(direction == -1 || direction == 1, "mc_ad: direction must be -1 or 1");
	MC_REQUIRE(D <= 65535, "mc_ad: D too large");
	return ad_tiled(x0, x1, vol, D, H, W, direction, as_stream(stream));
}

size_t mc_census_scratch_bytes(int Cimg, int H, int W)
{
	if (Cimg < 1 || H < 1 || W < 1) return 0;
	return census_scratch_bytes(Cimg, H, W);
}

int mc_census_ws(const float *x0, const float *x1, float *vol, int Cimg, int D, int H, int W, int direction, void *scratch,
                 size_t scratch_bytes, void *stream)
{
	MC_REQUIRE(x0 && x1 && vol && scratch, "mc_census_ws: null pointer");
	MC_REQUIRE(dims_ok(D, H, W) && Cimg >= 1 && D <= 65535, "mc_census_ws: bad dims");
	MC_REQUIRE(direction == -1 || direction == 1, "mc_census_ws: direction must be -1 or 1");
	MC_REQUIRE(scratch_bytes >= census_scratch_bytes(Cimg, H, W), "mc_census_ws: scratch holds %zu bytes, needs %zu", scratch_bytes,
	           census_scratch_bytes(Cimg, H, W));
	MC_REQUIRE((uintptr_t)scratch % 4 == 0, "mc_census_ws: scratch must be 4-byte aligned");
	return census_sig(x0, x1, vol, scratch, Cimg, D, H, W, direction, as_stream(stream));
}

size_t mc_fc_stack_workspace_bytes(int C, int n_layers, int H, int W)
{
	if (C < 1 || n_layers < 2 || n_layers > 8 || H < 1 || W < 1) return 0;
	return fc_workspace_bytes(C, n_layers - 2, H, W);
}

int mc_fc_stack(const float *featL, const float *featR, int C, int H, int W, int D, const float *const *weights,
                const float *const *biases, const int *layer_out, int n_layers, float *volL, float *volR, void *workspace,
                size_t workspace_bytes, void *stream)
{
	MC_REQUIRE(featL && featR && weights && biases && layer_out && volL && volR && workspace, "mc_fc_stack: null pointer");
	MC_REQUIRE(dims_ok(D, H, W) && C >= 1, "mc_fc_stack: bad dims");
	MC_REQUIRE(n_layers >= 2 && n_layers <= 8, "mc_fc_stack: 2..8 layers supported");
	for (int l = 0; l < n_layers - 1; ++l)
		MC_REQUIRE(layer_out[l] == 384, "mc_fc_stack: hidden width %d not supported (nh2 = 384 in every preset, main.lua:77,124)",
		           layer_out[l]);
	MC_REQUIRE(layer_out[n_layers - 1] == 1, "mc_fc_stack: the last layer must have one output");
	for (int l = 0; l < n_layers; ++l) MC_REQUIRE(weights[l] && biases[l], "mc_fc_stack: null layer %d", l);
	MC_REQUIRE(workspace_bytes >= fc_workspace_bytes(C, n_layers - 2, H, W), "mc_fc_stack: workspace %zu < %zu bytes",
	           workspace_bytes, fc_workspace_bytes(C, n_layers - 2, H, W));
	MC_REQUIRE((uintptr_t)workspace % 16 == 0, "mc_fc_stack: workspace must be 16-byte aligned");
	MC_REQUIRE((int64_t)((H + 7) / 8) * ((W + 95) / 96) * D * 8 < ((int64_t)1 << 31), "mc_fc_stack: problem too large for one launch");
	return fc_stack(featL, featR, C, H, W, D, weights, biases, n_layers, volL, volR, workspace, as_stream(stream));
}

size_t mc_conv3x3_workspace_bytes(int Cin, int Cout)
{
	if (Cin < 1 || Cout < 1 || Cout > 128) return 0;
	return conv3x3_workspace_bytes(Cin, Cout);
}

int mc_conv3x3(const float *in, const float *weight, const float *bias, float *out, int N, int Cin, int Cout, int H, int W,
               int relu, void *workspace, size_t workspace_bytes, void *stream)
{
	MC_REQUIRE(in && weight && bias && out && workspace && in != out, "mc_conv3x3: bad pointers");
	MC_REQUIRE(N >= 1 && Cin >= 1 && Cout >= 1 && dims_ok(1, H, W), "mc_conv3x3: bad dims");
	MC_REQUIRE(Cout <= 128, "mc_conv3x3: Cout=%d exceeds 128 (the nets of main.lua:73-75, 120-122 use 64 and 112)", Cout);
	// 32-bit byte offsets inside one image's planes (buffer addressing) and 32-bit unit counts
	MC_REQUIRE((int64_t)(Cout > Cin ? Cout : Cin + 1) * H * W * 4 < (int64_t)0xFFFFFF00 && (int64_t)N * ((W + 31) / 32) * H < ((int64_t)1 << 31),
	           "mc_conv3x3: %d x %d x (%d -> %d channels) x %d images exceeds the kernel's 32-bit offsets", H, W, Cin, Cout, N);
	MC_REQUIRE(workspace_bytes >= conv3x3_workspace_bytes(Cin, Cout), "mc_conv3x3: workspace %zu < %zu bytes", workspace_bytes,
	           conv3x3_workspace_bytes(Cin, Cout));
	MC_REQUIRE((uintptr_t)workspace % 16 == 0, "mc_conv3x3: workspace must be 16-byte aligned");
	return conv3x3(in, weight, bias, out, N, Cin, Cout, H, W, relu, workspace, as_stream(stream));
}

int mc_fix_border(float *vol, int D, int H, int W, int n, int direction, void *stream)
{
	MC_REQUIRE(vol, "mc_fix_border: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_fix_border: bad dims");
	MC_REQUIRE(n >= 0 && n < W, "mc_fix_border: n=%d out of range for W=%d", n, W);
	MC_REQUIRE(direction == -1 || direction == 1, "mc_fix_border: direction must be -1 or 1");
	return fix_border(vol, D, H, W, n, direction, as_stream(stream));
}

int mc_cross(const float *img, float *arms, int H, int W, int L1, float tau1, void *stream)
{
	MC_REQUIRE(img && arms, "mc_cross: null pointer");
	MC_REQUIRE(dims_ok(1, H, W), "mc_cross: bad dims");
	return cross(img, arms, H, W, L1, tau1, as_stream(stream));
}

int mc_cbca(const float *x0c, const float *x1c, const float *vol_in, float *vol_out, int D, int H, int W, int direction,
            void *stream)
{
	MC_REQUIRE(x0c && x1c && vol_in && vol_out, "mc_cbca: null pointer");
	MC_REQUIRE(vol_in != vol_out, "mc_cbca: in-place aggregation is not supported (the reference uses a tmp volume too)");
	MC_REQUIRE(dims_ok(D, H, W) && D <= 65535, "mc_cbca: bad dims");
	MC_REQUIRE(direction == -1 || direction == 1, "mc_cbca: direction must be -1 or 1");
	return cbca(x0c, x1c, vol_in, vol_out, D, H, W, direction, as_stream(stream));
}

size_t mc_cbca_scratch_bytes(int H, int W)
{
	if (H < 1 || W < 1) return 0;
	return cbca_scratch_bytes(H, W);
}

int mc_cbca_ws(const float *x0c, const float *x1c, const float *vol_in, float *vol_out, int D, int H, int W, int direction,
               void *scratch, size_t scratch_bytes, void *stream)
{
	MC_REQUIRE(x0c && x1c && vol_in && vol_out && scratch, "mc_cbca_ws: null pointer");
	MC_REQUIRE(vol_in != vol_out, "mc_cbca_ws: in-place aggregation is not supported (the reference uses a tmp volume too)");
	MC_REQUIRE(dims_ok(D, H, W) && D <= 65535 * 8, "mc_cbca_ws: bad dims");
	MC_REQUIRE(direction == -1 || direction == 1, "mc_cbca_ws: direction must be -1 or 1");
	MC_REQUIRE(scratch_bytes >= cbca_scratch_bytes(H, W), "mc_cbca_ws: scratch holds %zu bytes, needs %zu", scratch_bytes,
	           cbca_scratch_bytes(H, W));
	MC_REQUIRE((uintptr_t)scratch % 4 == 0, "mc_cbca_ws: scratch must be 4-byte aligned");
	hipStream_t st = as_stream(stream);
	if (!packed_dims_ok(H, W))   // (huge or extremely elongated images: one thread per voxel, 64-bit offsets)
		return cbca(x0c, x1c, vol_in, vol_out, D, H, W, direction, st);
	int rc = cbca_pack(x0c, x1c, scratch, H, W, st);
	if (rc) return rc;
	rc = cbca_by_arms(scratch, vol_in, vol_out, D, H, W, direction, -1, st);
	if (rc) return rc;
	return cbca_if_overflow(x0c, x1c, scratch, vol_in, vol_out, D, H, W, direction, st);
}

size_t mc_cbca_plan_bytes(int D, int H, int W)
{
	if (!dims_ok(D, H, W)) return 0;
	return align_up(cbca_scratch_bytes(H, W), 256) - cbca_scratch_bytes(H, W) + cbca_plan_bytes(D, H, W);
}

int mc_cbca_ws_cfg(const float *x0c, const float *x1c, const float *vol_in, float *vol_out, int D, int H, int W, int direction,
                   void *scratch, size_t scratch_bytes, int rb, int nt, int d0, int nd, int form, void *stream)
{
	MC_REQUIRE(x0c && x1c && vol_in && vol_out && scratch, "mc_cbca_ws_cfg: null pointer");
	MC_REQUIRE(vol_in != vol_out, "mc_cbca_ws_cfg: in-place aggregation is not supported");
	MC_REQUIRE(dims_ok(D, H, W) && D <= 65535 * 8, "mc_cbca_ws_cfg: bad dims");
	MC_REQUIRE(direction == -1 || direction == 1, "mc_cbca_ws_cfg: direction must be -1 or 1");
	MC_REQUIRE(scratch_bytes >= cbca_scratch_bytes(H, W), "mc_cbca_ws_cfg: scratch holds %zu bytes, needs %zu", scratch_bytes,
	           cbca_scratch_bytes(H, W));
	MC_REQUIRE((uintptr_t)scratch % 4 == 0, "mc_cbca_ws_cfg: scratch must be 4-byte aligned");
	MC_REQUIRE(packed_dims_ok(H, W), "mc_cbca_ws_cfg: image too large for the packed-length kernels (32-bit plane offsets, 24-bit row indices)");
	MC_REQUIRE(rb >= 0 && rb <= 4096 && nt >= -1 && nt <= 1 && form >= 0 && form <= 11, "mc_cbca_ws_cfg: bad rb / nt / form");
	MC_REQUIRE(d0 >= 0 && nd >= 0 && (form >= 8 || d0 + nd <= D), "mc_cbca_ws_cfg: planes [%d, %d) outside the volume", d0, d0 + nd);
	hipStream_t st = as_stream(stream);
	int rc = cbca_pack(x0c, x1c, scratch, H, W, st);
	if (rc) return rc;
	CbcaCfg cfg;
	cfg.nt = nt; cfg.d0 = d0; cfg.nd = nd;
	if (form >= 10) {   // TWO passes in one launch (cbca_lean2x): 10 writes the list of its wave geometry first, 11 reads it; vol_out = the volume after
		// the second pass.  The strip kernel takes both passes (through a volume behind the list) if the list is not this problem's or did not fit.
		const size_t off = align_up(cbca_scratch_bytes(H, W), 256), pb = align_up(cbca_plan_bytes(D, H, W), 256);
		const size_t vb = (size_t)D * H * W * sizeof(float);
		MC_REQUIRE(scratch_bytes >= off + pb + vb, "mc_cbca_ws_cfg: scratch holds %zu bytes, needs %zu with the list and a volume", scratch_bytes, off + pb + vb);
		MC_REQUIRE((uintptr_t)scratch % 16 == 0, "mc_cbca_ws_cfg: scratch must be 16-byte aligned for the list");
		cfg.plan = (char *)scratch + off;
		cfg.plan_bytes = pb;
		cfg.lean_rb = rb;
		cfg.lean_two_pass = true;
		cfg.d0 = 0; cfg.nd = 0;
		float *mid = (float *)((char *)scratch + off + pb);
		const bool fits = cbca_lean_fits(D, H, W, pb, true, rb);   // (the records of this many rows per wave fit the area; else: the strip kernel, unconditionally)
		if (fits && form == 10) {
			rc = cbca_classify(scratch, cfg.plan, cfg.plan_bytes, D, H, W, direction, CR_NOT_DIRECT, rb, 0, st, true, (float)d0);   // (d0 > 0: the cost limit, in values per voxel)
			if (rc) return rc;
		}
		if (fits) {
			rc = cbca_lean2x(scratch, cfg.plan, cfg.plan_bytes, vol_in, vol_out, D, H, W, direction, CR_NOT_DIRECT, st, cfg);
			if (rc) return rc;
		}
		const int sroute = fits ? CR_NOT_DIRECT_IF_NO_LIST : CR_NOT_DIRECT;
		rc = cbca_strips(scratch, vol_in, mid, D, H, W, direction, sroute, st, cfg);
		if (rc) return rc;
		rc = cbca_strips(scratch, mid, vol_out, D, H, W, direction, sroute, st, cfg);
		if (rc) return rc;
		rc = cbca_if_overflow(x0c, x1c, scratch, vol_in, mid, D, H, W, direction, st);
		if (rc) return rc;
		return cbca_if_overflow(x0c, x1c, scratch, mid, vol_out, D, H, W, direction, st);
	}
	if (form >= 8) {   // lean kernel (textures): 8 lists the outputs whose support is not the minimal 3 x 3 behind the packed lengths first, 9 reads that list
		const size_t off = align_up(cbca_scratch_bytes(H, W), 256), pb = cbca_plan_bytes(D, H, W);
		MC_REQUIRE(scratch_bytes >= off + pb, "mc_cbca_ws_cfg: scratch holds %zu bytes, needs %zu with the list", scratch_bytes, off + pb);
		MC_REQUIRE((uintptr_t)scratch % 16 == 0, "mc_cbca_ws_cfg: scratch must be 16-byte aligned for the list");
		MC_REQUIRE(cbca_lean_fits(D, H, W, pb), "mc_cbca_ws_cfg: volume too large for 32-bit list entries");
		// (forms 8 / 9 take the whole volume; rb = rows per wave, d0 = launch variant, nd > 0 = slots the list may hold, to exercise the fallback)
		cfg.plan = (char *)scratch + off;
		cfg.plan_bytes = pb;
		cfg.lean_rb = rb;
		cfg.lean_variant = d0; cfg.d0 = 0; cfg.nd = nd;
		if (form == 8) {
			rc = cbca_classify(scratch, cfg.plan, cfg.plan_bytes, D, H, W, direction, CR_NOT_DIRECT, rb, nd, st);
			if (rc) return rc;
		}
		rc = cbca_lean(scratch, cfg.plan, cfg.plan_bytes, vol_in, vol_out, D, H, W, direction, CR_NOT_DIRECT, st, cfg);
		if (rc) return rc;
		cfg.nd = 0;
		rc = cbca_strips(scratch, vol_in, vol_out, D, H, W, direction, CR_NOT_DIRECT_IF_NO_LIST, st, cfg);
		if (rc) return rc;
		return cbca_if_overflow(x0c, x1c, scratch, vol_in, vol_out, D, H, W, direction, st);
	}
	if (form >= 4) {   // tile kernel with the item order kept behind the packed lengths: 4 / 5 write it (short- / long-arm instance), 6 / 7 read it
		const size_t off = align_up(cbca_scratch_bytes(H, W), 256);
		MC_REQUIRE(scratch_bytes >= off + cbca_plan_bytes(D, H, W), "mc_cbca_ws_cfg: scratch holds %zu bytes, needs %zu with the plan", scratch_bytes,
		           off + cbca_plan_bytes(D, H, W));
		MC_REQUIRE((uintptr_t)scratch % 16 == 0, "mc_cbca_ws_cfg: scratch must be 16-byte aligned for the plan");
		cfg.plan = (char *)scratch + off;
		cfg.plan_mode = form <= 5 ? 1 : 2;
		const bool shortarm = form == 4 || form == 6;
		return cbca_tiles(scratch, vol_in, vol_out, D, H, W, direction, shortarm ? 4 : 13, shortarm ? CR_ARMS_LE4 : CR_ARMS_LE13, st, cfg);
	}
	if (form >= 2) {   // tile kernel, short-arm (2) / long-arm (3) instance, rb = geometry variant; nothing is written if an arm exceeds 4 / 13
		cfg.variant = rb;
		return cbca_tiles(scratch, vol_in, vol_out, D, H, W, direction, form == 2 ? 4 : 13, form == 2 ? CR_ARMS_LE4 : CR_ARMS_LE13, st, cfg);
	}
	cfg.rb = rb;
	rc = form == 1 ? cbca_strips(scratch, vol_in, vol_out, D, H, W, direction, CR_NOT_DIRECT, st, cfg)
	               : cbca_by_arms(scratch, vol_in, vol_out, D, H, W, direction, -1, st, cfg);
	if (rc) return rc;
	return cbca_if_overflow(x0c, x1c, scratch, vol_in, vol_out, D, H, W, direction, st);
}

int mc_transpose_cfg(const float *in, float *out, int64_t rows, int64_t cols, int64_t ldin, int64_t ldout, float scale_, int nt,
                     void *stream)
{
	MC_REQUIRE(in && out && in != out, "mc_transpose_cfg: bad pointers");
	MC_REQUIRE(rows >= 1 && cols >= 1 && ldin >= cols && ldout >= rows, "mc_transpose_cfg: bad dims");
	MC_REQUIRE(nt >= -1 && nt <= 1, "mc_transpose_cfg: bad nt");
	return transpose(in, out, rows, cols, ldin, ldout, scale_, as_stream(stream), nt);
}


int mc_sgm2_contract_violations(const float *in_hwd, int H, int W, int D, unsigned *count, void *stream)
{
	MC_REQUIRE(in_hwd && count, "mc_sgm2_contract_violations: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_sgm2_contract_violations: bad dims");
	return sgm_contract_violations(in_hwd, H, W, D, count, as_stream(stream));
}

size_t mc_sgm2_tmp_bytes(int H, int W, int D)
{
	(void)D;
	if (H < 1 || W < 1) return 0;
	return sgm_maps_bytes(H, W);
}

int mc_sgm2(const float *x0, const float *x1, const float *in_hwd, float *out_hwd, void *tmp, size_t tmp_bytes, int H, int W,
            int D, float pi1, float pi2, float tau_so, float alpha1, float sgm_q1, float sgm_q2, int direction, void *stream)
{
	MC_REQUIRE(x0 && x1 && in_hwd && out_hwd && tmp, "mc_sgm2: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_sgm2: bad dims");
	MC_REQUIRE(D <= MC_SGM_MAX_D, "mc_sgm2: D=%d exceeds %d", D, MC_SGM_MAX_D);
	MC_REQUIRE(direction == -1 || direction == 1, "mc_sgm2: direction must be -1 or 1");
	MC_REQUIRE(tmp_bytes >= sgm_maps_bytes(H, W), "mc_sgm2: tmp holds %zu bytes, needs %zu", tmp_bytes, sgm_maps_bytes(H, W));
	MC_REQUIRE(in_hwd != out_hwd, "mc_sgm2: input and output must differ");
	hipStream_t st = as_stream(stream);
	int rc = sgm_prep(x0, x1, tmp, H, W, tau_so, st);
	if (rc) return rc;
	const float *Cv[2] = {in_hwd, in_hwd};
	float *outv[2] = {out_hwd, out_hwd};
	const int dirv[2] = {direction, direction};
	return sgm_sweeps(Cv, outv, nullptr, nullptr, dirv, 1, H, W, D, D, tmp, pi1, pi2, alpha1, sgm_q1, sgm_q2, false, st);
}

int mc_dhw_to_hwd(const float *in, float *out, int D, int H, int W, void *stream)
{
	MC_REQUIRE(in && out && in != out, "mc_dhw_to_hwd: bad pointers");
	MC_REQUIRE(dims_ok(D, H, W), "mc_dhw_to_hwd: bad dims");
	return transpose(in, out, D, (int64_t)H * W, (int64_t)H * W, D, 1.0f, as_stream(stream));
}

int mc_hwd_to_dhw(const float *in, float *out, int D, int H, int W, float scale_, void *stream)
{
	MC_REQUIRE(in && out && in != out, "mc_hwd_to_dhw: bad pointers");
	MC_REQUIRE(dims_ok(D, H, W), "mc_hwd_to_dhw: bad dims");
	return transpose(in, out, (int64_t)H * W, D, D, (int64_t)H * W, scale_, as_stream(stream));
}

int mc_scale(const float *in, float *out, int64_t n, float s, void *stream)
{
	MC_REQUIRE((in && out) || n == 0, "mc_scale: null pointer");
	MC_REQUIRE(n >= 0, "mc_scale: negative size");
	return scale(in, out, n, s, as_stream(stream));
}

int mc_argmin(const float *vol, float *disp, int D, int H, int W, void *stream)
{
	MC_REQUIRE(vol && disp, "mc_argmin: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_argmin: bad dims");
	return argmin_dhw(vol, disp, D, H, W, 0, as_stream(stream));
}

int mc_spatial_argmin(const float *vol, float *out, int D, int H, int W, void *stream)
{
	MC_REQUIRE(vol && out, "mc_spatial_argmin: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_spatial_argmin: bad dims");
	return argmin_dhw(vol, out, D, H, W, 1, as_stream(stream));
}

int mc_outlier_detection(const float *d0, const float *d1, float *outlier, int H, int W, int disp_max, void *stream)
{
	MC_REQUIRE(d0 && d1 && outlier, "mc_outlier_detection: null pointer");
	MC_REQUIRE(dims_ok(1, H, W) && disp_max >= 0, "mc_outlier_detection: bad dims");
	return outlier_detection(d0, d1, outlier, H, W, disp_max, as_stream(stream));
}

int mc_interpolate_occlusion(const float *d0, const float *outlier, float *out, int H, int W, void *stream)
{
	MC_REQUIRE(d0 && outlier && out && out != d0, "mc_interpolate_occlusion: bad pointers");
	MC_REQUIRE(dims_ok(1, H, W), "mc_interpolate_occlusion: bad dims");
	return interpolate_occlusion(d0, outlier, out, H, W, as_stream(stream));
}

int mc_interpolate_mismatch(const float *d0, const float *outlier, float *out, int H, int W, void *stream)
{
	MC_REQUIRE(d0 && outlier && out && out != d0, "mc_interpolate_mismatch: bad pointers");
	MC_REQUIRE(dims_ok(1, H, W), "mc_interpolate_mismatch: bad dims");
	return interpolate_mismatch(d0, outlier, out, H, W, as_stream(stream));
}

int mc_subpixel_enchancement(const float *d0, const float *vol, float *out, int D, int H, int W, void *stream)
{
	MC_REQUIRE(d0 && vol && out, "mc_subpixel_enchancement: null pointer");
	MC_REQUIRE(dims_ok(D, H, W), "mc_subpixel_enchancement: bad dims");
	return subpixel(d0, vol, out, D, H, W, (int64_t)H * W, 1, as_stream(stream));
}

int mc_median2d(const float *img, float *out, int H, int W, int kernel_size, void *stream)
{
	MC_REQUIRE(img && out && img != out, "mc_median2d: bad pointers");
	MC_REQUIRE(dims_ok(1, H, W), "mc_median2d: bad dims");
	MC_REQUIRE(kernel_size % 2 == 1 && kernel_size >= 1 && kernel_size <= 11,
	           "mc_median2d: kernel_size must be odd and <= 11 (adcensus.cu:1601-1602)");
	return median2d(img, out, H, W, kernel_size, as_stream(stream));
}

int mc_mean2d(const float *img, const float *kernel, float *out, int H, int W, int ks, float alpha2, void *stream)
{
	MC_REQUIRE(img && kernel && out && img != out, "mc_mean2d: bad pointers");
	MC_REQUIRE(dims_ok(1, H, W), "mc_mean2d: bad dims");
	MC_REQUIRE(ks % 2 == 1 && ks >= 1, "mc_mean2d: kernel size must be odd (adcensus.cu:1269)");
	return mean2d(img, kernel, out, H, W, ks, alpha2, as_stream(stream));
}

int mc_gaussian_host(double sigma, float *host_kernel, int capacity)
{
	MC_REQUIRE(sigma > 0, "mc_gaussian_host: sigma must be > 0");
	const int ks = 2 * (int)ceil(sigma * 3) + 1;
	if (host_kernel) {  // plain host arithmetic: works without a device
		MC_REQUIRE(capacity >= ks * ks, "mc_gaussian_host: capacity %d < %d", capacity, ks * ks);
		gaussian_fill(sigma, host_kernel);
	}
	return ks;
}

int mc_normalize_forward(const float *in, float *norm, float *out, int N, int C, int H, int W, void *stream)
{
	MC_REQUIRE(in && out, "mc_normalize_forward: null pointer");
	MC_REQUIRE(N >= 1 && C >= 1 && dims_ok(1, H, W), "mc_normalize_forward: bad dims");
	return normalize_forward(in, norm, out, N, C, H, W, as_stream(stream));
}

size_t mc_predict_workspace_bytes(const mc_params *p, int C, int D, int H, int W)
{
	(void)C;
	if (!p || !dims_ok(D, H, W) || !(p->blur_sigma > 0)) return 0;
	return make_plan(p, D, H, W).total;
}

int mc_predict(const mc_params *p, const float *x0, const float *x1, const float *featL, const float *featR, int C,
               const float *rawL, const float *rawR, int D, int H, int W, void *workspace, size_t workspace_bytes,
               float *volL_out, float *volR_out, float *dispL0_out, float *dispR0_out, float *disp_out, void *stream)
{
	StageTimer tm;
	return predict_impl(p, x0, x1, featL, featR, C, rawL, rawR, D, H, W, workspace, workspace_bytes, volL_out, volR_out,
	                    dispL0_out, dispR0_out, disp_out, as_stream(stream), tm);
}

int mc_predict_timed(const mc_params *p, const float *x0, const float *x1, const float *featL, const float *featR, int C,
                     const float *rawL, const float *rawR, int D, int H, int W, void *workspace, size_t workspace_bytes,
                     float *disp_out, void *stream, float *stage_ms)
{
	StageTimer tm;
	tm.on = stage_ms != nullptr;
	tm.st = as_stream(stream);
	const int rc = predict_impl(p, x0, x1, featL, featR, C, rawL, rawR, D, H, W, workspace, workspace_bytes, nullptr, nullptr,
	                            nullptr, nullptr, disp_out, tm.st, tm);
	if (stage_ms) tm.collect(stage_ms);
	return rc;
}

}  // extern "C"
