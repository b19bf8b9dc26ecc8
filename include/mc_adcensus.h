/*
 * mc_adcensus.h -- C ABI of libmcadcensus.so, the MI355X (gfx950) replacement
 * for the hot path of jzbontar/mc-cnn's libadcensus.so.
 *
 * The reference exposes this path as a Lua-C module: `require 'libadcensus'`
 * registers the table `adcensus` with the functions listed in funcs[]
 * (adcensus.cu:2061-2096), each `int f(lua_State*)` taking torch.CudaTensor
 * userdata.  A LuaJIT-FFI shim (mc-cnn_amd/lua/adcensus.lua)
 * rebuilds that table on top of the entry points below, so `main.lua -a
 * predict` keeps its call sites (main.lua:929-1082).  Each declaration cites
 * the reference binding it replaces.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to contiguous fp32 data (the reference
 *     reads raw THCudaTensor_data and assumes contiguity, adcensus.cu:97-111);
 *   - dims are passed explicitly (the reference reads them from tensor sizes);
 *   - `stream` is a hipStream_t (NULL = default stream); calls are
 *     asynchronous, never synchronise the device and never allocate, except
 *     mc_sgm2 / mc_predict which use caller-provided workspaces;
 *   - return 0 on success, a hipError_t (> 0) for a runtime failure or
 *     MC_EINVAL (< 0) for a bad argument; mc_last_error() returns a
 *     thread-local message (the reference raises luaL_error after
 *     cudaPeekAtLastError, adcensus.cu:31-36; the Lua shim turns rc != 0
 *     into error()).
 *   - volumes are (D,H,W) "DHW" unless a name says hwd = (H,W,D), D contiguous.
 *   - `direction` is -1 for the left-referenced volume (partner x-d) and +1
 *     for the right-referenced one (partner x+d), as in main.lua:986.
 */
#ifndef MC_ADCENSUS_H
#define MC_ADCENSUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */

#define MC_ABI_VERSION 8
#define MC_EINVAL (-22)
#define MC_SGM_MAX_D 512   /* reference: __shared__ float[400], adcensus.cu:574 */
#define MC_JOIN_MAX_C 128  /* reference: float L_cache[128], adcensus.cu:1460-1461 */

/* adcensus.version (adcensus.cu funcs[]) analogue. */
int mc_version(void);
const char *mc_last_error(void);

/* ---- cost volume ------------------------------------------------------ */

/* cutorch :fill(0/0) of the volumes, main.lua:933,939,946,966. */
int mc_fill_nan(float *p, int64_t n, void *stream);

/* adcensus.StereoJoin(input_L, input_R, output_L, output_R), adcensus.cu:1479-1498
 * (kernel 1455-1477).  feat*: (C,H,W), C <= 128; vol*: (D,H,W).  Writes only
 * voxels with x-d >= 0 (volL[d,y,x], volR[d,y,x-d]); the rest keeps the
 * caller's fill, as in the reference. */
int mc_stereo_join(const float *featL, const float *featR, float *volL, float *volR,
                   int C, int D, int H, int W, void *stream);

/* adcensus.ad(x0, x1, out, direction), adcensus.cu:95-114 (kernel 62-93). x: (H,W). */
int mc_ad(const float *x0, const float *x1, float *vol, int D, int H, int W, int direction, void *stream);

/* adcensus.census(x0, x1, out, direction), adcensus.cu:155-175 (kernel 117-153). x: (Cimg,H,W).  Computed from
 * per-pixel census signatures (the 81 comparison bits of a pixel's 9x9 window per channel do not depend on the
 * disparity: they are packed once per image into `scratch`, mc_census_scratch_bytes, and a voxel's cost is the
 * reference's count of out-of-bounds taps plus a popcount of differing bits) -- the same small integers as the
 * reference's 81 comparisons per voxel. */
size_t mc_census_scratch_bytes(int Cimg, int H, int W);
int mc_census_ws(const float *x0, const float *x1, float *vol, int Cimg, int D, int H, int W,
                 int direction, void *scratch, size_t scratch_bytes, void *stream);

/* Accurate architecture (arch slow), main.lua:958-983: for every pixel x and disparity d with x-d >= 0 the stack
 * net_te2 (main.lua:688-695: nn.SpatialConvolution1_fw layers = addmm + bias, SpatialConvolution1_fw.lua:11-31, ReLU
 * between, Sigmoid at the end) is applied to concat(featL[:,y,x], featR[:,y,x-d]); the result goes to volL[d,y,x] and
 * volR[d,y,x-d] (what the reference computes in two passes, one per direction).  Other voxels keep the caller's fill.
 *   weights[l]: DEVICE pointer to layer l's (out,in) row-major matrix, biases[l]: (out); the two arrays themselves are
 *   HOST arrays of n_layers device pointers; layer_out[l] = out width: 384 for all but the last, 1 for the last;
 *   layer 0 has in = 2*C.  fp32 on the matrix cores; parity by tolerance (the reference's GEMM order is cuBLAS's). */
size_t mc_fc_stack_workspace_bytes(int C, int n_layers, int H, int W);
int mc_fc_stack(const float *featL, const float *featR, int C, int H, int W, int D,
                const float *const *weights, const float *const *biases, const int *layer_out, int n_layers,
                float *volL, float *volR, void *workspace, size_t workspace_bytes, void *stream);

/* One layer of the feature net net_te: cudnn.SpatialConvolution(Cin, Cout, 3, 3, 1, 1, 1, 1) [+ cudnn.ReLU]
 * (main.lua:681-686 arch slow, 727-746 arch fast with the test-time padding of 1).  in (N,Cin,H,W), weight
 * (Cout,Cin,3,3), bias (Cout), out (N,Cout,H,W), Cout <= 128; fp32 on the matrix cores (implicit GEMM over the 9 taps).
 * `workspace` (mc_conv3x3_workspace_bytes) receives the weights re-laid per tap.  The reference's cuDNN picks its
 * algorithm at run time (cudnn.benchmark, main.lua:330): parity is by tolerance, not bit-exact. */
size_t mc_conv3x3_workspace_bytes(int Cin, int Cout);
int mc_conv3x3(const float *in, const float *weight, const float *bias, float *out, int N, int Cin, int Cout, int H, int W,
               int relu, void *workspace, size_t workspace_bytes, void *stream);

/* fix_border(net, vol, direction), main.lua:922-927: n = (window-1)/2 columns. */
int mc_fix_border(float *vol, int D, int H, int W, int n, int direction, void *stream);

/* ---- cross-based cost aggregation -------------------------------------- */

/* adcensus.cross(x0, out, L1, tau1), adcensus.cu:324-341 (kernel 280-322).
 * img (H,W) -> arms (4,H,W): exclusive arm ends for -x,+x,-y,+y. */
int mc_cross(const float *img, float *arms, int H, int W, int L1, float tau1, void *stream);

/* adcensus.cbca(x0c, x1c, vol_in, vol_out, direction), adcensus.cu:379-400
 * (kernel 343-377).  Reference accumulation order (yy outer, xx inner, one
 * fp32 accumulator, IEEE divide) is kept, so results are bit-identical. */
int mc_cbca(const float *x0c, const float *x1c, const float *vol_in, float *vol_out,
            int D, int H, int W, int direction, void *stream);

/* The same operator on the fast paths: the arm lengths of both images are packed into `scratch` (mc_cbca_scratch_bytes,
 * 4 bytes per pixel and image) together with what they look like -- an arm longer than 4 / 13 / 254 pixels present, the
 * share of pixels whose four arms are all minimal -- and a route word derived from that on the device.  Four kernels are
 * launched and exactly one runs (the others leave at their first instruction; the host reads nothing back):
 *   every arm <= 4 (arms from mc_cross with L1 <= 5) or <= 13 (L1 <= 14), real-scene statistics: the tile kernel
 *     (cbca_tile.hip) -- a block streams a strip of one disparity plane through an LDS ring, a lane owns a column x 4
 *     output rows and adds every run value once into four accumulators, items sorted by height per step;
 *   longer arms, or a pair on which nearly every support is the minimal 3x3 (textures: a bandwidth problem): the strip
 *     kernel -- one wave walks 256 staged columns of a plane top to bottom, minimal supports out of registers;
 *   an arm longer than 254 pixels (not representable in the packed form): mc_cbca's kernel.
 * Same accumulation order everywhere, bit-identical to mc_cbca.  (mc_predict knows L1 and launches fewer of them.) */
size_t mc_cbca_scratch_bytes(int H, int W);
int mc_cbca_ws(const float *x0c, const float *x1c, const float *vol_in, float *vol_out,
               int D, int H, int W, int direction, void *scratch, size_t scratch_bytes, void *stream);

/* ---- semiglobal matching ------------------------------------------------ */

/* Bytes of scratch mc_sgm2 needs in `tmp` (edge-class maps; the reference's
 * tmp (W,D) line state lives in registers here). */
size_t mc_sgm2_tmp_bytes(int H, int W, int D);

/* adcensus.sgm2(x0, x1, input, output, tmp, pi1, pi2, tau_so, alpha1, sgm_q1,
 * sgm_q2, direction), adcensus.cu:620-697 (kernels 535-618).  x0,x1: (H,W);
 * in_hwd/out_hwd: (H,W,D); the four directional costs are ADDED to out_hwd in
 * the reference's order (right, left, down, up), so the caller zeroes it first
 * (main.lua:1014).  D <= MC_SGM_MAX_D.  `tmp` must hold mc_sgm2_tmp_bytes();
 * pass tmp_bytes so that an undersized buffer is rejected, not overrun.
 * Contract: NaNs in a pixel's cost vector form a (possibly empty) tail in d
 * and d=0 is finite -- true for every volume the pipeline produces. */
int mc_sgm2(const float *x0, const float *x1, const float *in_hwd, float *out_hwd,
            void *tmp, size_t tmp_bytes, int H, int W, int D,
            float pi1, float pi2, float tau_so, float alpha1, float sgm_q1, float sgm_q2,
            int direction, void *stream);

/* Debug aid for mc_sgm2's contract (no reference counterpart): *count (a device word, zeroed by the call) receives the
 * number of pixels of in_hwd whose cost vector violates it -- d = 0 not finite, or a non-NaN value behind a NaN.  The
 * Python mirror calls it before every sgm2 when MC_CHECK_CONTRACTS=1 is set in the environment (and synchronises). */
int mc_sgm2_contract_violations(const float *in_hwd, int H, int W, int D, unsigned *count, void *stream);

/* vol:transpose(2,3):transpose(3,4):clone(), main.lua:1008: (D,H,W) -> (H,W,D). */
int mc_dhw_to_hwd(const float *in, float *out, int D, int H, int W, void *stream);
/* vol:copy(out:transpose(3,4):transpose(2,3)):div(4), main.lua:1019-1020:
 * (H,W,D) -> (D,H,W), every element multiplied by `scale` (0.25 = /4, exact). */
int mc_hwd_to_dhw(const float *in, float *out, int D, int H, int W, float scale, void *stream);
/* vol:copy(out):div(4), main.lua:1017 (any layout): out[i] = in[i] * scale. */
int mc_scale(const float *in, float *out, int64_t n, float scale, void *stream);

/* ---- disparity and post-processing ------------------------------------- */

/* _, d = torch.min(vol, 2); d:add(-1), main.lua:1049-1050 (cutorch).  0-based
 * argmin over d as float; first strict minimum, NaN never wins (the in-repo
 * convention of spatial_argmin, adcensus.cu:251-260). */
int mc_argmin(const float *vol, float *disp, int D, int H, int W, void *stream);

/* adcensus.spatial_argmin(input, output), adcensus.cu:264-278: 1-based. */
int mc_spatial_argmin(const float *vol, float *out, int D, int H, int W, void *stream);

/* adcensus.outlier_detection(d0, d1, outlier, disp_max), adcensus.cu:901-918 (kernel 878-899). */
int mc_outlier_detection(const float *d0, const float *d1, float *outlier, int H, int W,
                         int disp_max, void *stream);

/* adcensus.interpolate_occlusion(d0, outlier) -> new tensor, adcensus.cu:1107-1125
 * (kernel 1079-1105).  `out` is caller-provided here; the Lua shim allocates it. */
int mc_interpolate_occlusion(const float *d0, const float *outlier, float *out, int H, int W, void *stream);

/* adcensus.interpolate_mismatch(d0, outlier) -> new tensor, adcensus.cu:1060-1077 (kernel 1001-1058). */
int mc_interpolate_mismatch(const float *d0, const float *outlier, float *out, int H, int W, void *stream);

/* adcensus.subpixel_enchancement(d0, c2, disp_max) -> new tensor, adcensus.cu:1222-1239
 * (kernel 1205-1220).  vol: (D,H,W). */
int mc_subpixel_enchancement(const float *d0, const float *vol, float *out, int D, int H, int W, void *stream);

/* adcensus.median2d(img, kernel_size) -> new tensor, adcensus.cu:1596-1613 (kernel 1575-1594). k odd, <= 11. */
int mc_median2d(const float *img, float *out, int H, int W, int kernel_size, void *stream);

/* adcensus.mean2d(img, kernel, alpha2) -> new tensor, adcensus.cu:1263-1282 (kernel 1241-1261).
 * kernel: (ks,ks) device array, ks odd. */
int mc_mean2d(const float *img, const float *kernel, float *out, int H, int W, int ks, float alpha2, void *stream);

/* gaussian(sigma), main.lua:528-540, computed on the host in double and
 * rounded to float.  Returns ks = 2*ceil(3 sigma)+1; fills host_kernel
 * (ks*ks floats) when it is non-NULL and capacity >= ks*ks. */
int mc_gaussian_host(double sigma, float *host_kernel, int capacity);

/* adcensus.Normalize_forward(input, norm, output), adcensus.cu:1310-1333
 * (kernels 1284-1308).  x: (N,C,H,W); norm: (N,1,H,W). */
int mc_normalize_forward(const float *in, float *norm, float *out, int N, int C, int H, int W, void *stream);

/* ---- fused pipeline: stereo_predict, main.lua:929-1082 ------------------ */

typedef struct mc_params {
	int L1;            /* -L1        main.lua:87,132,222 ... */
	float tau1;        /* -tau1 */
	int cbca_i1;       /* -cbca_i1 */
	int cbca_i2;       /* -cbca_i2 */
	float pi1;         /* -pi1 */
	float pi2;         /* -pi2 (passed to sgm2 as P2, main.lua:1015) */
	int sgm_i;         /* -sgm_i */
	float sgm_q1;      /* -sgm_q1 */
	float sgm_q2;      /* -sgm_q2 */
	float alpha1;      /* -alpha1 */
	float tau_so;      /* -tau_so */
	double blur_sigma; /* -blur_sigma (double: gaussian() runs in Lua doubles) */
	float blur_t;      /* -blur_t */
	int lr_check;      /* 1 = kitti/kitti2015 branch main.lua:1054-1066, 0 = mb */
	int border_n;      /* fix_border n = (get_window_size(net)-1)/2, main.lua:923 */
	int median_k;      /* 5, main.lua:1073 */
	int sm_terminate;  /* -sm_terminate <stage>: MC_SM_* below, 0 = run everything (main.lua:25,988-1075) */
	int sm_skip;       /* -sm_skip <stage>: MC_SKIP_* below, 0 = skip nothing (main.lua:26,992-1077) */
	int left_only;     /* 1 = direction -1 only where the reference does so: dataset mb outside `-a predict`
	                      (mb_directions, main.lua:953-955); ignored when lr_check = 1 or a right-side output is
	                      requested.  0 = both directions, as `-a predict` computes them */
} mc_params;

/* -sm_terminate stages, in pipeline order (the stereo method stops being "active" after the named stage) */
enum { MC_SM_NONE = 0, MC_SM_CNN = 1, MC_SM_CBCA1 = 2, MC_SM_SGM = 3, MC_SM_CBCA2 = 4, MC_SM_OCCLUSION = 5,
       MC_SM_MISMATCH = 6, MC_SM_SUBPIXEL = 7, MC_SM_MEDIAN = 8, MC_SM_BILATERAL = 9 };
/* -sm_skip stages ('cbca' skips both aggregation blocks, 'occlusion' both interpolations, as in main.lua) */
enum { MC_SKIP_NONE = 0, MC_SKIP_CBCA = 1, MC_SKIP_SGM = 2, MC_SKIP_OCCLUSION = 3, MC_SKIP_SUBPIXEL = 4,
       MC_SKIP_MEDIAN = 5, MC_SKIP_BILATERAL = 6 };

/* Workspace bytes mc_predict needs for the given problem: six volumes, six maps, the SGM edge classes, arms and packed
 * arm lengths; for parameter sets that aggregate at least twice with L1 <= 14 also the plan area of the aggregation kernels
 * (mc_cbca_plan_bytes, ~3.6 bytes per voxel: the tile kernel's item order or, on textured pairs, the per-wave records of the
 * two-passes-per-launch kernel) per computed direction. */
size_t mc_predict_workspace_bytes(const mc_params *p, int C, int D, int H, int W);

/* stereo_predict(x_batch, id), main.lua:929-1082, from the cost-volume stage on.
 *   x0,x1    (H,W) normalised left/right images (x_batch[1], x_batch[2]);
 *   featL/R  (C,H,W) normalised features (arch fast: StereoJoin + fix_border,
 *            main.lua:945-949), or NULL to start from
 *   rawL/R   (D,H,W) raw left/right volumes (arch slow output of main.lua:958-983
 *            / ad / census); NOT modified.
 *   volL_out/volR_out  optional (D,H,W): what left.bin/right.bin hold (main.lua:1042-1047);
 *   dispL0_out/dispR0_out optional (H,W): argmin maps before post-processing;
 *   disp_out (H,W): the returned disparity (main.lua:1081).
 * Runs asynchronously on `stream` inside `workspace`. */
int mc_predict(const mc_params *p, const float *x0, const float *x1,
               const float *featL, const float *featR, int C,
               const float *rawL, const float *rawR, int D, int H, int W,
               void *workspace, size_t workspace_bytes,
               float *volL_out, float *volR_out, float *dispL0_out, float *dispR0_out,
               float *disp_out, void *stream);

/* Timing hook for bench.py: the same pipeline with HIP events recorded on `stream`
 * at the stage boundaries of this call.  stage_ms[MC_N_STAGES] receives the summed
 * duration (ms) of each stage; the call synchronises the stream to read them. */
enum {
	MC_STAGE_PREP = 0,   /* SGM edge-class maps + cross arms */
	MC_STAGE_JOIN = 1,   /* StereoJoin (+ NaN fill, fix_border) */
	MC_STAGE_CBCA = 2,   /* all cbca iterations */
	MC_STAGE_LAYOUT = 3, /* (D,H,W) <-> (H,W,D) transposes */
	MC_STAGE_SGM = 4,    /* the four direction sweeps, nothing else */
	MC_STAGE_ARGMIN = 5, /* stand-alone argmin + optional volume export */
	MC_STAGE_POST = 6,   /* LR check .. range-gated Gaussian */
	MC_N_STAGES = 8
};
int mc_predict_timed(const mc_params *p, const float *x0, const float *x1,
                     const float *featL, const float *featR, int C,
                     const float *rawL, const float *rawR, int D, int H, int W,
                     void *workspace, size_t workspace_bytes, float *disp_out, void *stream,
                     float *stage_ms);

/* ---- host side of libadcensus: submission / ground-truth image formats -------------------------------------------------
 * HOST pointers (the reference takes torch.FloatTensor here), no device, no stream.  16-bit greyscale PNG through a codec
 * written against the PNG specification on zlib (png++ / libpng, which the reference links, have no headers in the build
 * image); 8-bit greyscale files are accepted on read (v * 257). */

/* adcensus.readPNG16(img, fname), adcensus.cu:1670-1686: img[y * width + x] = val == 0 ? 0 : val / 256 (float).
 * *height, *width receive the file's size; img == NULL: size query only; capacity = floats img can hold. */
int mc_read_png16(const char *fname, float *img, int64_t capacity, int *height, int *width);

/* adcensus.writePNG16(img, height, width, fname), adcensus.cu:1688-1704: pixel = (uint16_t)(val < 1e-5 ? 0 : val * 256). */
int mc_write_png16(const float *img, int height, int width, const char *fname);

/* adcensus.writePFM(img, fname), adcensus.cu:1706-1721: "Pf", "width height", scale -0.003922, rows as stored, raw floats. */
int mc_write_pfm(const float *img, int height, int width, const char *fname);

/* adcensus.grey2jet(grey_img, col_img), adcensus.cu:2000-2053 ("CPU implementation": torch.DoubleTensor arguments; the debug images of
 * main.lua:503,1242,1260): val = 4 * grey[y][x] through the five linear pieces of the jet colour map into col[(c * height + y) * width + x],
 * c = 0 (red), 1, 2, in doubles.  Where the reference prints the value and asserts (val outside [-0.1, 4.1], NaN) this returns MC_EINVAL
 * with the pixel in mc_last_error; pixels before it are written, as in the reference. */
int mc_grey2jet(const double *grey, double *col, int height, int width);

/* ---- test / bench hooks (not part of the reference's surface) ---------------- */

/* mc_cbca_ws with the launch configuration forced instead of derived from the problem: cache policy `nt` (-1 = auto,
 * 0 = default policy, 1 = non-temporal volume accesses), planes [d0, d0+nd) only (nd = 0: all), kernel `form`:
 * 0 = what mc_cbca_ws does, 1 = strip kernel (`rb` = output rows per strip, 0 = auto), 2 / 3 = tile kernel, short-arm /
 * long-arm instance (`rb` = tile geometry variant, 0 = the product's; the launch writes NOTHING if an arm exceeds 4 / 13).
 * 4 / 5 = tile kernel (short- / long-arm instance) that also WRITES the pair's item order ("plan": what mc_predict keeps from
 * the first aggregation pass of a direction for the other 17) behind the packed lengths, 6 / 7 = tile kernel that READS it:
 * `scratch` must be 16-byte aligned and hold mc_cbca_scratch_bytes(H, W) + mc_cbca_plan_bytes(D, H, W) bytes, and a 6 / 7 call
 * must follow a 4 / 5 call with the same arms, D, H, W, direction and scratch.
 * 8 / 9 = the lean kernel mc_predict runs on textured pairs (route of the strip kernel, arms <= 13): 8 first lists the outputs
 * whose support is not the minimal 3 x 3 behind the packed lengths (same scratch size as 4 - 7), 9 reads that list; the strip
 * kernel takes over if the list is another problem's or did not fit.  There `rb` = rows per wave (2 / 4 / 8, else the product's),
 * `d0` = launch variant (bit 2 the listed outputs in a launch of their own, bit 4 one band of rows per XCD, bits 5 / 6
 * non-temporal loads / stores), `nd` = slots the list may hold (0 = all of its room).
 * 10 / 11 = TWO aggregation passes in one launch (cbca_lean2x_kernel: what mc_predict runs pairs of passes in on textured pairs);
 * vol_out = the volume after the second pass.  10 first writes the per-wave records of the listed outputs behind the packed lengths,
 * 11 reads them; the strip kernel takes both passes if the records are another problem's, a wave's entries do not fit its record or
 * the pair is not a texture (cost limit).  `scratch` additionally holds 512 bytes + one volume (the strip kernel's middle volume)
 * behind the plan area; `rb` = rows per wave (4 / 8 / 12, else the product's), `d0` > 0 = the cost limit in recomputed values per voxel.
 * Lets small-shape parity tests reach what the benchmarked sizes and parameter sets select.
 * mc_cbca_plan_bytes: the larger of the tile kernel's plan and the two-pass records (4 KB per wave of 8 rows x 252 columns). */
size_t mc_cbca_plan_bytes(int D, int H, int W);
int mc_cbca_ws_cfg(const float *x0c, const float *x1c, const float *vol_in, float *vol_out,
                   int D, int H, int W, int direction, void *scratch, size_t scratch_bytes,
                   int rb, int nt, int d0, int nd, int form, void *stream);

/* (H,W,D)<->(D,H,W) transpose with the cache policy forced (nt as above); scale multiplies every element. */
int mc_transpose_cfg(const float *in, float *out, int64_t rows, int64_t cols, int64_t ldin, int64_t ldout,
                     float scale, int nt, void *stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* MC_ADCENSUS_H */
