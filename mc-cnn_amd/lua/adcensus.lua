--[[ adcensus.lua -- LuaJIT-FFI shim that rebuilds the reference's `adcensus` table
(luaopen_libadcensus, adcensus.cu:2061-2105) on top of libmcadcensus.so, the MI355X (gfx950)
C-ABI library of this repository (include/mc_adcensus.h).

Usage in main.lua (replaces `require 'libadcensus'`, main.lua:327):

    adcensus = require 'adcensus'          -- this file, with libmcadcensus.so on the library path

Every function keeps the reference's name, argument order and in-place / allocate-and-return
behaviour, so stereo_predict (main.lua:929-1082) runs unchanged.  Tensors are torch.CudaTensor
(hipified cutorch on ROCm); like the reference, contiguous 4-D float tensors are assumed
(adcensus.cu reads raw THCudaTensor_data); unlike it, contiguity is checked.  A non-zero return
code becomes error(mc_last_error()), where the reference calls luaL_error after
cudaPeekAtLastError (adcensus.cu:31-36).  All launches go to the NULL stream, which is the
stream the reference's kernels and cutorch's default stream use.

NOTE: LuaJIT/Torch7 are not available in the build image of this repository, so this file cannot be
executed there; tests/test_lua_shim.py checks its ffi.cdef block, ABI constant and mc_params layout
against include/mc_adcensus.h, and the identical C ABI is driven from Python ctypes in tests/.
]]
local ffi = require 'ffi'

ffi.cdef[[
int mc_version(void);
const char *mc_last_error(void);
int mc_fill_nan(float *p, int64_t n, void *stream);
int mc_stereo_join(const float *featL, const float *featR, float *volL, float *volR, int C, int D, int H, int W, void *stream);
int mc_ad(const float *x0, const float *x1, float *vol, int D, int H, int W, int direction, void *stream);
size_t mc_census_scratch_bytes(int Cimg, int H, int W);
int mc_census_ws(const float *x0, const float *x1, float *vol, int Cimg, int D, int H, int W, int direction,
                 void *scratch, size_t scratch_bytes, void *stream);
int mc_fix_border(float *vol, int D, int H, int W, int n, int direction, void *stream);
int mc_cross(const float *img, float *arms, int H, int W, int L1, float tau1, void *stream);
int mc_cbca(const float *x0c, const float *x1c, const float *vol_in, float *vol_out, int D, int H, int W, int direction, void *stream);
size_t mc_fc_stack_workspace_bytes(int C, int n_layers, int H, int W);
int mc_fc_stack(const float *featL, const float *featR, int C, int H, int W, int D,
                const float *const *weights, const float *const *biases, const int *layer_out, int n_layers,
                float *volL, float *volR, void *workspace, size_t workspace_bytes, void *stream);
size_t mc_cbca_scratch_bytes(int H, int W);
int mc_cbca_ws(const float *x0c, const float *x1c, const float *vol_in, float *vol_out, int D, int H, int W, int direction,
               void *scratch, size_t scratch_bytes, void *stream);
size_t mc_sgm2_tmp_bytes(int H, int W, int D);
int mc_sgm2(const float *x0, const float *x1, const float *in_hwd, float *out_hwd, void *tmp, size_t tmp_bytes,
            int H, int W, int D, float pi1, float pi2, float tau_so, float alpha1, float sgm_q1, float sgm_q2,
            int direction, void *stream);
int mc_sgm2_contract_violations(const float *in_hwd, int H, int W, int D, unsigned *count, void *stream);
int mc_spatial_argmin(const float *vol, float *out, int D, int H, int W, void *stream);
int mc_outlier_detection(const float *d0, const float *d1, float *outlier, int H, int W, int disp_max, void *stream);
int mc_interpolate_occlusion(const float *d0, const float *outlier, float *out, int H, int W, void *stream);
int mc_interpolate_mismatch(const float *d0, const float *outlier, float *out, int H, int W, void *stream);
int mc_subpixel_enchancement(const float *d0, const float *vol, float *out, int D, int H, int W, void *stream);
int mc_median2d(const float *img, float *out, int H, int W, int kernel_size, void *stream);
int mc_mean2d(const float *img, const float *kernel, float *out, int H, int W, int ks, float alpha2, void *stream);
int mc_normalize_forward(const float *in, float *norm, float *out, int N, int C, int H, int W, void *stream);
typedef struct mc_params {
	int L1;
	float tau1;
	int cbca_i1;
	int cbca_i2;
	float pi1;
	float pi2;
	int sgm_i;
	float sgm_q1;
	float sgm_q2;
	float alpha1;
	float tau_so;
	double blur_sigma;
	float blur_t;
	int lr_check;
	int border_n;
	int median_k;
	int sm_terminate;
	int sm_skip;
	int left_only;
} mc_params;
size_t mc_predict_workspace_bytes(const mc_params *p, int C, int D, int H, int W);
int mc_predict(const mc_params *p, const float *x0, const float *x1,
               const float *featL, const float *featR, int C,
               const float *rawL, const float *rawR, int D, int H, int W,
               void *workspace, size_t workspace_bytes,
               float *volL_out, float *volR_out, float *dispL0_out, float *dispR0_out,
               float *disp_out, void *stream);
int mc_read_png16(const char *fname, float *img, int64_t capacity, int *height, int *width);
int mc_write_png16(const float *img, int height, int width, const char *fname);
int mc_write_pfm(const float *img, int height, int width, const char *fname);
int mc_grey2jet(const double *grey, double *col, int height, int width);
]]

local lib = ffi.load('mcadcensus')
local MC_ABI_VERSION = 8   -- include/mc_adcensus.h; tests/test_lua_shim.py checks this constant and every prototype above
assert(lib.mc_version() == MC_ABI_VERSION, ('libmcadcensus ABI version %d, this shim is written for %d'):format(
   lib.mc_version(), MC_ABI_VERSION))

local function check(rc, what)
   if rc ~= 0 then
      error(('%s: %s (rc=%d)'):format(what, ffi.string(lib.mc_last_error()), rc), 3)
   end
end

-- luaT_checkudata(L, i, "torch.CudaTensor") + the contiguity the reference silently assumes
local function ptr(t, what)
   if torch.typename(t) ~= 'torch.CudaTensor' then
      error(('%s: torch.CudaTensor expected, got %s'):format(what, torch.typename(t) or type(t)), 3)
   end
   if not t:isContiguous() then
      error(what .. ': contiguous tensor expected', 3)
   end
   return t:data()   -- float* device pointer (cutorch FFI)
end

local function like(x)   -- new_tensor_like, adcensus.cu:40-45
   return torch.CudaTensor():resizeAs(x)
end

local adcensus = {}

function adcensus.ad(x0, x1, out, direction)                 -- adcensus.cu:95-114
   check(lib.mc_ad(ptr(x0, 'ad'), ptr(x1, 'ad'), ptr(out, 'ad'), out:size(2), out:size(3), out:size(4), direction, nil), 'ad')
end

-- adcensus.census, adcensus.cu:155-175, on the signature kernels (mc_census_ws); scratch kept like cbca's below
local census_scratch = torch.CudaTensor()
function adcensus.census(x0, x1, out, direction)
   local C, D, H, W = x0:size(2), out:size(2), out:size(3), out:size(4)
   local need = tonumber(lib.mc_census_scratch_bytes(C, H, W))
   if census_scratch:nElement() * 4 < need then census_scratch:resize(math.ceil(need / 4)) end
   check(lib.mc_census_ws(ptr(x0, 'census'), ptr(x1, 'census'), ptr(out, 'census'), C, D, H, W, direction,
                          census_scratch:data(), census_scratch:nElement() * 4, nil), 'census')
end

function adcensus.StereoJoin(input_L, input_R, output_L, output_R)   -- adcensus.cu:1479-1498
   check(lib.mc_stereo_join(ptr(input_L, 'StereoJoin'), ptr(input_R, 'StereoJoin'), ptr(output_L, 'StereoJoin'),
                            ptr(output_R, 'StereoJoin'), input_L:size(2), output_L:size(2), output_L:size(3),
                            output_L:size(4), nil), 'StereoJoin')
end

-- NEW entry (no counterpart in libadcensus): the per-disparity loop of arch slow, main.lua:958-983, in one call.
--   net_te2: the nn.Sequential of SpatialConvolution1_fw / ReLU / Sigmoid (main.lua:688-695); output: net_te.output (2,C,H,W);
--   volL, volR: (1,disp_max,H,W) CudaTensors pre-filled with 0/0 as main.lua:968 does; both directions come out of one pass,
--   so stereo_predict calls this once before its direction loop and skips lines 958-983 (fix_border stays, main.lua:984).
local fc_ws = torch.CudaTensor()
function adcensus.fc_stack(net_te2, output, volL, volR)
   local layers = {}
   for _, m in ipairs(net_te2.modules) do
      if torch.typename(m) == 'nn.SpatialConvolution1_fw' then layers[#layers + 1] = m end
   end
   local n = #layers
   local w = ffi.new('const float *[?]', n)
   local b = ffi.new('const float *[?]', n)
   local width = ffi.new('int[?]', n)
   for i, m in ipairs(layers) do
      w[i - 1], b[i - 1], width[i - 1] = m.weight:data(), m.bias:data(), m.weight:size(1)
   end
   local C, H, W, D = output:size(2), output:size(3), output:size(4), volL:size(2)
   local need = tonumber(lib.mc_fc_stack_workspace_bytes(C, n, H, W))
   if fc_ws:nElement() * 4 < need then fc_ws:resize(math.ceil(need / 4)) end
   check(lib.mc_fc_stack(output[1]:data(), output[2]:data(), C, H, W, D, w, b, width, n, ptr(volL, 'fc_stack'),
                         ptr(volR, 'fc_stack'), fc_ws:data(), fc_ws:nElement() * 4, nil), 'fc_stack')
end

function adcensus.cross(x0, out, L1, tau1)                   -- adcensus.cu:324-341
   check(lib.mc_cross(ptr(x0, 'cross'), ptr(out, 'cross'), out:size(3), out:size(4), L1, tau1, nil), 'cross')
end

-- adcensus.cbca, adcensus.cu:379-400, through mc_cbca_ws: both images' arm lengths are packed into the scratch below, the
-- library derives on the device which kernel the pair's arms call for (tile kernel short / long arms, strip kernel, one
-- thread per voxel) and runs exactly that one -- nothing is kept between calls, nothing is read back by the host.  The packed
-- arm lengths live in a scratch tensor that grows with the largest image seen.
local cbca_scratch = torch.CudaTensor()
function adcensus.cbca(x0c, x1c, vol_in, vol_out, direction)
   local D, H, W = vol_out:size(2), vol_out:size(3), vol_out:size(4)
   local need = tonumber(lib.mc_cbca_scratch_bytes(H, W))
   if cbca_scratch:nElement() * 4 < need then cbca_scratch:resize(math.ceil(need / 4)) end
   check(lib.mc_cbca_ws(ptr(x0c, 'cbca'), ptr(x1c, 'cbca'), ptr(vol_in, 'cbca'), ptr(vol_out, 'cbca'),
                        D, H, W, direction, cbca_scratch:data(), cbca_scratch:nElement() * 4, nil), 'cbca')
end

-- adcensus.sgm2, adcensus.cu:620-697.  `tmp` is the reference's (W,D) line-state tensor; this
-- library keeps line state in registers and needs a differently sized scratch (edge-class maps),
-- so `tmp` is resized to that many floats (a CudaTensor the caller already owns and reuses).
function adcensus.sgm2(x0, x1, input, output, tmp, pi1, pi2, tau_so, alpha1, sgm_q1, sgm_q2, direction)
   local H, W, D = input:size(2), input:size(3), input:size(4)
   local need = tonumber(lib.mc_sgm2_tmp_bytes(H, W, D))
   if tmp:nElement() * 4 < need then tmp:resize(math.ceil(need / 4)) end
   check(lib.mc_sgm2(ptr(x0, 'sgm2'), ptr(x1, 'sgm2'), ptr(input, 'sgm2'), ptr(output, 'sgm2'), ptr(tmp, 'sgm2'),
                     tmp:nElement() * 4, H, W, D, pi1, pi2, tau_so, alpha1, sgm_q1, sgm_q2, direction, nil), 'sgm2')
end

function adcensus.spatial_argmin(input, output)              -- adcensus.cu:264-278
   check(lib.mc_spatial_argmin(ptr(input, 'spatial_argmin'), ptr(output, 'spatial_argmin'),
                               input:size(2), input:size(3), input:size(4), nil), 'spatial_argmin')
end

function adcensus.outlier_detection(d0, d1, outlier, disp_max)   -- adcensus.cu:901-918
   check(lib.mc_outlier_detection(ptr(d0, 'outlier_detection'), ptr(d1, 'outlier_detection'),
                                  ptr(outlier, 'outlier_detection'), d0:size(3), d0:size(4), disp_max, nil),
         'outlier_detection')
end

function adcensus.interpolate_occlusion(d0, outlier)         -- adcensus.cu:1107-1125
   local out = like(d0)
   check(lib.mc_interpolate_occlusion(ptr(d0, 'interpolate_occlusion'), ptr(outlier, 'interpolate_occlusion'),
                                      out:data(), d0:size(3), d0:size(4), nil), 'interpolate_occlusion')
   return out
end

function adcensus.interpolate_mismatch(d0, outlier)          -- adcensus.cu:1060-1077
   local out = like(d0)
   check(lib.mc_interpolate_mismatch(ptr(d0, 'interpolate_mismatch'), ptr(outlier, 'interpolate_mismatch'),
                                     out:data(), d0:size(3), d0:size(4), nil), 'interpolate_mismatch')
   return out
end

function adcensus.subpixel_enchancement(d0, c2, disp_max)    -- adcensus.cu:1222-1239
   local out = like(d0)
   check(lib.mc_subpixel_enchancement(ptr(d0, 'subpixel_enchancement'), ptr(c2, 'subpixel_enchancement'),
                                      out:data(), disp_max, d0:size(3), d0:size(4), nil), 'subpixel_enchancement')
   return out
end

function adcensus.median2d(img, kernel_size)                 -- adcensus.cu:1596-1613
   local out = like(img)
   check(lib.mc_median2d(ptr(img, 'median2d'), out:data(), img:size(3), img:size(4), kernel_size, nil), 'median2d')
   return out
end

function adcensus.mean2d(img, kernel, alpha2)                -- adcensus.cu:1263-1282
   local out = like(img)
   check(lib.mc_mean2d(ptr(img, 'mean2d'), ptr(kernel, 'mean2d'), out:data(), img:size(3), img:size(4),
                       kernel:size(1), alpha2, nil), 'mean2d')
   return out
end

function adcensus.Normalize_forward(input, norm, output)     -- adcensus.cu:1310-1333
   check(lib.mc_normalize_forward(ptr(input, 'Normalize_forward'), ptr(norm, 'Normalize_forward'),
                                  ptr(output, 'Normalize_forward'), input:size(1), input:size(2), input:size(3),
                                  input:size(4), nil), 'Normalize_forward')
end

-- host side of libadcensus (torch.FloatTensor arguments, as in the reference): KITTI ground truth / submission files
local function hostptr(t, what)
   if torch.typename(t) ~= 'torch.FloatTensor' then
      error(('%s: torch.FloatTensor expected, got %s'):format(what, torch.typename(t) or type(t)), 3)
   end
   if not t:isContiguous() then error(what .. ': contiguous tensor expected', 3) end
   return t:data()
end

function adcensus.readPNG16(img, fname)                      -- adcensus.cu:1670-1686
   local h, w = ffi.new('int[1]'), ffi.new('int[1]')
   check(lib.mc_read_png16(fname, hostptr(img, 'readPNG16'), img:nElement(), h, w), 'readPNG16')
end

function adcensus.writePNG16(img, height, width, fname)      -- adcensus.cu:1688-1704
   if img:nElement() < height * width then                  -- (the reference reads past the tensor here; sizes are checked in this package)
      error(('writePNG16: the tensor holds %d elements, %d x %d asked for'):format(img:nElement(), height, width), 2)
   end
   check(lib.mc_write_png16(hostptr(img, 'writePNG16'), height, width, fname), 'writePNG16')
end

function adcensus.writePFM(img, fname)                       -- adcensus.cu:1706-1721
   if img:nDimension() ~= 2 then error('writePFM: 2-D tensor expected, got ' .. img:nDimension() .. ' dimensions', 2) end
   check(lib.mc_write_pfm(hostptr(img, 'writePFM'), img:size(1), img:size(2), fname), 'writePFM')
end

function adcensus.grey2jet(grey_img, col_img)                -- adcensus.cu:2000-2053 (torch.DoubleTensor arguments; main.lua:503,1242,1260)
   for _, t in ipairs({grey_img, col_img}) do
      if torch.typename(t) ~= 'torch.DoubleTensor' then error('grey2jet: torch.DoubleTensor expected, got ' .. (torch.typename(t) or type(t)), 2) end
      if not t:isContiguous() then error('grey2jet: contiguous tensor expected', 2) end
   end
   assert(grey_img:nDimension() == 2)
   if 3 * grey_img:nElement() ~= col_img:nElement() then error('Size mismatch', 2) end
   check(lib.mc_grey2jet(grey_img:data(), col_img:data(), grey_img:size(1), grey_img:size(2)), 'grey2jet')
end

-- NEW entry (no counterpart in libadcensus): the whole of stereo_predict (main.lua:929-1082) from the cost-volume stage
-- on in ONE call (mc_predict): NaN fill, StereoJoin, fix_border, cross, cbca, sgm2, both arg-mins, the LR check,
-- interpolation, sub-pixel, median and the range-gated Gaussian -- what main.lua does in ~25 adcensus.* / cutorch calls.
--   o        the option table of main.lua (opt.L1, opt.tau1, opt.cbca_i1, ... opt.sm_terminate, opt.sm_skip)
--   dataset  'kitti' | 'kitti2015' | 'mb'  (selects the LR-check branch, main.lua:1054)
--   x_batch  (2,1,H,W) CudaTensor, the normalised pair
--   input    arch fast: net_te.output (2,C,H,W) after forward_free; any other arch: the raw volumes (2,disp_max,H,W)
--            (adcensus.ad / census / fc_stack output, NaN outside the valid range)
--   border_n (get_window_size(net_te) - 1) / 2 for arch fast (fix_border, main.lua:922-927); ignored otherwise
--   both     true for `-a predict` (both directions; returns the left.bin / right.bin volumes), false otherwise
-- Returns disp (1,1,H,W) and, when `both`, volL, volR (1,disp_max,H,W).
local SM_TERMINATE = {[''] = 0, cnn = 1, cbca1 = 2, sgm = 3, cbca2 = 4, occlusion = 5, mismatch = 6,
                      subpixel_enchancement = 7, median = 8, bilateral = 9}
local SM_SKIP = {[''] = 0, cbca = 1, sgm = 2, occlusion = 3, subpixel_enchancement = 4, median = 5, bilateral = 6}
local predict_ws = torch.CudaTensor()
function adcensus.predict(o, dataset, arch, x_batch, input, disp_max, border_n, both)
   local p = ffi.new('mc_params')
   p.L1, p.tau1, p.cbca_i1, p.cbca_i2 = o.L1, o.tau1, o.cbca_i1, o.cbca_i2
   p.pi1, p.pi2, p.sgm_i, p.sgm_q1, p.sgm_q2 = o.pi1, o.pi2, o.sgm_i, o.sgm_q1, o.sgm_q2
   p.alpha1, p.tau_so, p.blur_sigma, p.blur_t = o.alpha1, o.tau_so, o.blur_sigma, o.blur_t
   p.lr_check = (dataset == 'kitti' or dataset == 'kitti2015') and 1 or 0
   p.border_n = arch == 'fast' and border_n or 0
   p.median_k = 5
   p.sm_terminate = SM_TERMINATE[o.sm_terminate or ''] or error('unknown -sm_terminate ' .. tostring(o.sm_terminate))
   p.sm_skip = SM_SKIP[o.sm_skip or ''] or error('unknown -sm_skip ' .. tostring(o.sm_skip))
   p.left_only = (not both and dataset == 'mb') and 1 or 0   -- mb_directions, main.lua:953-955: dataset mb outside `-a predict` only
   local H, W = x_batch:size(3), x_batch:size(4)
   local C = arch == 'fast' and input:size(2) or 0
   local need = tonumber(lib.mc_predict_workspace_bytes(p, C, disp_max, H, W))
   if need == 0 then error('mc_predict_workspace_bytes: bad arguments', 2) end
   if predict_ws:nElement() * 4 < need + 256 then predict_ws:resize(math.ceil((need + 256) / 4)) end
   local ws = ffi.cast('char *', predict_ws:data())
   ws = ws + (256 - tonumber(ffi.cast('uintptr_t', ws)) % 256) % 256      -- mc_predict wants a 256-byte aligned workspace
   local disp = torch.CudaTensor(1, 1, H, W)
   local volL, volR, pl, pr = nil, nil, nil, nil
   if both then
      volL, volR = torch.CudaTensor(1, disp_max, H, W), torch.CudaTensor(1, disp_max, H, W)
      pl, pr = volL:data(), volR:data()
   end
   local x0, x1 = x_batch[1]:data(), x_batch[2]:data()
   ptr(x_batch, 'predict'); ptr(input, 'predict')
   local fl, fr, rl, rr = nil, nil, nil, nil
   if arch == 'fast' then fl, fr = input[1]:data(), input[2]:data() else rl, rr = input[1]:data(), input[2]:data() end
   check(lib.mc_predict(p, x0, x1, fl, fr, C, rl, rr, disp_max, H, W, ws, need, pl, pr, nil, nil, disp:data(), nil), 'predict')
   return disp, volL, volR
end

function adcensus.version()                                  -- adcensus.cu:2055-2059
   print(('libmcadcensus (MI355X) ABI version %d'):format(lib.mc_version()))
end

return adcensus
