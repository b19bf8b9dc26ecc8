/* Generator of tests/golden/net_tiny_fast_{ascii,binary}.t7 -- a Torch7-serialised `{net_te, opt}` as main.lua:587-600 saves it
 * for the fast architecture, written WITHOUT the repo's own t7.py: this file restates torch7's writer side
 *   torch7/File.lua        File:writeObject  (type tags, object indices, "V 1" + class name, tables as n x (key, value))
 *   torch7/generic/Tensor.c / Storage.c  torch_Tensor_(write) / torch_Storage_(write)
 *   TH/THDiskFile.c        ASCII mode: the n values of ONE write call separated by ' ', '\n' after the call; "%d" "%ld" "%.9g" "%.17g"
 *                          binary mode: native little-endian int32 / int64 / float32 / float64
 * so that the reader (mc-cnn_amd/t7.py) meets a file it did not write.  The net: nn.Sequential { cudnn.SpatialConvolution(1 -> 4, 3x3,
 * pad 1), cudnn.ReLU(true), cudnn.SpatialConvolution(4 -> 4, 3x3, pad 1), nn.Normalize2, nn.StereoJoin1 } whose four parameter
 * tensors are views into ONE torch.CudaStorage (what net:getParameters() leaves behind): the storage is written once and
 * referenced by index afterwards.  Parameter k of the storage holds (float)(0.5 * sin(0.37 * k)).
 *   gcc -O1 -o /tmp/make_t7 tests/golden/make_t7_fixture.c -lm && /tmp/make_t7 tests/golden */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static FILE *f;
static int ascii, next_index;

static void w_int(int v) { if (ascii) fprintf(f, "%d\n", v); else fwrite(&v, 4, 1, f); }
static void w_long_n(const long *v, int n)
{
	if (ascii) { for (int i = 0; i < n; ++i) fprintf(f, i + 1 < n ? "%ld " : "%ld", v[i]); if (n > 0) fprintf(f, "\n"); }
	else fwrite(v, 8, n, f);
}
static void w_long(long v) { w_long_n(&v, 1); }
static void w_double(double v) { if (ascii) fprintf(f, "%.17g\n", v); else fwrite(&v, 8, 1, f); }
static void w_float_n(const float *v, long n)
{
	if (ascii) { for (long i = 0; i < n; ++i) fprintf(f, i + 1 < n ? "%.9g " : "%.9g", v[i]); if (n > 0) fprintf(f, "\n"); }
	else fwrite(v, 4, n, f);
}
static void w_chars(const char *s) { fwrite(s, 1, strlen(s), f); if (ascii && strlen(s) > 0) fprintf(f, "\n"); }

/* File:writeObject for the Lua value kinds */
static void o_number(double v) { w_int(1); w_double(v); }
static void o_string(const char *s) { w_int(2); w_int((int)strlen(s)); w_chars(s); }
static void o_bool(int b) { w_int(5); w_int(b ? 1 : 0); }
static int o_table_begin(int npairs) { w_int(3); int idx = ++next_index; w_int(idx); w_int(npairs); return idx; }
static int o_torch_begin(const char *cls) { w_int(4); int idx = ++next_index; w_int(idx); w_int(3); w_chars("V 1"); w_int((int)strlen(cls)); w_chars(cls); return idx; }
static void o_torch_ref(int idx) { w_int(4); w_int(idx); }

#define NPAR (4 * 1 * 9 + 4 + 4 * 4 * 9 + 4)
static float params[NPAR];
static int storage_index;

/* torch_Tensor_(write): nDimension, size[], stride[], storageOffset (1-based), then the storage as an object */
static void o_tensor(int nd, const long *size, long offset0)
{
	o_torch_begin("torch.CudaTensor");
	long stride[4];
	long s = 1;
	for (int i = nd - 1; i >= 0; --i) { stride[i] = s; s *= size[i]; }
	w_int(nd);
	w_long_n(size, nd);
	w_long_n(stride, nd);
	w_long(offset0 + 1);
	if (storage_index) o_torch_ref(storage_index);
	else {
		storage_index = o_torch_begin("torch.CudaStorage");
		w_long(NPAR);
		w_float_n(params, NPAR);
	}
}

static void conv(int nin, int nout, long woff, long boff)
{
	o_torch_begin("cudnn.SpatialConvolution");
	o_table_begin(11);   /* an object without a write method is the table of its fields */
	o_string("nInputPlane"); o_number(nin);
	o_string("nOutputPlane"); o_number(nout);
	o_string("kW"); o_number(3);
	o_string("kH"); o_number(3);
	o_string("dW"); o_number(1);
	o_string("dH"); o_number(1);
	o_string("padW"); o_number(1);
	o_string("padH"); o_number(1);
	o_string("groups"); o_number(1);
	const long ws[4] = {nout, nin, 3, 3}, bs[1] = {nout};
	o_string("weight"); o_tensor(4, ws, woff);
	o_string("bias"); o_tensor(1, bs, boff);
}

static void simple(const char *cls, int inplace)
{
	o_torch_begin(cls);
	if (inplace >= 0) { o_table_begin(1); o_string("inplace"); o_bool(inplace); }
	else o_table_begin(0);
}

static void write_file(const char *path, int as_ascii)
{
	f = fopen(path, "wb");
	if (!f) { perror(path); exit(1); }
	ascii = as_ascii; next_index = 0; storage_index = 0;
	o_table_begin(2);                       /* { net_te, opt } */
	o_number(1);
	o_torch_begin("nn.Sequential");
	o_table_begin(2);
	o_string("train"); o_bool(0);
	o_string("modules");
	o_table_begin(5);
	o_number(1); conv(1, 4, 0, 36);
	o_number(2); simple("cudnn.ReLU", 1);
	o_number(3); conv(4, 4, 40, 184);
	o_number(4); simple("nn.Normalize2", -1);
	o_number(5); simple("nn.StereoJoin1", -1);
	o_number(2);
	o_table_begin(4);                       /* opt: a few of main.lua's options */
	o_string("a"); o_string("train_all");
	o_string("l1"); o_number(2);
	o_string("fm"); o_number(4);
	o_string("debug"); o_bool(0);
	fclose(f);
}

int main(int argc, char **argv)
{
	const char *dir = argc > 1 ? argv[1] : ".";
	char path[1024];
	for (int k = 0; k < NPAR; ++k) params[k] = (float)(0.5 * sin(0.37 * k));
	snprintf(path, sizeof path, "%s/net_tiny_fast_ascii.t7", dir); write_file(path, 1);
	snprintf(path, sizeof path, "%s/net_tiny_fast_binary.t7", dir); write_file(path, 0);
	return 0;
}
