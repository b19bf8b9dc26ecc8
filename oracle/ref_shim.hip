/*
 * oracle/ref_shim.hip -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Compiles the reference's OWN source file, unmodified and where it lies
 * (#include "adcensus.cu" resolved through -I/root/reference), for gfx950 with hipcc,
 * against the fake Lua / TH / THC headers in oracle/ref_stubs/.  The result,
 * oracle/_ref/libadcensus_ref.so, lets tests drive the reference's binding functions
 * (`int f(lua_State*)`, registered in funcs[], adcensus.cu:2061-2096) BY NAME on the
 * MI355X: the reference's kernels, launch geometry and argument decoding are the
 * reference's code; only the Lua stack and the tensor structs are fakes.
 *
 * Only tests/ and bench.py's reference leg load this library (see oracle/ref_lib.py).
 * It is built by oracle/build_ref.py in the container that has /root/reference;
 * the .so is git-ignored and travels to the GPU box with the repo snapshot.
 * No reference source is copied into the repository.
 */
#include <map>
#include <string>

extern "C" {
#include "lua.h"
#include "lauxlib.h"
}
#include "luaT.h"
#include "THC.h"

static THCState g_state;
void *mcref_state_ptr() { return &g_state; }

static std::map<std::string, lua_CFunction> &registry()
{
	static std::map<std::string, lua_CFunction> r;
	return r;
}
void mcref_register(const char *libname, const luaL_Reg *l)
{
	for (; l && l->name; ++l) registry()[std::string(libname) + "." + l->name] = l->func;
}

/* ---- the reference, verbatim, from its own location ---- */
#include "adcensus.cu"

extern "C" {
#define MCREF_API __attribute__((visibility("default")))

/* Runs luaopen_libadcensus (adcensus.cu:2100-2105) once: registers `adcensus.*` (and nn.SpatialLogSoftMax_*). */
static void ensure_open()
{
	static bool opened = false;
	if (!opened) {
		lua_State L;
		memset(&L, 0, sizeof(L));
		luaopen_libadcensus(&L);
		opened = true;
	}
}

MCREF_API int mcref_nfuncs(void)
{
	ensure_open();
	return (int)registry().size();
}
MCREF_API const char *mcref_func_name(int i)
{
	ensure_open();
	for (auto &kv : registry())
		if (i-- == 0) return kv.first.c_str();
	return 0;
}

/* tensors: kind 0 = torch.CudaTensor (device float), 1 = FloatTensor, 2 = DoubleTensor, 3 = IntTensor, 4 = LongTensor */
MCREF_API void *mcref_tensor_new(int kind, void *data, int ndim, const long *sizes)
{
	if (ndim < 0 || ndim > MCREF_MAX_DIM) return 0;
	/* all five structs share one layout (MCREF_TENSOR); the element type only matters to the callee */
	THCudaTensor *t = (THCudaTensor *)calloc(1, sizeof(THCudaTensor));
	(void)kind;
	t->data = (float *)data;
	t->nDimension = ndim;
	for (int i = 0; i < ndim; i++) t->size[i] = sizes[i];
	t->owns = 0;
	return t;
}
MCREF_API void *mcref_tensor_data(void *t) { return ((THCudaTensor *)t)->data; }
MCREF_API int mcref_tensor_ndim(void *t) { return ((THCudaTensor *)t)->nDimension; }
MCREF_API long mcref_tensor_size(void *t, int i) { return ((THCudaTensor *)t)->size[i]; }
/* frees the struct and, for tensors the reference allocated itself (new_tensor_like), the device/host memory */
MCREF_API void mcref_tensor_free(void *tp, int device)
{
	THCudaTensor *t = (THCudaTensor *)tp;
	if (!t) return;
	if (t->owns && t->data) {
		if (device) (void)hipFree(t->data);
		else free(t->data);
	}
	free(t);
}
MCREF_API int mcref_copy_d2d(void *dst, const void *src, size_t bytes)
{
	return (int)hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice);
}
MCREF_API int mcref_device_sync(void) { return (int)hipDeviceSynchronize(); }

/* Call "adcensus.<name>" with args[0..nargs); returned values land in rets[0..*nrets).
 * rc 0 ok; 1 = Lua error raised by the reference (message in err); 2 = unknown function. */
MCREF_API int mcref_call(const char *name, const mcref_val *args, int nargs, mcref_val *rets, int *nrets, char *err, int errlen)
{
	ensure_open();
	auto it = registry().find(name);
	if (it == registry().end() || nargs > MCREF_MAX_ARGS) {
		if (err && errlen > 0) snprintf(err, errlen, "unknown function %s", name);
		return 2;
	}
	lua_State L;
	memset(&L, 0, sizeof(L));
	for (int i = 0; i < nargs; i++) L.arg[i] = args[i];
	L.narg = nargs;
	int rc = 0;
	try {
		(void)it->second(&L);
	} catch (const mcref_lua_error &) {
		rc = 1;
		if (err && errlen > 0) snprintf(err, errlen, "%s", L.err);
	}
	int n = L.nret < 8 ? L.nret : 8;
	if (rets && nrets) {
		for (int i = 0; i < n; i++) rets[i] = L.ret[i];
		*nrets = n;
	}
	return rc;
}
}
