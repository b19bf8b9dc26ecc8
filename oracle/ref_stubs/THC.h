/*
 * oracle/ref_stubs/THC.h -- TEST INFRASTRUCTURE (checker build only).
 *
 * Just enough of Torch7's TH / THC tensor C API for adcensus.cu to compile with hipcc:
 * contiguous tensors described by a data pointer and sizes.  The CUDA runtime names the
 * reference uses are mapped onto HIP.  Not Torch code; see lua.h in this directory.
 */
#ifndef MCREF_THC_H
#define MCREF_THC_H
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#define cudaError_t hipError_t
#define cudaError hipError_t
#define cudaSuccess hipSuccess
#define cudaPeekAtLastError hipPeekAtLastError
#define cudaGetLastError hipGetLastError
#define cudaGetErrorString hipGetErrorString

typedef struct THCState { int dummy; } THCState;

#define MCREF_MAX_DIM 8
#define MCREF_TENSOR(NAME, T)                                                                     \
	typedef struct NAME {                                                                         \
		T *data;                                                                                  \
		long size[MCREF_MAX_DIM];                                                                 \
		int nDimension;                                                                           \
		int owns; /* 1 = data allocated by the fake (hipMalloc / malloc) */                       \
	} NAME;

MCREF_TENSOR(THCudaTensor, float)
MCREF_TENSOR(THFloatTensor, float)
MCREF_TENSOR(THDoubleTensor, double)
MCREF_TENSOR(THIntTensor, int)
MCREF_TENSOR(THLongTensor, long)

template <typename TT> static inline long mcref_nelem(const TT *t)
{
	if (t->nDimension == 0) return 0;
	long n = 1;
	for (int i = 0; i < t->nDimension; i++) n *= t->size[i];
	return n;
}

/* ---- THCudaTensor (device) ---- */
static inline float *THCudaTensor_data(THCState *, THCudaTensor *t) { return t->data; }
static inline long THCudaTensor_size(THCState *, THCudaTensor *t, int dim) { return t->size[dim]; }
static inline long THCudaTensor_nElement(THCState *, THCudaTensor *t) { return mcref_nelem(t); }
static inline THCudaTensor *THCudaTensor_new(THCState *) { return (THCudaTensor *)calloc(1, sizeof(THCudaTensor)); }
static inline void THCudaTensor_resizeAs(THCState *, THCudaTensor *y, THCudaTensor *x)
{
	if (y->owns && y->data) (void)hipFree(y->data);
	y->nDimension = x->nDimension;
	memcpy(y->size, x->size, sizeof(y->size));
	void *p = 0;
	(void)hipMalloc(&p, sizeof(float) * (size_t)(mcref_nelem(x) > 0 ? mcref_nelem(x) : 1));
	y->data = (float *)p;
	y->owns = 1;
}
static inline THCudaTensor *THCudaTensor_newContiguous(THCState *, THCudaTensor *t) { return t; }  /* always contiguous here */
static inline void THCudaTensor_free(THCState *, THCudaTensor *) {}

/* ---- host tensors ---- */
#define MCREF_HOST_API(NAME, T)                                                                   \
	static inline T *NAME##_data(NAME *t) { return t->data; }                                     \
	static inline long NAME##_size(NAME *t, int dim) { return t->size[dim]; }                     \
	static inline long NAME##_nElement(NAME *t) { return mcref_nelem(t); }                        \
	static inline NAME *NAME##_new(void) { return (NAME *)calloc(1, sizeof(NAME)); }              \
	static inline NAME *NAME##_newWithSize1d(long n)                                              \
	{                                                                                             \
		NAME *t = (NAME *)calloc(1, sizeof(NAME));                                                \
		t->nDimension = 1;                                                                        \
		t->size[0] = n;                                                                           \
		t->data = (T *)malloc(sizeof(T) * (size_t)(n > 0 ? n : 1));                               \
		t->owns = 1;                                                                              \
		return t;                                                                                 \
	}                                                                                             \
	template <typename XT> static inline void NAME##_resizeAs(NAME *y, XT *x)                     \
	{                                                                                             \
		if (y->owns && y->data) free(y->data);                                                    \
		y->nDimension = x->nDimension;                                                            \
		memcpy(y->size, x->size, sizeof(y->size));                                                \
		y->data = (T *)malloc(sizeof(T) * (size_t)(mcref_nelem(x) > 0 ? mcref_nelem(x) : 1));     \
		y->owns = 1;                                                                              \
	}                                                                                             \
	static inline void NAME##_zero(NAME *t) { memset(t->data, 0, sizeof(T) * (size_t)mcref_nelem(t)); }

MCREF_HOST_API(THFloatTensor, float)
MCREF_HOST_API(THDoubleTensor, double)
MCREF_HOST_API(THIntTensor, int)
MCREF_HOST_API(THLongTensor, long)

#define THError(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); abort(); } while (0)
#define THArgCheck(cond, argn, msg) do { if (!(cond)) { fprintf(stderr, "bad argument %d: %s\n", argn, msg); abort(); } } while (0)
#endif
