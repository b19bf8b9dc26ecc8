/*
 * oracle/ref_stubs/lua.h -- TEST INFRASTRUCTURE (checker build only).
 *
 * A fake of the handful of Lua C-API entry points that /root/reference/adcensus.cu
 * uses, so that the reference's OWN, UNMODIFIED source file can be compiled where it
 * lies (hipcc, gfx950) and its binding functions (`int f(lua_State*)`) can be driven
 * from Python through oracle/ref_shim.hip.  Nothing here comes from Lua or Torch7;
 * it only has to satisfy the call sites in adcensus.cu / SpatialLogSoftMax.cu.
 *
 * The "Lua stack" is an argument array filled by the caller (positive indices, as the
 * reference uses them) plus a list of returned values (luaT_pushudata / lua_pushinteger).
 */
#ifndef MCREF_LUA_H
#define MCREF_LUA_H
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

enum { MCREF_NIL = 0, MCREF_NUMBER = 1, MCREF_STRING = 2, MCREF_UDATA = 3 };

typedef struct mcref_val {
	int tag;
	double num;
	const char *str;   /* MCREF_STRING: the string; MCREF_UDATA: the torch type name */
	void *ud;
} mcref_val;

#define MCREF_MAX_ARGS 32
typedef struct lua_State {
	mcref_val arg[MCREF_MAX_ARGS];
	int narg;
	mcref_val ret[8];
	int nret;
	char err[512];
} lua_State;

typedef int (*lua_CFunction)(lua_State *L);
typedef double lua_Number;
typedef long lua_Integer;

struct mcref_lua_error { int dummy; };

/* getCutorchState() (adcensus.cu:21-29) fetches cutorch.getState(): the fake has one global state. */
static inline void lua_getglobal(lua_State *, const char *) {}
static inline void lua_getfield(lua_State *, int, const char *) {}
static inline void lua_call(lua_State *, int, int) {}
static inline void lua_pop(lua_State *, int) {}
extern "C++" void *mcref_state_ptr();
static inline void *lua_touserdata(lua_State *, int) { return mcref_state_ptr(); }
static inline void lua_pushinteger(lua_State *L, lua_Integer v)
{
	if (L->nret < 8) {
		L->ret[L->nret].tag = MCREF_NUMBER;
		L->ret[L->nret].num = (double)v;
		L->nret++;
	}
}
#endif
