/* oracle/ref_stubs/luaT.h -- TEST INFRASTRUCTURE (checker build only); see lua.h here. */
#ifndef MCREF_LUAT_H
#define MCREF_LUAT_H
extern "C" {
#include "lauxlib.h"
}

static inline void *luaT_checkudata(lua_State *L, int idx, const char *tname)
{
	mcref_val *v = mcref_arg(L, idx, MCREF_UDATA, tname);
	if (!v->str || strcmp(v->str, tname) != 0) luaL_error(L, "bad argument #%d (%s expected, got %s)", idx, tname, v->str ? v->str : "?");
	return v->ud;
}
static inline void luaT_pushudata(lua_State *L, void *ud, const char *tname)
{
	if (L->nret < 8) {
		L->ret[L->nret].tag = MCREF_UDATA;
		L->ret[L->nret].ud = ud;
		L->ret[L->nret].str = tname;
		L->nret++;
	}
}
/* only SpatialLogSoftMax.cu (unused by the predict path) reads module fields */
static inline int luaT_getfieldcheckboolean(lua_State *L, int, const char *f) { return luaL_error(L, "field %s: not supported by the fake", f); }
static inline double luaT_getfieldchecknumber(lua_State *L, int, const char *f) { return luaL_error(L, "field %s: not supported by the fake", f); }
static inline void *luaT_getfieldcheckudata(lua_State *L, int, const char *f, const char *)
{
	luaL_error(L, "field %s: not supported by the fake", f);
	return 0;
}
#endif
