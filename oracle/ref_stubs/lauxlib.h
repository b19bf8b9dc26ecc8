/* oracle/ref_stubs/lauxlib.h -- TEST INFRASTRUCTURE (checker build only); see lua.h here. */
#ifndef MCREF_LAUXLIB_H
#define MCREF_LAUXLIB_H
#include "lua.h"

typedef struct luaL_Reg {
	const char *name;
	lua_CFunction func;
} luaL_Reg;

/* luaL_error longjmps into Lua in the real library; here it throws into ref_shim.hip's dispatcher. */
static inline int luaL_error(lua_State *L, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(L->err, sizeof(L->err), fmt, ap);
	va_end(ap);
	throw mcref_lua_error{0};
	return 0;
}

static inline mcref_val *mcref_arg(lua_State *L, int idx, int tag, const char *what)
{
	if (idx < 1 || idx > L->narg || L->arg[idx - 1].tag != tag) luaL_error(L, "bad argument #%d (%s expected)", idx, what);
	return &L->arg[idx - 1];
}
static inline lua_Number luaL_checknumber(lua_State *L, int idx) { return mcref_arg(L, idx, MCREF_NUMBER, "number")->num; }
static inline lua_Integer luaL_checkinteger(lua_State *L, int idx)
{
	return (lua_Integer)mcref_arg(L, idx, MCREF_NUMBER, "number")->num;
}
static inline const char *luaL_checkstring(lua_State *L, int idx) { return mcref_arg(L, idx, MCREF_STRING, "string")->str; }

/* registration: remembered so that the shim can look functions up by name */
extern "C++" void mcref_register(const char *libname, const luaL_Reg *l);
static inline void luaL_openlib(lua_State *, const char *libname, const luaL_Reg *l, int) { mcref_register(libname, l); }
#endif
