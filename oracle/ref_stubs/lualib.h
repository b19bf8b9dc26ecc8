/* oracle/ref_stubs/lualib.h -- checker build only; see lua.h in this directory. */
#include "lua.h"
