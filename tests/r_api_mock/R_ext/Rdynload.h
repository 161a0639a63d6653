/* see ../Rinternals.h: declaration-only stand-in, test infrastructure */
#ifndef LDW_TEST_R_API_MOCK_RDYNLOAD_H
#define LDW_TEST_R_API_MOCK_RDYNLOAD_H
#include "../Rinternals.h"
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *const croutines, const R_CallMethodDef *const callRoutines, const R_FortranMethodDef *const fortranRoutines,
                       const R_ExternalMethodDef *const externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
Rboolean R_forceSymbols(DllInfo *info, Rboolean value);
#endif
