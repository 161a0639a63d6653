/* see Rinternals.h in this directory: declaration-only stand-in, test infrastructure */
#ifndef LDW_TEST_R_API_MOCK_R_H
#define LDW_TEST_R_API_MOCK_R_H
#include <stddef.h>
#include <stdlib.h>
#endif
