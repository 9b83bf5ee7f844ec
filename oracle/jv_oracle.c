/* jv_oracle.c -- CPU Jonker-Volgenant oracle, float and double instantiations.
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 * PARITY UNPINNED vs lapjv==1.3.14 (package absent); pinned vs scipy on certified-unique
 * instances.  Details: jv_oracle_impl.h. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "jv_oracle.h"

#define T float
#define SUFFIX f32
#define JV_T_IS_FLOAT 1
#include "jv_oracle_impl.h"
#undef T
#undef SUFFIX
#undef JV_T_IS_FLOAT

#define T double
#define SUFFIX f64
#include "jv_oracle_impl.h"
#undef T
#undef SUFFIX
