/* TEST INFRASTRUCTURE ONLY — see ../hwloc.h */
#include "../hwloc.h"
