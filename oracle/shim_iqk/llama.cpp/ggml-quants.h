/* TEST INFRASTRUCTURE ONLY — see ggml-common.h in this directory. */
#include "ggml-impl.h"
