/* TEST INFRASTRUCTURE ONLY — see ../ggml.h */
#include "../ggml.h"
