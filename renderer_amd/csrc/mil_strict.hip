// Strict numeric mode: GLSL operator order, IEEE divide/sqrt; build with -ffp-contract=off.
#define MIL_STRICT 1
#include "mil_device.hpp"
