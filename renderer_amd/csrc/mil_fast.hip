// Fast numeric mode: hoisted invariants, hardware rcp/rsq/sqrt, FMA contraction.
#define MIL_STRICT 0
#include "mil_device.hpp"
