// The bevmsda_backward_* entry points of bevmsda_capi.hip (sampling backward: grad_loc / grad_attn gather, grad_value sort,
// first-generation kernels) as a translation unit of their own: `if constexpr (BWD)` in the launchers keeps each unit's
// kernels out of the other, so the two code objects can be inspected — and were, in rounds 5 and 6 — separately.
//
// Round 5 compiled this unit with -fno-slp-vectorize because of one sporadic wrong grad_loc_y; round 6 isolated the cause to
// ONE instruction form (packed fp32 add with an op_sel bit set: profiles/r6/r6_pk_forensics.txt) and removed it in the source
// (scalar_ops.h), so the unit is built like every other one again and tests/test_build_flags.py checks all five code objects.
#define BEVMSDA_PART_BACKWARD 1
#include "bevmsda_capi.hip"
