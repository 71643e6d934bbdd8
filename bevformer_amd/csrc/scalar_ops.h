// Scalar fp32 VALU operations the optimiser cannot see through (and so cannot pair into packed instructions).
//
// Why (profiles/r6/r6_pk_forensics.txt): on gfx950, a packed fp32 instruction whose op_sel modifier makes a LOW result lane
// read the HIGH half of a source register pair — `v_pk_add_f32 vD, vA, vB op_sel:[0,1] op_sel_hi:[0,0]`, the form the SLP
// vectoriser gives to the x / y differences of a sampling point — delivered, about once in 150 training passes and only with
// a second PROCESS busy on the same GPU, a wrong result for one 16-lane pass of one wavefront (one wrong grad_loc_y from
// bit-identical inputs).  Established by class-by-class bisection on edited assembly (tools/probes/pk_repro/): replacing just
// those instructions by the scalar pairs they stand for removes the events (0 / 4,000 passes against 24 .. 38 / 4,000 with
// them; every other packed instruction of the kernel kept); s_nop padding, full waits before every packed instruction,
// copies of in-place operands do not.  The library therefore contains NO packed fp32 arithmetic with an op_sel bit set
// (tests/test_build_flags.py disassembles every code object), and the expressions the vectoriser would turn into one are
// written with these helpers.  No compiler flag is involved: the guarantee survives a maintainer's own build.
#pragma once
#include <hip/hip_runtime.h>

namespace bevmsda {

__device__ __forceinline__ float sub_scalar(float a, float b) {
  float d;
  asm("v_sub_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ float add_scalar(float a, float b) {
  float d;
  asm("v_add_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ float mul_scalar(float a, float b) {
  float d;
  asm("v_mul_f32_e32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// a * b + c, fused (what -ffp-contract=fast makes of the C++ expression)
__device__ __forceinline__ float fma_scalar(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

}  // namespace bevmsda
