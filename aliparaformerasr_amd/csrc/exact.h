// exact.h — separately-rounded fp32 mul/add/sub.  HIP's __fmul_rn/__fadd_rn are plain
// operators and hipcc (-ffp-contract=fast-honor-pragmas) would contract a*b+c into one FMA;
// the reference's graphs (and the numpy oracle) round the product and the sum separately.
#pragma once
#include <hip/hip_runtime.h>

namespace pf {
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
}  // namespace pf
