// rb_device_common.hpp -- shared prelude of the device code (included by rb_kernels.hip only).
#pragma once
#include <hip/hip_runtime.h>

#include "rb_internal.hpp"

#pragma clang fp contract(off)

#define DEV __device__ __forceinline__

// Scene data is immutable for the duration of a launch.  Reading it through the
// constant address space lets the compiler use scalar loads (s_load_*: one
// fetch per wavefront, operands land in SGPRs) whenever the address is
// wave-uniform -- spheres, lights, a single-leaf BVH -- and ordinary vector loads
// otherwise.  Without this every lane issues its own VMEM load of the same
// address, because the kernels also store (accumulation, traversal stack).
#define RB_CONST __attribute__((address_space(4)))
template <class T>
DEV const RB_CONST T* cptr(const T* p) {
    return (const RB_CONST T*)p;
}
// native vector types: HIP's float4/uint4 classes cannot be copy-constructed from address space 4
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef const RB_CONST v4f* cf4p;
typedef const RB_CONST v4u* cu4p;
typedef v4f nt_f4;  // nontemporal builtins want a native vector

