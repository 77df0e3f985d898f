// tests/cpp/refshim/enoki/dynamic.h -- what the reference's tests get when they `#include <enoki/dynamic.h>` in the
// retargeted build: this repository's compatibility header (compat/enoki/dynamic.h, the opt-in include root: DynamicArray<Packet<T>> = HIPArray<T>)
// plus the throwing assert() of the shim.
#pragma once

// the retargeted build asks for the substitution explicitly (compat/enoki/dynamic.h: opt-in)
#ifndef ENOKI_HIP_DYNAMIC_IS_DEVICE
#  define ENOKI_HIP_DYNAMIC_IS_DEVICE 1
#endif
#include_next <enoki/dynamic.h>

#include <iostream>

#include <test_assert.h>
