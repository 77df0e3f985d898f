// tests/cpp/refshim/enoki/dynamic.h -- what the reference's tests get when they `#include <enoki/dynamic.h>` in the
// retargeted build: this repository's compatibility header (include/enoki/dynamic.h: DynamicArray<Packet<T>> = HIPArray<T>)
// plus the throwing assert() of the shim.
#pragma once

#include_next <enoki/dynamic.h>

#include <iostream>

#include <test_assert.h>
