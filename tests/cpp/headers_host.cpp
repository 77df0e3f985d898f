// Every header name of the reference that user code may include -- fwd.h first, then the per-topic array_*.h names (which
// forward to enoki/array.h here), then the type headers -- in ONE translation unit: nothing clashes, forward declarations
// match the definitions.  Compile-only (the objects need libenoki-hip.so to link); tests/test_half.py builds it.
//
//     g++ -std=c++17 -Iinclude -c tests/cpp/headers_host.cpp
#include <enoki/fwd.h>
namespace mine { enoki::HIPArray<float> *make(); void take(const enoki::Array<enoki::DiffArray<enoki::HIPArray<float>>, 3> &); }
#include <enoki/array_traits.h>
#include <enoki/array_router.h>
#include <enoki/array_utils.h>
#include <enoki/array_struct.h>
#include <enoki/array_idiv.h>
#include <enoki/array_macro.h>
#include <enoki/array_base.h>
#include <enoki/array_generic.h>
#include <enoki/array_static.h>
#include <enoki/array_masked.h>
#include <enoki/array_fallbacks.h>
#include <enoki/array_round.h>
#include <enoki/array_math.h>
#include <enoki/cuda.h>
#include <enoki/dynamic.h>      // (without ENOKI_HIP_DYNAMIC_IS_DEVICE: the name exists, using it is a compile error -- tests/test_half.py)
#include <enoki/autodiff.h>
#include <enoki/matrix.h>
#include <enoki/complex.h>
#include <enoki/quaternion.h>
#include <enoki/random.h>
#include <enoki/half.h>
#include <enoki/stl.h>
#include <enoki/fwd.h>
int main() { enoki::Array<float, 3> v(1.f, 2.f, 3.f); return enoki::hsum(v) == 6.f && enoki::poly2(v, 1.0, 0.0, 1.0)[1] == 5.f ? 0 : 1; }
