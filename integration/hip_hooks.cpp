// The hooks of the reference's JIT backend that its generic headers call on "CUDA" arrays (include/enoki/cuda.h:33-63,
// 181; callers: array_struct.h scatter / gather wrappers, array_router.h any_or, array_macro.h) -- for an eager backend
// there is no trace to flush and no variable to mark, so they do nothing.  A maintainer would compile this file into the
// library that ships integration/enoki/hip.h; the reference tree itself is untouched.
#include <enoki/array.h>
#include <enoki/cuda.h>
#include <enoki_hip.h>

NAMESPACE_BEGIN(enoki)
void cuda_eval(bool) { }
void cuda_eval_var(uint32_t, bool) { }
void cuda_sync() { ek_hip_sync(); }
void cuda_var_mark_dirty(uint32_t) { }
void cuda_set_scatter_gather_operand(uint32_t, bool) { }
NAMESPACE_END(enoki)
