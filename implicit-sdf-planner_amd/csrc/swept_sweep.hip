// V1 swept-volume sweep — placeholder translation unit (kernels land in the next commit).
#include "isdf_internal.hpp"
namespace isdf {
void launch_swept_prepare(const SweptParams &, hipStream_t) {}
void launch_swept_sweep(const SweptParams &, hipStream_t) {}
void launch_swept_reduce(const SweptParams &, double *, hipStream_t) {}
}
