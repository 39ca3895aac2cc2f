// conv3x tilings for f16 tensors, 2D (see conv3x_impl.h; split per dtype and dimension for build time)
#include "conv3x_impl.h"

namespace seg {
namespace c3x {
template <> bool launch_2d<f16>(int id, const Conv3xArgs& a, hipStream_t s) {
    typedef f16 T;
    SEG_C3X_2D_BODY
}
}  // namespace c3x
}  // namespace seg
