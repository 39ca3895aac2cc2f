// conv3x tilings for f16 tensors, 3D, with the producer's GroupNorm + dropout + ReLU applied while the halo is staged (conv3x_impl.h, FUSE)
#include "conv3x_impl.h"

namespace seg {
namespace c3x {
template <> bool launch_3d_gn<f16>(int id, const Conv3xArgs& a, hipStream_t s) {
    typedef f16 T;
    constexpr bool FUSE = true;
    SEG_C3X_3D_GN_BODY
}
}  // namespace c3x
}  // namespace seg
