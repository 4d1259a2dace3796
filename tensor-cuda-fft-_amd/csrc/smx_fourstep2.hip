// smx_fourstep2.hip -- four-step path, two-level columns with a first-level length L1 = 9 ... 15 (round 3):
// tile counts L = 36 ... 60 (x 4 threads per column pair), 72 ... 120 (x 8), 144 ... 240 (x 16), i.e. N = 9216 ... 61440.
// These lengths ran as band groups (one pass over x and y per 512 bins) before.  The kernel is k_fs_big of
// smx_fs_big.h, the arithmetic fsb_* of smx_core.h; a translation unit of its own so that the 105 instantiations
// compile beside smx_fourstep.hip.
//
// Replaces for those lengths: reference fft_tensor/spectral_enhancements.py:147, :164 (PhaseAware: rfft / irfft at
// any T), complex_rope.py:207, :216, fft_lm/frequency_native.py:314-317, :359-360, fft_tensor/frequency_ops.py:201
// (the complex sequence FFT of fnet_attention).
#include "smx_fs_big.h"

namespace smx {

template <int L2>
static hipError_t launch_general_l2(const DecimArgs& a, int mode, int l1, hipStream_t s) {
  switch (l1) {
#define SMX_FS2_CASE(LL) case LL: launch_fs_big_t<L2, LL>(a, mode, s); break;
    SMX_FS2_CASE(9) SMX_FS2_CASE(10) SMX_FS2_CASE(11) SMX_FS2_CASE(12) SMX_FS2_CASE(13) SMX_FS2_CASE(14) SMX_FS2_CASE(15)
#undef SMX_FS2_CASE
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_fs_big_general(const DecimArgs& a, int mode, int l1, int l2, hipStream_t s) {
  switch (l2) {
    case 4: return launch_general_l2<4>(a, mode, l1, s);
    case 8: return launch_general_l2<8>(a, mode, l1, s);
    case 16: return launch_general_l2<16>(a, mode, l1, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace smx
