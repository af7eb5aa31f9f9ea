// Tile configurations of the implicit-GEMM convolution whose main loop runs on v_mfma_f32_16x16x32_bf16
// (conv_igemm_kernel.h: Cfg<..., M16 = true>, conv_mainloop16).  bf16 only; a translation unit of its own so that it
// builds beside conv_igemm.hip.  Dispatch: conv_igemm.hip::launch_mfma (configurations 13 and up).
#include "conv_igemm_kernel.h"

namespace {

template <int TAPS>
int launch16(const ConvArgs& a, int cfg, hipStream_t st) {
  using T = bf16_t;
  switch (cfg) {
    case 13: return launch_cfg<Cfg<T, 8, 64, 4, 2, TAPS, 1, 4, true>>(a, st);    // 64 px x 32 couts per wave, eight waves
    case 14: return launch_cfg<Cfg<T, 16, 64, 4, 2, TAPS, 1, 4, true>>(a, st);   // 128 px x 32 couts per wave
    case 15: return launch_cfg<Cfg<T, 4, 64, 2, 2, TAPS, 1, 4, true>>(a, st);    // four waves
    case 16: return launch_cfg<Cfg<T, 2, 32, 2, 1, TAPS, 2, 4, true>>(a, st);    // the 32 x 32-pixel layers: K split over two wave groups
    case 17: return launch_cfg<Cfg<T, 4, 32, 4, 1, TAPS, 2, 4, true>>(a, st);
    // 18: 16 rows x 128 couts on the two-deep ring (32x32x16 MFMAs; lives here only to build beside conv_igemm.hip): 0.58 of
    // the 8x128 tile's LDS-DMA pieces per FLOP - the term that binds this kernel's issue port
    case 18: return launch_cfg<Cfg<T, 16, 128, 4, 2, TAPS, 1, 2>>(a, st);
    default: return STV_ERR_ARG;
  }
}

}  // namespace

int stv_conv_launch_m16(const ConvArgs& a, int cfg, int taps, hipStream_t st) {
  return taps == 9 ? launch16<9>(a, cfg, st) : launch16<1>(a, cfg, st);
}
