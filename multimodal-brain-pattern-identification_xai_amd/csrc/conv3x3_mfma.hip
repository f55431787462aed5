// MFMA implicit-GEMM 3x3 convolution (bf16 operands, fp32 accumulate) -- forward / data-gradient and
// weight-gradient.  (placeholder: the entry points report "unsupported" until the kernels land)
#include "bx_common.h"

extern "C" size_t bx_conv3x3_packed_mfma_bytes(int I_p, int O_p) { (void)I_p; (void)O_p; return 0; }
int bx_conv3x3_mfma_supported(int Ci, int Co, int dtype) { (void)Ci; (void)Co; (void)dtype; return 0; }
void bx_conv3x3_mfma_pack_launch(const float*, void*, int, int, int, int, int, hipStream_t) {}
int bx_conv3x3_mfma_launch(const void*, const void*, const float*, const void*, const void*, void*, int, int, int, int, int, int, hipStream_t) {
  BX_FAIL(BX_EUNSUPPORTED, "MFMA conv path not built");
}
size_t bx_wgrad_mfma_workspace(int, int, int, int, int) { return 0; }
int bx_wgrad_mfma_supported(int, int, int) { return 0; }
int bx_wgrad_mfma_launch(const void*, const void*, float*, float*, int, int, int, int, int, int, void*, size_t, hipStream_t) {
  BX_FAIL(BX_EUNSUPPORTED, "MFMA wgrad path not built");
}
