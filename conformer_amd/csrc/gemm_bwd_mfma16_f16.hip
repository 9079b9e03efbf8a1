// fp16 half of the 16-bit-MFMA backward GEMM (see gemm_bwd_mfma16.hip)
#define CFM_T16 _Float16
#define CFM_T16_FN f16
#include "gemm_bwd_mfma16_impl.h"
