// bf16 half of the 16-bit weight-gradient GEMM (see gemm_dw16_impl.h)
#define CFM_T16 __bf16
#define CFM_T16_FN bf16
#include "gemm_dw16_impl.h"
