// GEMM instantiations for the TN operand layout (see gemm.hpp).
#define PDT_AKS true
#define PDT_BKS true
#define PDT_LAYOUT_FN pdt_gemm_tn
#include "gemm_inst.hpp"
