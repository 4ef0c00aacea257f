// GEMM instantiations for the NT operand layout (see gemm.hpp).
#define PDT_AKS false
#define PDT_BKS false
#define PDT_LAYOUT_FN pdt_gemm_nt
#include "gemm_inst.hpp"
