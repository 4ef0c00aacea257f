// GEMM instantiations for the NN operand layout (see gemm.hpp).
#define PDT_AKS false
#define PDT_BKS true
#define PDT_LAYOUT_FN pdt_gemm_nn
#include "gemm_inst.hpp"
