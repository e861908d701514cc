// one translation unit per (precision, geometry): fp16 storage / operands, weights as hi + lo (conv_inst.inc)
#define HSIDM_PREC_BF16 2
#define HSIDM_KS 3
#define HSIDM_S 2
#define HSIDM_NCHW 0
#define HSIDM_TAG conv_run_f16_k3s2
#include "conv_inst.inc"
