// conv_v2.h instantiated for fp16 activations / weights, one weight pass.
#define HSIDM_V2_E f16
#define HSIDM_V2_NP 1
#define HSIDM_V2_TAG conv_v2_run_f16
#include "conv_v2_inst.inc"
