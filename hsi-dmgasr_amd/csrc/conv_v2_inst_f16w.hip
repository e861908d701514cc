// conv_v2.h instantiated for fp16 activations and (high, low) weight pairs: two MFMA passes per product.
#define HSIDM_V2_E f16
#define HSIDM_V2_NP 2
#define HSIDM_V2_TAG conv_v2_run_f16w
#include "conv_v2_inst.inc"
