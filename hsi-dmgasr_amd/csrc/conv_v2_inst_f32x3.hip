// conv_v2.h instantiated for the fp32 mode: fp32 storage, bf16 hi + lo activations AND weights, three MFMAs per product.
#define HSIDM_V2_E bf16
#define HSIDM_V2_NP 2
#define HSIDM_V2_S float
#define HSIDM_V2_AP 2
#define HSIDM_V2_TAG conv_v2_run_f32x3
#include "conv_v2_inst.inc"
