// conv_v2.h instantiated for the "fp32h" kernel set: fp32 storage, ONE fp16 activation operand (the staged value rounded once, after the
// fp32 GroupNorm + SiLU), fp16 hi + lo weights: two MFMAs per product.  The kernel set of a reverse chain's steps 2 .. 8 (precision.py).
#define HSIDM_V2_E f16
#define HSIDM_V2_NP 2
#define HSIDM_V2_S float
#define HSIDM_V2_AP 1
#define HSIDM_V2_TILE0_ONLY 1
#define HSIDM_V2_TAG conv_v2_run_f32h
#include "conv_v2_inst.inc"
