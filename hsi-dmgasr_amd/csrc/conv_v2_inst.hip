// Instantiations of the persistent 3x3 convolution (conv_v2.h) for bf16 + the helpers shared by all element types.
#define HSIDM_V2_E bf16
#define HSIDM_V2_NP 1
#define HSIDM_V2_TAG conv_v2_run_bf16
#include "conv_v2_inst.inc"

namespace hsidm {

unsigned long long* g_stamps = nullptr;   // diagnostic builds only


void conv_v2_set_stamps(unsigned long long* p) { g_stamps = p; }

int conv_v2_subs(int tile_kind, int bn) {
    const int wm = 4 / (bn / 32);
    if (tile_kind == 2) return wm;                             // one-image 8x8 tile (bn = 128: one entry)
    return tile_kind == 0 ? wm : (wm >= 2 ? wm / 2 : 1);
}

int conv_v2_slots() { return 2 * device_cus(); }      // co-resident workgroups: 2 per CU


}  // namespace hsidm
