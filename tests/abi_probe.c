/* Prints the layout of the C-ABI structs so the test can compare them with the ctypes mirror. */
#include <stddef.h>
#include <stdio.h>
#include "hsidm.h"
int main(void) {
    printf("%zu %zu\n", sizeof(hsidm_conv_phase), sizeof(hsidm_conv_desc));
    printf("%zu %zu %zu %zu %zu %zu %zu\n", offsetof(hsidm_conv_phase, src0), offsetof(hsidm_conv_phase, src1),
           offsetof(hsidm_conv_phase, gn_ab), offsetof(hsidm_conv_phase, C0), offsetof(hsidm_conv_phase, C1),
           offsetof(hsidm_conv_phase, transform), offsetof(hsidm_conv_phase, ntaps));
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(hsidm_conv_desc, nphase),
           offsetof(hsidm_conv_desc, w_hi), offsetof(hsidm_conv_desc, w_lo), offsetof(hsidm_conv_desc, bias),
           offsetof(hsidm_conv_desc, film), offsetof(hsidm_conv_desc, film_stride), offsetof(hsidm_conv_desc, res),
           offsetof(hsidm_conv_desc, res_scale), offsetof(hsidm_conv_desc, out), offsetof(hsidm_conv_desc, stats),
           offsetof(hsidm_conv_desc, B), offsetof(hsidm_conv_desc, ksize), offsetof(hsidm_conv_desc, prec),
           offsetof(hsidm_conv_desc, bn), offsetof(hsidm_conv_desc, workspace), offsetof(hsidm_conv_desc, workspace_bytes), offsetof(hsidm_conv_desc, w_v2_lo),
           offsetof(hsidm_conv_desc, w_v2_ls), offsetof(hsidm_conv_desc, w_v2_li));
    return 0;
}
