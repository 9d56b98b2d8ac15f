// Fused shared-MLP chain kernels (pa_chain_kernel.h), instantiation family: wave-private tiles, MODE_FPX: the finest feature-propagation level (pa_fp_chain_premul with c1 <= 4) -- the dominant kernel of the step.
// One translation unit per family: see pa_chain_kernel.h.
#include "pa_chain_kernel.h"

int pa_chain_launch_wp_fpx(const PaChain &a, int rt, int wpw, long ntiles, hipStream_t st)
{
    if (rt == 1) return launch_chain<1, 16, MODE_FPX, false, 1>(a, wpw, ntiles, st);
    return launch_chain<2, 16, MODE_FPX, false, 1>(a, wpw, ntiles, st);
}
