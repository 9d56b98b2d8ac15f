// Fused shared-MLP chain kernels (pa_chain_kernel.h), instantiation family: wave-private tiles, plain rows / set-abstraction gather / feature propagation (large launches: every wave owns its rows end to end).
// One translation unit per family: see pa_chain_kernel.h.
#include "pa_chain_kernel.h"

template <int MODE>
static int wp_rows(const PaChain &a, int rt, int wpw, long ntiles, hipStream_t st)
{
    if (rt == 1 && MODE != MODE_SA) return launch_chain<1, 16, MODE, false, 1>(a, wpw, ntiles, st);
    return launch_chain<2, 16, MODE, false, 1>(a, wpw, ntiles, st);
}

int pa_chain_launch_wp_rows(const PaChain &a, int mode, int rt, int wpw, long ntiles, hipStream_t st)
{
    if (mode == MODE_PLAIN) return wp_rows<MODE_PLAIN>(a, rt, wpw, ntiles, st);
    if (mode == MODE_SA) return wp_rows<MODE_SA>(a, rt, wpw, ntiles, st);
    return wp_rows<MODE_FP>(a, rt, wpw, ntiles, st);
}
