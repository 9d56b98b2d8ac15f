// Fused shared-MLP chain kernels (pa_chain_kernel.h), instantiation family: shared tiles (the four waves of a workgroup split every layer's columns), plain rows: pa_linear and the small pre-multiplies.
// One translation unit per family: see pa_chain_kernel.h.
#include "pa_chain_kernel.h"

int pa_chain_launch_split_plain(const PaChain &a, int rt, long ntiles, hipStream_t st)
{
    if (rt == 2) return launch_chain<2, 8, MODE_PLAIN, false, 4>(a, 4, ntiles, st);
    return launch_chain<1, 8, MODE_PLAIN, false, 4>(a, 4, ntiles, st);
}
