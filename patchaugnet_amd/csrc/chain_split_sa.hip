// Fused shared-MLP chain kernels (pa_chain_kernel.h), instantiation family: shared tiles, set-abstraction gather in group order: unpooled output, or the max over the neighbourhood folded into the epilogue (atomicMax).
// One translation unit per family: see pa_chain_kernel.h.
#include "pa_chain_kernel.h"

int pa_chain_launch_split_sa(const PaChain &a, int rt, bool atomic_pool, long ntiles, hipStream_t st)
{
    if (atomic_pool) {
        if (rt == 2) return launch_chain<2, 8, MODE_SA, false, 4, true>(a, 4, ntiles, st);
        return launch_chain<1, 8, MODE_SA, false, 4, true>(a, 4, ntiles, st);
    }
    if (rt == 2) return launch_chain<2, 8, MODE_SA, false, 4>(a, 4, ntiles, st);
    return launch_chain<1, 8, MODE_SA, false, 4>(a, 4, ntiles, st);
}
