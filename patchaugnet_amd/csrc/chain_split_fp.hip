// Fused shared-MLP chain kernels (pa_chain_kernel.h), instantiation family: shared tiles, feature propagation (3-NN interpolation prologue; MODE_FP with or without the folded first layer, MODE_FPX).
// One translation unit per family: see pa_chain_kernel.h.
#include "pa_chain_kernel.h"

template <int MODE>
static int split_fp(const PaChain &a, int rt, long ntiles, hipStream_t st)
{
    if (rt == 2) return launch_chain<2, 8, MODE, false, 4>(a, 4, ntiles, st);
    return launch_chain<1, 8, MODE, false, 4>(a, 4, ntiles, st);
}

int pa_chain_launch_split_fp(const PaChain &a, int mode, int rt, long ntiles, hipStream_t st)
{
    if (mode == MODE_FP) return split_fp<MODE_FP>(a, rt, ntiles, st);
    return split_fp<MODE_FPX>(a, rt, ntiles, st);
}
