// Fused shared-MLP chain kernels (pa_chain_kernel.h), instantiation family: pooled set-abstraction tilings: neighbour-major rows, the max over the neighbourhood in registers (wave-private or shared 4-group tiles).
// One translation unit per family: see pa_chain_kernel.h.
#include "pa_chain_kernel.h"

int pa_chain_launch_pooled(const PaChain &a, int rt, bool split, int wpw, long ntiles, hipStream_t st)
{
    if (split) {
        // Shared 4-group tiles.  The four-column-tile chunking keeps 170 registers = 3 workgroups per CU; with the last layer in two-tile chunks
        // (NCMAX = 2: 110 registers, same sums in the same order) FOUR fit, and 1024 tiles (the second level at batch 32) run as ONE round on
        // the 256 CUs instead of 768 + 256.  PA_CHAIN_POOLED_NC4 = A/B knob for the former tiling.
        // A hidden layer is written back in place and must stay ONE chunk per wave: n / 64 <= 2 column tiles for the two-tile chunking.
        static const bool nc4 = getenv("PA_CHAIN_POOLED_NC4") != nullptr;
        bool narrow = true;
        for (int l = 0; l + 1 < a.nlayers; ++l) narrow = narrow && a.L[l].n <= 128;
        if (nc4 || !narrow) return launch_chain<5, 4, MODE_SA, true, 4>(a, 4, ntiles, st);
        return launch_chain<5, 2, MODE_SA, true, 4>(a, 4, ntiles, st);
    }
    switch (rt) {
        case 4: return launch_chain<4, 4, MODE_SA, true, 1>(a, wpw, ntiles, st);
        case 5: return launch_chain<5, 4, MODE_SA, true, 1>(a, wpw, ntiles, st);
        case 8: return launch_chain<8, 4, MODE_SA, true, 1>(a, wpw, ntiles, st);
        default: return -1;      // nsample range not built: chain_dispatch reports PA_EUNSUPPORTED
    }
}
