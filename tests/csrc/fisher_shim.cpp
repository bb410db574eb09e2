// Host build of minorseq_amd/csrc/jl_fisher.h so the algorithm the device runs can be checked against
// the mpmath golden vectors without a GPU (tests/test_fisher_host.py).  Not part of the product.
#include "../../minorseq_amd/csrc/jl_fisher.h"
extern "C" double shim_fisher(uint32_t a, uint32_t c, uint32_t n, double *lp)
{
    return jl_fisher_greater_equal_rows(a, c, n, lp);
}
