// kernels_s2b.hip — split-modulus kernels for groups of 2 lanes, L in {18} (see split_kernels.inc)
#define PHE_PART s2b
#define PHE_PART_G 2
#define PHE_FOR_EACH_L(X) X(18)
#include "split_kernels.inc"
