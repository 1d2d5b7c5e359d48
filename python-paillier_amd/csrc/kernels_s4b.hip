// kernels_s4b.hip — split-modulus kernels for groups of 4 lanes, L in {18} (see split_kernels.inc)
#define PHE_PART s4b
#define PHE_PART_G 4
#define PHE_FOR_EACH_L(X) X(18)
#include "split_kernels.inc"
