// kernels_s8c.hip — split-modulus kernels for groups of 8 lanes, L in {18} (see split_kernels.inc)
#define PHE_PART s8c
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(18)
#include "split_kernels.inc"
