// kernels_s2a.hip — split-modulus kernels for groups of 2 lanes, L in {9} (see split_kernels.inc)
#define PHE_PART s2a
#define PHE_PART_G 2
#define PHE_FOR_EACH_L(X) X(9)
#include "split_kernels.inc"
