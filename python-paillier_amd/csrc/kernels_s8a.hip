// kernels_s8a.hip — split-modulus kernels for groups of 8 lanes, L in {5, 7, 9} (see split_kernels.inc)
#define PHE_PART s8a
#define PHE_PART_G 8
#define PHE_FOR_EACH_L(X) X(5) X(7) X(9)
#include "split_kernels.inc"
