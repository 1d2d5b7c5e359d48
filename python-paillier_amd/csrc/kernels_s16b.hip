// kernels_s16b.hip — split-modulus kernels for groups of 16 lanes, L in {7, 9} (see split_kernels.inc)
#define PHE_PART s16b
#define PHE_PART_G 16
#define PHE_FOR_EACH_L(X) X(7) X(9)
#include "split_kernels.inc"
