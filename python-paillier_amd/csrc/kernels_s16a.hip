// kernels_s16a.hip — split-modulus kernels for groups of 16 lanes, L in {1, 2, 3, 4, 5} (see split_kernels.inc)
#define PHE_PART s16a
#define PHE_PART_G 16
#define PHE_FOR_EACH_L(X) X(1) X(2) X(3) X(4) X(5)
#include "split_kernels.inc"
