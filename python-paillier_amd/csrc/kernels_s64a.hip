// kernels_s64a.hip — split-modulus kernels for the whole wavefront as one limb group, L in {1, 2} (see split_kernels.inc)
#define PHE_PART s64a
#define PHE_PART_G 64
#define PHE_FOR_EACH_L(X) X(1) X(2)
#include "split_kernels.inc"
