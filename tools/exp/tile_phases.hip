// tile_phases.hip — where does k_mulmod_tile<9> (csrc/mul_tile.h) spend its clocks?  Measurement only: the kernel built with
// PHE_TILE_PROFILE sums the shader clocks of every wave per phase (load | product | barrier | carries in + barrier | fold |
// carries + barrier | settle | barrier).  Operands and table are random words: the timing does not depend on the values, the results are not checked.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -DPHE_TILE_PROFILE \
//         -I python-paillier_amd/csrc -o tools/exp/tile_phases tools/exp/tile_phases.hip && tools/exp/tile_phases
#include "../../python-paillier_amd/csrc/kernels_t16.hip"

#include <stdio.h>
#include <stdlib.h>

#include <vector>

int main(int argc, char** argv) {
    using namespace phe;
    constexpr int L = 9, S = 16 * L, P = 142, D = 143, DP = 144;
    const size_t batch = argc > 1 ? (size_t)atol(argv[1]) : (size_t)1 << 20;
    const int limbs = 128;
    std::vector<uint32_t> h((size_t)batch * limbs);
    for (auto& w : h) w = (uint32_t)rand() * 2654435761u + (uint32_t)rand();
    uint32_t *a, *b, *o, *cst, *tbl;
    uint64_t* prof;
    hipMalloc((void**)&a, h.size() * 4);
    hipMalloc((void**)&b, h.size() * 4);
    hipMalloc((void**)&o, h.size() * 4);
    hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data() + 64, (h.size() - 64) * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> t((size_t)kTileWaves * (DP + kFoldPadRows) * (S / kTileWaves) + 3 * S);
    for (auto& w : t) w = ((uint32_t)rand() * 2654435761u) & kLimbMask;
    hipMalloc((void**)&tbl, t.size() * 4);
    hipMemcpy(tbl, t.data(), t.size() * 4, hipMemcpyHostToDevice);
    cst = tbl + (size_t)kTileWaves * (DP + kFoldPadRows) * (S / kTileWaves);
    hipMalloc((void**)&prof, 64 + 256);
    TableMulArgs A;
    A.n = cst; A.ncomp = cst + S; A.ncomp1 = cst + 2 * S; A.table = tbl; A.inv = 1e-25;  // (q of ~2^33 with a random fraction: the settle takes its single candidate on 15 tiles of 16, as on real residues)
     A.split = P; A.digits = D; A.digits_padded = DP;
    A.base = P - 2; A.a = a; A.b = b; A.out = o; A.a_stride = A.b_stride = A.out_stride = limbs; A.limbs = limbs; A.batch = batch;
    A.profile = prof; A.tile_waves = kTileWaves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int blocks : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(prof, 0, 64 + 256);
            hipEventRecord(e0);
            if (t16::launch_mul_tile(L, blocks, 0, A) != 0) { printf("launch failed\n"); return 1; }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t p[40];
        hipMemcpy(p, prof, 64 + 256, hipMemcpyDeviceToHost);
        double tot = 0;
        for (int i = 0; i < 8; ++i) tot += (double)p[i];
        printf("product clocks per tile by wave:");
        for (int w = 0; w < 16; ++w) printf(" %.0f", (double)p[8 + w] / ((batch + 63) / 64));
        printf("\nfold clocks per tile by wave:   ");
        for (int w = 0; w < 16; ++w) printf(" %.0f", (double)p[24 + w] / ((batch + 63) / 64));
        printf("\n");
        printf("blocks %d: %.3f ms, %.1f M products/s; share of the waves' clocks: load %.1f%% | product %.1f%% | barrier %.1f%% | carries in + barrier %.1f%% | "
               "fold %.1f%% | carries + barrier %.1f%% | settle %.1f%% | barrier %.1f%%   (clocks per wave and tile: %.0f)\n",
               blocks, ms, batch / ms / 1e3, 100 * p[0] / tot, 100 * p[1] / tot, 100 * p[2] / tot, 100 * p[3] / tot, 100 * p[4] / tot,
               100 * p[5] / tot, 100 * p[6] / tot, 100 * p[7] / tot, tot / ((double)kTileWaves * ((batch + 63) / 64)));
    }
    return 0;
}
