// kernels_t16.hip — a*b mod N as one plain product + one fold against a per-key table in LDS (mul_table.h): 512-thread workgroups
// (8 wavefronts = 32 limb groups of 16 lanes), one per CU, the whole LDS of the CU to itself.
#include <hip/hip_runtime.h>
#include <stdint.h>

// clang-format off
#include "wave_gfx950.h"
#include "mont_core.h"
#include "mul_io.h"
#include "split_core.h"
#include "mul_table.h"
#include "mul_tile.h"
// clang-format on

namespace phe {

constexpr int kTableBlock = 512;

template <int L>
__global__ void __launch_bounds__(kTableBlock, 1) k_mulmod_table(TableMulArgs A) {
    constexpr int S = 16 * L, kRowT = S + kTableRowSlack, kGroups = kTableBlock / 16;
    using IO = RowIO<16, L>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_t[];
    uint32_t* tbl = lds_t;
    uint32_t* cst = tbl + (size_t)A.digits * S;
    uint32_t* rows = cst + 3 * S;
    uint32_t* stage = rows + kGroups * kRowT;
    // the table and the constant rows: global (L2) -> LDS once per workgroup, 16 bytes per thread and trip
    {
        const int n4 = A.digits * S / 4;  // (S is a multiple of 16)
        const Words4* src = reinterpret_cast<const Words4*>(A.table);
        Words4* dst = reinterpret_cast<Words4*>(tbl);
        for (int i = (int)threadIdx.x; i < n4; i += kTableBlock) dst[i] = src[i];
        for (int i = (int)threadIdx.x; i < S; i += kTableBlock) {
            cst[i] = A.n[i];
            cst[S + i] = A.ncomp[i];
            cst[2 * S + i] = A.ncomp1[i];
        }
    }
    __syncthreads();
    const uint32_t grp = threadIdx.x / 16, wv = threadIdx.x / 64u;
    mul_table_body<L>(A, rows + grp * kRowT, stage + wv * 2 * IO::kStageWave, tbl, cst, blockIdx.x * kGroups + grp, gridDim.x * kGroups,
                      threadIdx.x & 63u);
}

// mul_tile.h: the same product with one element per lane: the workgroup owns tiles of 64 products, its sixteen waves split the
// columns and meet at barriers between the phases; the fold's table words come through the scalar cache
// W = 16: 1024 threads, one workgroup per CU (four waves per SIMD: 128 registers each).  W = 8 (round 5; 1024-bit keys): 512 threads
// and 58 KB of LDS — two workgroups per CU, i.e. the same four waves per SIMD, in two groups whose barriers do not meet
template <int L, int W>
__global__ void __launch_bounds__(64 * W, W == 8 ? 4 : 1) k_mulmod_tile(TableMulArgs A) {
    using T = TileShape<L, W>;
    constexpr int kBlock = 64 * W;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_t[];
    uint32_t* tile = lds_t;
    uint32_t* prod_carry = tile + T::kRows * kTile;
    uint32_t* top = prod_carry + 2 * 2 * W * kTile;
    uint32_t* fold_carry = top + kTile * kTableRowSlack;
    uint32_t* cst = fold_carry + 2 * W * kTile;
    for (int i = (int)threadIdx.x; i < T::S; i += kBlock) {
        cst[i] = A.n[i];
        cst[T::S + i] = A.ncomp[i];
        cst[2 * T::S + i] = A.ncomp1[i];
    }
    __syncthreads();
    const uint32_t wv = wave::uniform(threadIdx.x / 64u);
    mul_tile_body<L, W>(A, tile, prod_carry, top, fold_carry, cst, wv, blockIdx.x, gridDim.x, threadIdx.x & 63u);
}

namespace t16 {

template <int L, int W>
static int launch_tile_L(int blocks, hipStream_t st, const TableMulArgs& A) {
    constexpr size_t lds_bytes = (size_t)tile_lds_words<L, W>() * 4;
    // (per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and one process may drive
    //  several — phe/fleet.py; a few microseconds beside a kernel of tens)
    if (hipFuncSetAttribute((const void*)k_mulmod_tile<L, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
    k_mulmod_tile<L, W><<<dim3(blocks), dim3(64 * W), lds_bytes, st>>>(A);
    return 0;
}
// workgroups of mul_tile.h a CU holds at once: LDS decides (one of 1024 threads, two of 512)
int tile_blocks_per_cu(int waves) { return waves == 8 ? 2 : 1; }
// (A.table: the column-block layout, cut for A.tile_waves waves)  -1: no kernel for this lane width and workgroup shape; -2: the
// device refused the LDS size
int launch_mul_tile(int L, int blocks, hipStream_t st, const TableMulArgs& A) {
    if (A.tile_waves == 8) return L == 9 ? launch_tile_L<9, 8>(blocks, st, A) : -1;
    if (A.tile_waves != kTileWaves) return -1;  // (the table's column blocks are cut for another workgroup shape)
    switch (L) {
        case 5: return launch_tile_L<5, kTileWaves>(blocks, st, A);
        case 9: return launch_tile_L<9, kTileWaves>(blocks, st, A);
        case 14: return launch_tile_L<14, kTileWaves>(blocks, st, A);
        default: return -1;
    }
}

template <int L>
static int launch_L(int blocks, size_t lds_bytes, hipStream_t st, const TableMulArgs& A) {
    // (the default cap of dynamic LDS is 64 KB; set per launch: see launch_tile_L)
    if (hipFuncSetAttribute((const void*)k_mulmod_table<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
    k_mulmod_table<L><<<dim3(blocks), dim3(kTableBlock), lds_bytes, st>>>(A);
    return 0;
}
// -1: no kernel for this lane width; -2: the device refused the LDS size
int launch_mul_table(int L, int blocks, size_t lds_bytes, hipStream_t st, const TableMulArgs& A) {
    switch (L) {
        case 5: return launch_L<5>(blocks, lds_bytes, st, A);
        case 9: return launch_L<9>(blocks, lds_bytes, st, A);
        default: return -1;
    }
}

}  // namespace t16
}  // namespace phe
