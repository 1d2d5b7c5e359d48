// kernels_radix.hip — the batch form of the reference's decimal wire format (radix_conv.h): one number per thread,
// 64 numbers per workgroup.  The workgroup loads its 64 rows with coalesced reads into an LDS tile indexed
// [word][thread] (row pitch 65 words: conflict-free both for the transposing load and for the per-thread walks),
// every thread converts its own number inside the tile, and the results leave the same way.
// HBM traffic is the algorithmic minimum (each word / digit read once, written once); the conversion itself is
// O(words^2) full-rate 32-bit VALU work per number.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PHE_DEV __device__ __forceinline__
#include "radix_conv.h"

namespace phe {

constexpr int kRadixBlock = 64;             // threads (= numbers) per workgroup: one wavefront
constexpr int kRadixPitch = kRadixBlock + 1;  // words between consecutive limbs of one number in the tile

struct TileWords {
    uint32_t* base;  // tile + thread index
    __device__ __forceinline__ uint32_t& operator()(int j) const { return base[j * kRadixPitch]; }
};

__device__ __forceinline__ void tile_load(uint32_t* tile, const uint32_t* rows, int words, uint64_t first, uint64_t batch) {
    const uint64_t live = batch - first < (uint64_t)kRadixBlock ? batch - first : (uint64_t)kRadixBlock;
    const uint32_t total = (uint32_t)live * (uint32_t)words;
    const uint32_t* src = rows + first * (uint64_t)words;
    for (uint32_t i = threadIdx.x; i < total; i += kRadixBlock) {
        const uint32_t r = i / (uint32_t)words, j = i - r * (uint32_t)words;
        tile[j * kRadixPitch + r] = src[i];
    }
}

__device__ __forceinline__ void tile_store(uint32_t* rows, const uint32_t* tile, int words, uint64_t first, uint64_t batch) {
    const uint64_t live = batch - first < (uint64_t)kRadixBlock ? batch - first : (uint64_t)kRadixBlock;
    const uint32_t total = (uint32_t)live * (uint32_t)words;
    uint32_t* dst = rows + first * (uint64_t)words;
    for (uint32_t i = threadIdx.x; i < total; i += kRadixBlock) {
        const uint32_t r = i / (uint32_t)words, j = i - r * (uint32_t)words;
        dst[i] = tile[j * kRadixPitch + r];
    }
}

// limbs (batch, words) -> digits (batch, width) ASCII; *bad = first row that needs more than `width` digits
__global__ void __launch_bounds__(kRadixBlock) k_to_decimal(const uint32_t* limbs, int words, char* digits, int width,
                                                            uint64_t batch, unsigned long long* bad) {
    extern __shared__ uint32_t tile[];
    for (uint64_t first = (uint64_t)blockIdx.x * kRadixBlock; first < batch; first += (uint64_t)gridDim.x * kRadixBlock) {
        tile_load(tile, limbs, words, first, batch);
        __syncthreads();
        const uint64_t row = first + threadIdx.x;
        if (row < batch) {
            if (!limbs_to_decimal(TileWords{tile + threadIdx.x}, words, digits + row * (uint64_t)width, width))
                atomicMin(bad, (unsigned long long)row);
        }
        __syncthreads();
    }
}

// digits (batch, width) ASCII -> limbs (batch, words); *bad_char / *bad_size = first offending row
__global__ void __launch_bounds__(kRadixBlock) k_from_decimal(const char* digits, int width, uint32_t* limbs, int words,
                                                              uint64_t batch, unsigned long long* bad_char,
                                                              unsigned long long* bad_size) {
    extern __shared__ uint32_t tile[];
    for (uint64_t first = (uint64_t)blockIdx.x * kRadixBlock; first < batch; first += (uint64_t)gridDim.x * kRadixBlock) {
        const uint64_t row = first + threadIdx.x;
        if (row < batch) {
            const int st = decimal_to_limbs(digits + row * (uint64_t)width, width, TileWords{tile + threadIdx.x}, words);
            if (st == 1) atomicMin(bad_char, (unsigned long long)row);
            if (st == 2) atomicMin(bad_size, (unsigned long long)row);
        }
        __syncthreads();
        tile_store(limbs, tile, words, first, batch);
        __syncthreads();
    }
}

namespace radix {

size_t tile_bytes(int words) { return (size_t)words * kRadixPitch * sizeof(uint32_t); }

int launch_to_decimal(const uint32_t* limbs, int words, char* digits, int width, uint64_t batch, unsigned long long* bad,
                      int max_blocks, hipStream_t st) {
    const uint64_t want = (batch + kRadixBlock - 1) / kRadixBlock;
    const int blocks = (int)(want < (uint64_t)max_blocks ? want : (uint64_t)max_blocks);
    if (tile_bytes(words) > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_to_decimal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes(words)) != hipSuccess)
        return -1;
    k_to_decimal<<<dim3(blocks), dim3(kRadixBlock), tile_bytes(words), st>>>(limbs, words, digits, width, batch, bad);
    return 0;
}

int launch_from_decimal(const char* digits, int width, uint32_t* limbs, int words, uint64_t batch,
                        unsigned long long* bad_char, unsigned long long* bad_size, int max_blocks, hipStream_t st) {
    const uint64_t want = (batch + kRadixBlock - 1) / kRadixBlock;
    const int blocks = (int)(want < (uint64_t)max_blocks ? want : (uint64_t)max_blocks);
    if (tile_bytes(words) > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_from_decimal, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_bytes(words)) != hipSuccess)
        return -1;
    k_from_decimal<<<dim3(blocks), dim3(kRadixBlock), tile_bytes(words), st>>>(digits, width, limbs, words, batch, bad_char,
                                                                               bad_size);
    return 0;
}

}  // namespace radix
}  // namespace phe
