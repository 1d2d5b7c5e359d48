// placement_probe.hip — where do the wavefronts of small grids land?  (round 4: the wave-pair kernels lose 47 % going from 256
// to 512 workgroups of 128 threads although every wave should still find a SIMD of its own, profiles/r03g_*)
//
// Every wave records HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID, then runs a fixed chain-free v_mad_u64_u32 loop and
// reports the time of the whole launch.  Output: per (grid, block) how many SIMDs hold 0 / 1 / 2 / ... waves.
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip && ./placement_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <vector>

__global__ void k_probe(uint32_t* ids, uint64_t* sink, int iters) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63u) == 0u) {
        ids[2 * wave] = hw;
        ids[2 * wave + 1] = xcc;
    }
    uint64_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const uint32_t x = threadIdx.x | 1u, y = blockIdx.x | 3u;
    for (int i = 0; i < iters; ++i) {
        a0 = (uint64_t)x * y + a0; a1 = (uint64_t)x * y + a1; a2 = (uint64_t)x * y + a2; a3 = (uint64_t)x * y + a3;
        a4 = (uint64_t)x * y + a4; a5 = (uint64_t)x * y + a5; a6 = (uint64_t)x * y + a6; a7 = (uint64_t)x * y + a7;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x1234567u) sink[0] = a0;
}

int main() {
    const int max_waves = 1 << 16;
    uint32_t* d_ids;
    uint64_t* d_sink;
    hipMalloc((void**)&d_ids, max_waves * 8);
    hipMalloc((void**)&d_sink, 8);
    std::vector<uint32_t> ids(2 * max_waves);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks[] = {64, 128, 256};
    const int grids[] = {128, 256, 512, 1024, 2048};
    for (int lds : {0, 24 * 1024, 44 * 1024}) {
        for (int b : blocks) {
            for (int g : grids) {
                const int waves = g * b / 64;
                if (waves > max_waves) continue;
                k_probe<<<g, b, lds>>>(d_ids, d_sink, 10);  // warm
                hipDeviceSynchronize();
                hipEventRecord(e0);
                k_probe<<<g, b, lds>>>(d_ids, d_sink, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(ids.data(), d_ids, waves * 8, hipMemcpyDeviceToHost);
                std::map<uint32_t, int> per_simd, per_cu;
                for (int w = 0; w < waves; ++w) {
                    const uint32_t hw = ids[2 * w], xcc = ids[2 * w + 1] & 0xf;
                    const uint32_t simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    const uint32_t cu_key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
                    per_cu[cu_key]++;
                    per_simd[(cu_key << 2) | simd]++;
                }
                int hist[9] = {0};
                for (auto& kv : per_simd) hist[kv.second > 8 ? 8 : kv.second]++;
                int cu_hist[17] = {0};
                for (auto& kv : per_cu) cu_hist[kv.second > 16 ? 16 : kv.second]++;
                printf("lds %5d block %3d grid %4d waves %5d: %7.3f ms | CUs used %3zu SIMDs used %4zu | SIMDs with 1..8 waves:", lds, b, g,
                       waves, ms, per_cu.size(), per_simd.size());
                for (int k = 1; k <= 8; ++k) printf(" %d", hist[k]);
                printf(" | CUs with k waves:");
                for (int k = 1; k <= 16; ++k)
                    if (cu_hist[k]) printf(" %d:%d", k, cu_hist[k]);
                printf("\n");
            }
        }
    }
    return 0;
}
