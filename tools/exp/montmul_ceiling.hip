// Experiment (not part of the product): the bare squaring loop of mont_core.h at several occupancies, to see
// how far the full modexp kernels are from what the montmul loop alone sustains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../../python-paillier_amd/csrc/wave_gfx950.h"
#include "../../python-paillier_amd/csrc/mont_core.h"
using namespace phe;
template <int G, int L>
__global__ void __launch_bounds__(256) k_sq(const uint32_t* nn, uint32_t* io, int reps, uint32_t n0inv) {
    constexpr int S = G * L, kGroups = 256 / G;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kGroups * (S + kLdsPad)];
    const Lanes<G> ln(threadIdx.x & 63u);
    uint32_t* row = lds + (threadIdx.x / G) * (S + kLdsPad);
    uint32_t n[L], acc[L];
    load_row<L>(n, nn, ln.g);
    load_row<L>(acc, io + (size_t)(blockIdx.x * kGroups + threadIdx.x / G) * S, ln.g);
    for (int r = 0; r < reps; ++r) {
        lds_put<L>(row, acc, ln.g);
        montmul<G, L>(acc, row, acc, n, n0inv, ln);
    }
    store_row<L>(io + (size_t)(blockIdx.x * kGroups + threadIdx.x / G) * S, acc, ln.g);
}
template <int G, int L>
void run(const char* name, int cus) {
    constexpr int S = G * L, kGroups = 256 / G;
    const int reps = 2000;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        if (L > 18 && bpc > 2) break;
        const int blocks = cus * bpc;
        std::vector<uint32_t> h((size_t)blocks * kGroups * S, 0x0abcdef1u & 0x1fffffffu), n(S, 0x1ffffffdu);
        n[0] = 0x1ffffffbu | 1u;
        uint32_t *dn, *dio;
        hipMalloc((void**)&dn, S * 4); hipMalloc((void**)&dio, h.size() * 4);
        hipMemcpy(dn, n.data(), S * 4, hipMemcpyHostToDevice); hipMemcpy(dio, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_sq<G, L><<<blocks, 256>>>(dn, dio, 10, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k_sq<G, L><<<blocks, 256>>>(dn, dio, reps, 12345u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double mm = (double)blocks * kGroups * reps / (ms * 1e-3);
        const double macs = mm * 2.0 * S * S;   // multiplies actually issued (radix-29 limbs)
        printf("{\"kernel\": \"%s\", \"blocks_per_cu\": %d, \"ms\": %.3f, \"montmul_per_s\": %.4e, \"issued_TMAC_per_s\": %.2f}\n", name, bpc, ms, mm, macs / 1e12);
        hipFree(dn); hipFree(dio);
    }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    run<8, 18>("sq<8,18>", p.multiProcessorCount);
    run<4, 18>("sq<4,18>", p.multiProcessorCount);
    run<8, 9>("sq<8,9>", p.multiProcessorCount);
    run<16, 9>("sq<16,9>", p.multiProcessorCount);
    run<4, 36>("sq<4,36>", p.multiProcessorCount);
    run<8, 27>("sq<8,27>", p.multiProcessorCount);
    run<16, 14>("sq<16,14>", p.multiProcessorCount);
    return 0;
}
