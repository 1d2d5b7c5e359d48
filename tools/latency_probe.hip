// latency_probe.hip — where the time of ONE exponentiation on a wave pair goes (measurement tool, not part of the library).
//
// The wave-pair kernels (split_core.h, "one number on TWO wavefronts") are latency code: one wavefront per SIMD, issue in
// order, nothing to hide a dependent chain behind.  This tool times their pieces on a lone wavefront with s_memtime:
//   - the sweeps themselves (ab_first_word / ab_second_word of the shipped headers) at several row counts: slope = cycles per
//     digit step, intercept = fixed cost per product (first digit read, carry resolution);
//   - the workgroup barrier between the two waves of a pair;
//   - dependent chains of the instructions a step is made of (v_mad_u64_u32 -> v_mad_u64_u32, through an SGPR, through a DPP
//     lane shift), so that a change to the step can be priced before it is written.
// Build + run: tools/run_latency_probe.sh (hipcc --offload-arch=gfx950); prints one JSON object.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>

#include "../python-paillier_amd/csrc/wave_gfx950.h"
#include "../python-paillier_amd/csrc/mont_core.h"
#include "../python-paillier_amd/csrc/mul_io.h"
#include "../python-paillier_amd/csrc/split_core.h"

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                               \
        }                                                           \
    } while (0)

using namespace phe;

// one wave: `reps` first-word sweeps over `rows` digits, each feeding the next
template <int L>
__global__ void __launch_bounds__(64) k_first(uint32_t* out, unsigned long long* ticks, int rows, int reps, uint32_t seed) {
    constexpr int H = 64 * L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * H + 4 * 64 + 32];
    const uint32_t lane = threadIdx.x;
    const Lanes<64> ln(lane);
    uint32_t w[L], nbar[L];
    for (int k = 0; k < L; ++k) {
        const bool in = (int)(lane * L + k) < rows;
        w[k] = in ? ((seed * 2654435761u + lane * 97u + k) & kLimbMask) : 0u;
        nbar[k] = in ? ((seed * 40503u + lane * 131u + k * 7u) & kLimbMask) : 0u;
    }
    lds_put<L>(lds, w, lane);
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        lds_put<L>(lds, w, lane);
        ab_first_word<L>(w, lds, w, lds + H, lds + 2 * H, nbar, ln, rows);
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) ticks[0] = t1 - t0;
    for (int k = 0; k < L; ++k) out[lane * L + k] = w[k];
}

template <int L, bool MUL>
__global__ void __launch_bounds__(64) k_second(uint32_t* out, unsigned long long* ticks, int rows, int reps, uint32_t seed) {
    constexpr int H = 64 * L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[5 * H + 64];
    const uint32_t lane = threadIdx.x;
    const Lanes<64> ln(lane);
    uint32_t w[L], v[L], nbar[L];
    for (int k = 0; k < L; ++k) lds[3 * H + lane * L + k] = 0u;
    if (lane < 48u) lds[4 * H + lane] = 0u;
    for (int k = 0; k < L; ++k) {
        const bool in = (int)(lane * L + k) < rows;
        w[k] = in ? ((seed * 2654435761u + lane * 97u + k) & kLimbMask) : 0u;
        v[k] = in ? ((seed * 69069u + lane * 17u + k) & kLimbMask) : 0u;
        nbar[k] = in ? ((seed * 40503u + lane * 131u + k * 7u) & kLimbMask) : 0u;
    }
    lds_put<L>(lds, w, lane);          // a
    lds_put<L>(lds + H, v, lane);      // quotient digits of the first word
    lds_put<L>(lds + 2 * H, v, lane);  // c
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        lds_put<L>(lds + 2 * H, w, lane);
        ab_second_word<L, MUL>(w, lds, lds + 2 * H, lds + H, lds + 3 * H, v, w, nbar, ln, rows);
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) ticks[0] = t1 - t0;
    for (int k = 0; k < L; ++k) out[lane * L + k] = w[k];
}

// the wave pair as the kernel runs it: wave 0 = first words + publish, wave 1 = second words one product behind, one
// workgroup barrier per product (a chain of squarings); reports ticks per product and the SIMDs the two waves sit on
template <int L>
__global__ void __launch_bounds__(128) k_pair(uint32_t* out, unsigned long long* ticks, int rows, int reps, uint32_t seed) {
    constexpr int H = 64 * L;
    __shared__ __attribute__((aligned(16))) uint32_t lds[ab_lds_words<L>()];
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const Lanes<64> ln(lane);
    uint32_t w[L], nbar[L];
    for (int k = 0; k < L; ++k) {
        const bool in = (int)(lane * L + k) < rows;
        w[k] = in ? ((seed * 2654435761u + lane * 97u + k + role) & kLimbMask) : 0u;
        nbar[k] = in ? ((seed * 40503u + lane * 131u + k * 7u) & kLimbMask) : 0u;
    }
    uint32_t* dump = lds + 6 * H;
    uint32_t* zeros = dump + 4 * 64 + H + 16;
    if (role) {
        for (int k = 0; k < L; ++k) zeros[lane * L + k] = 0u;
        if (lane < 48u) zeros[H + lane] = 0u;
    }
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        uint32_t* slot = lds + (r & 1) * 2 * H;
        if (role == 0u) {
            lds_put<L>(slot, w, lane);
            ab_first_word<L>(w, slot, w, slot + H, dump, nbar, ln, rows);
            wave::block_barrier();
        } else {
            uint32_t d[L];
            for (int k = 0; k < L; ++k) d[k] = w[k];
            add_normalize<64, L>(d, w, ln);
            wave::block_barrier();
            ab_second_word<L, false>(w, slot, nullptr, slot + H, zeros, d, d, nbar, ln, rows);
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) {
        ticks[role] = t1 - t0;
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        ticks[2 + role] = hwid;
    }
    for (int k = 0; k < L; ++k) out[threadIdx.x * L + k] = w[k];
}

// two waves, `reps` workgroup barriers (the wave-pair kernels' one barrier per product)
__global__ void __launch_bounds__(128) k_barrier(uint32_t* out, unsigned long long* ticks, int reps) {
    __shared__ uint32_t lds[128];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = clock64();
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        lds[threadIdx.x] = acc + r;
        wave::block_barrier();
        acc += lds[threadIdx.x ^ 64u];
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = acc;
}

// dependent chains, one wave.  KIND: 0 mad->mad; 1 mad -> readfirstlane -> s_and -> mad (SGPR operand); 2 mad -> dpp shift -> mad;
// 3 v_and -> readfirstlane -> (SGPR multiplier) mad, with the dpp shift + alignbit + add in its shadow (the L = 1 step);
// 4 v_mul_lo chain; 5 alignbit+add chain; 6 lshr64 + add64 chain; 7 LDS write -> read round trip
template <int KIND>
__global__ void __launch_bounds__(64) k_chain(uint32_t* out, unsigned long long* ticks, int reps, uint32_t seed) {
    __shared__ uint32_t lds[256];
    const uint32_t lane = threadIdx.x;
    uint64_t x = ((uint64_t)(seed | 1u) << 20) | lane;
    const uint32_t b = (seed * 2654435761u + lane) | 1u, c = (seed ^ (lane * 40503u)) | 1u;
    lds[lane] = lane;
    const unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (KIND == 0) {
                x = wave::mad64((uint32_t)x, b, x);
            } else if constexpr (KIND == 1) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) & kLimbMask;
                x = wave::mad64(m, b, x);
            } else if constexpr (KIND == 2) {
                const uint32_t d = wave::grp_down1_raw<64>((uint32_t)x);
                x = wave::mad64(d, b, x);
            } else if constexpr (KIND == 3) {
                const uint32_t t = wave::reread((uint32_t)x & kLimbMask);
                const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                const uint32_t sum = wave::grp_down1_raw<64>(t) + (uint32_t)(x >> kRadixBits);
                x = wave::mad64(c, b, (uint64_t)sum);
                x = wave::mad64(m, c, x);
            } else if constexpr (KIND == 4) {
                uint32_t y = (uint32_t)x;
                asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y) : "v"(b));
                x = y;
            } else if constexpr (KIND == 5) {
                uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
                asm volatile("v_alignbit_b32 %0, %1, %0, 29\n\tv_add_u32 %0, %0, %2" : "+v"(lo) : "v"(hi), "v"(b));
                x = ((uint64_t)hi << 32) | lo;
            } else if constexpr (KIND == 6) {
                x = (x >> 29) + (uint64_t)b;
                asm volatile("" : "+v"(x));
            } else {
                lds[lane] = (uint32_t)x;
                wave::lds_fence();
                x += lds[(lane + 1) & 63u];
                wave::lds_fence();
            }
        }
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) ticks[0] = t1 - t0;
    out[lane] = (uint32_t)(x ^ (x >> 32));
}

// The whole chip saturated (8 waves per SIMD on every CU) with one instruction kind: the s_memtime ticks a wave sees against the
// HIP-event time of the launch = the shader clock the chip holds UNDER THAT LOAD, and ticks per wave-instruction per SIMD = the
// issue cost of the instruction in real cycles, whatever that clock is.  KIND 0: v_mad_u64_u32, 1: v_fma_f32, 2: v_pk_fma_f32.
template <int KIND>
__global__ void __launch_bounds__(1024) k_saturate(uint32_t* out, unsigned long long* ticks, int iters, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t aa = seed * 2654435761u + tid, bb = (seed ^ tid) | 1u;
    uint64_t x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        x[i] = ((uint64_t)bb << 32) | (aa + i);
        f[i] = (float)(aa + i) * 1e-9f;
    }
    const float fa = 1.0000001f, fb = 1e-7f;
    uint64_t pa = 0x3f8000013f800001ull, pb = 0x33d6bf9533d6bf95ull;  // (1.0000001f, 1.0000001f), (1e-7f, 1e-7f)
    asm volatile("" : "+v"(pa), "+v"(pb));
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(aa), "v"(bb) : "vcc");
                else if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fa), "v"(fb));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(pa), "v"(pb));  // two fp32 FMAs per lane
            }
    }
    const unsigned long long t1 = clock64();
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) ticks[0] = t1 - t0;
    uint64_t r = 0;
    float g = 0;
    for (int i = 0; i < 8; ++i) {
        r ^= x[i];
        g += f[i];
    }
    if ((uint32_t)(r ^ (r >> 32)) == 0x12345678u || g == 1.2345f) out[tid & 1023u] = (uint32_t)r;
}

template <typename F>
static int timed(F launch, unsigned long long* dt, double& cycles) {
    unsigned long long tk = 0;
    launch();
    CK(hipDeviceSynchronize());
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&tk, dt, 8, hipMemcpyDeviceToHost));
    cycles = (double)tk;
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "--burn") {
        // keep the whole chip's VALU saturated for a few seconds with one instruction kind (tools/sample_clock_under_load.sh
        // reads the shader clock and the power beside it): --burn mad|fma [seconds]
        uint32_t* d = nullptr;
        unsigned long long* dt = nullptr;
        CK(hipMalloc((void**)&d, 1 << 16));
        CK(hipMalloc((void**)&dt, 64));
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        const std::string kind = argv[2];
        const bool mad = kind == "mad";
        const double seconds = argc >= 4 ? atof(argv[3]) : 5.0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        double total_ms = 0, instr = 0;
        while (total_ms < seconds * 1e3) {
            float ms = 0;
            CK(hipEventRecord(e0));
            for (int k = 0; k < 16; ++k) {
                if (mad) k_saturate<0><<<prop.multiProcessorCount * 2, 1024>>>(d, dt, 8192, 11u);
                else if (kind == "pkfma") k_saturate<2><<<prop.multiProcessorCount * 2, 1024>>>(d, dt, 8192, 11u);
                else k_saturate<1><<<prop.multiProcessorCount * 2, 1024>>>(d, dt, 8192, 11u);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            instr += 16.0 * 8192 * 64 * 32.0 * prop.multiProcessorCount;
        }
        printf("{\"burn\": \"%s\", \"seconds\": %.2f, \"wave_instr_per_s\": %.4g, \"lane_ops_per_s\": %.4g}\n", argv[2], total_ms * 1e-3,
               instr / (total_ms * 1e-3), 64.0 * instr / (total_ms * 1e-3));
        return 0;
    }
    uint32_t* d = nullptr;
    unsigned long long* dt = nullptr;
    CK(hipMalloc((void**)&d, 1 << 16));
    CK(hipMalloc((void**)&dt, 64));
    const int reps = 400;
    double cy = 0;
    printf("{\"unit\": \"s_memtime ticks (shader cycles) of a lone wavefront\", \"first_word\": {");
    const int rows1[] = {20, 37, 40, 56}, rows2[] = {72, 80, 112}, rows3[] = {144, 180};
    bool sep = false;
#define SWEEP(KERN, LL, ROWS)                                                                   \
    for (int rows : ROWS) {                                                                    \
        if (timed([&] { KERN<<<1, 64>>>(d, dt, rows, reps, 7u); }, dt, cy)) return 1;          \
        printf("%s\"L%d_rows%d\": %.1f", sep ? ", " : "", LL, rows, cy / reps);                \
        sep = true;                                                                            \
    }
    SWEEP(k_first<1>, 1, rows1)
    SWEEP(k_first<2>, 2, rows2)
    SWEEP(k_first<3>, 3, rows3)
    printf("}, \"second_word_square\": {");
    sep = false;
    SWEEP((k_second<1, false>), 1, rows1)
    SWEEP((k_second<2, false>), 2, rows2)
    SWEEP((k_second<3, false>), 3, rows3)
    printf("}, \"second_word_multiply\": {");
    sep = false;
    SWEEP((k_second<1, true>), 1, rows1)
    SWEEP((k_second<2, true>), 2, rows2)
    printf("}, ");
    {   // the chip saturated with one instruction kind: real cycles per wave-instruction per SIMD and the clock it holds meanwhile
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        const int cus = prop.multiProcessorCount, iters = 4096;
        const char* kinds[2] = {"v_mad_u64_u32", "v_fma_f32"};
        printf("\"saturated_chip\": {");
        for (int kind = 0; kind < 2; ++kind) {
            unsigned long long tk = 0;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0));
                if (kind == 0) k_saturate<0><<<cus * 2, 1024>>>(d, dt, iters, 11u);
                else k_saturate<1><<<cus * 2, 1024>>>(d, dt, iters, 11u);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&tk, dt, 8, hipMemcpyDeviceToHost));
            const double per_wave = (double)iters * 64;  // instructions a wave issues
            printf("%s\"%s\": {\"ticks_per_wave_instr_per_simd\": %.3f, \"ticks_per_us\": %.1f, \"kernel_ms\": %.3f, "
                   "\"wave_instr_per_s\": %.4g}", kind ? ", " : "", kinds[kind], (double)tk / per_wave / 8.0, (double)tk / (ms * 1e3), ms,
                   per_wave * 32.0 * cus / (ms * 1e-3));
        }
        printf("}, ");
    }
    {   // the shader clock a lone wavefront really gets: s_memtime ticks of one long launch against its HIP-event time
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int waves : {1, 1024}) {
            unsigned long long tk = 0;
            float ms = 0;
            k_first<1><<<waves, 64>>>(d, dt, 37, 4000, 9u);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            k_first<1><<<waves, 64>>>(d, dt, 37, 4000, 9u);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&tk, dt, 8, hipMemcpyDeviceToHost));
            printf("\"ticks_per_us_%d_workgroups\": %.1f, ", waves, (double)tk / (ms * 1e3));
        }
    }
    {
        unsigned long long tk[4];
        for (int rows : {20, 37, 56}) {
            k_pair<1><<<1, 128>>>(d, dt, rows, 400, 5u);
            CK(hipDeviceSynchronize());
            k_pair<1><<<1, 128>>>(d, dt, rows, 400, 6u);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(tk, dt, 32, hipMemcpyDeviceToHost));
            printf("\"pair_L1_rows%d\": {\"ticks_per_product\": %.1f, \"simd_of_wave0\": %llu, \"simd_of_wave1\": %llu, \"cu0\": %llu, \"cu1\": %llu}, ", rows,
                   (double)tk[0] / 400, (tk[2] >> 4) & 3, (tk[3] >> 4) & 3, (tk[2] >> 8) & 15, (tk[3] >> 8) & 15);
        }
        k_pair<3><<<1, 128>>>(d, dt, 144, 100, 6u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(tk, dt, 32, hipMemcpyDeviceToHost));
        printf("\"pair_L3_rows144\": %.1f, ", (double)tk[0] / 100);
    }
    if (timed([&] { k_barrier<<<1, 128>>>(d, dt, 2000); }, dt, cy)) return 1;
    printf("\"barrier_with_lds_exchange\": %.1f, \"chains_per_link\": {", cy / 2000);
    const char* names[8] = {"mad_mad", "mad_readfirstlane_sand_mad", "mad_dpp_mad", "step_L1_two_mads", "v_mul_lo", "alignbit_add", "lshr64_add64", "lds_write_read"};
#define CHAIN(K)                                                                               \
    if (timed([&] { k_chain<K><<<1, 64>>>(d, dt, 500, 3u); }, dt, cy)) return 1;               \
    printf("%s\"%s\": %.2f", K ? ", " : "", names[K], cy / (500.0 * 8));
    CHAIN(0) CHAIN(1) CHAIN(2) CHAIN(3) CHAIN(4) CHAIN(5) CHAIN(6) CHAIN(7)
    printf("}}\n");
    return 0;
}
