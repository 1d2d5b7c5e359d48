// tests/emu/wave_emu.h — TEST INFRASTRUCTURE: a host stand-in for csrc/wave_gfx950.h.
//
// Runs the 64 lanes of one wavefront as 64 fibers so that mont_core.h / split_core.h (the code the
// GPU executes) can be exercised on a CPU-only box.  Cross-lane primitives exchange values
// through a double-buffered mailbox and yield to the scheduler; all lanes must execute the same
// sequence of cross-lane primitives (true for these kernels: control flow is wave-uniform).
// Semantics mirror the DPP forms documented in wave_gfx950.h; tests/test_gpu_parity.py::test_wave_primitives checks
// the real instructions against the same expectations on the GPU.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define PHE_DEV inline
// the device's LDS-bounds assertions (wave_gfx950.h PHE_BOUNDS) are ALWAYS on in the emulator: an index out of its area aborts the test
#define PHE_BOUNDS(...)               \
    do {                              \
        if (!(__VA_ARGS__)) abort();  \
    } while (0)

namespace wave {

constexpr int kRow = 16;
constexpr int kLanes = 64;

// Fiber switch.  glibc's swapcontext makes a sigprocmask system call per switch, and a 2048-bit product switches
// ~10^5 times; on x86-64 the switch is therefore a dozen instructions of our own (callee-saved registers + stack
// pointer), with ucontext kept as the portable fallback.
#if defined(__x86_64__)
#define PHE_EMU_ASM_SWITCH 1
extern "C" void phe_emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl phe_emu_switch
    .type phe_emu_switch,@function
phe_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size phe_emu_switch,.-phe_emu_switch
)");
#else
#define PHE_EMU_ASM_SWITCH 0
#endif

struct Emu {
#if PHE_EMU_ASM_SWITCH
    void* sched_sp = nullptr;
    void* fib_sp[kLanes];
#else
    ucontext_t sched;
    ucontext_t fib[kLanes];
#endif
    char* stacks[kLanes];
    bool done[kLanes];
    int cur = 0;
    unsigned phase = 0;  // per-lane phase counters advance in lock step
    unsigned lane_phase[kLanes];
    uint32_t box[2][kLanes];
    std::function<void(uint32_t)> body;
};

inline Emu*& emu() {
    static thread_local Emu* e = nullptr;
    return e;
}

inline void yield_all() {
    Emu* e = emu();
#if PHE_EMU_ASM_SWITCH
    phe_emu_switch(&e->fib_sp[e->cur], e->sched_sp);
#else
    swapcontext(&e->fib[e->cur], &e->sched);
#endif
}

inline void fiber_entry() {
    Emu* e = emu();
    const int l = e->cur;
    e->body((uint32_t)l);
    e->done[l] = true;
    yield_all();
    abort();  // a finished fiber is never resumed
}

// run `body(lane)` for the 64 lanes of one wave
inline void run_wave(const std::function<void(uint32_t)>& body) {
    Emu* e = new Emu();
    Emu* prev = emu();
    emu() = e;
    e->body = body;
    const size_t kStack = 1u << 20;
    for (int l = 0; l < kLanes; ++l) {
        e->done[l] = false;
        e->lane_phase[l] = 0;
        e->stacks[l] = (char*)malloc(kStack);
#if PHE_EMU_ASM_SWITCH
        // initial frame: six zeroed callee-saved registers, then the entry point as the return address (its slot is
        // 16-byte aligned so that the entry function sees the stack alignment of a normal call)
        uintptr_t top = ((uintptr_t)e->stacks[l] + kStack) & ~(uintptr_t)15;
        void** frame = (void**)(top - 16) - 6;
        for (int i = 0; i < 6; ++i) frame[i] = nullptr;
        frame[6] = (void*)&fiber_entry;
        frame[7] = nullptr;
        e->fib_sp[l] = (void*)frame;
#else
        getcontext(&e->fib[l]);
        e->fib[l].uc_stack.ss_sp = e->stacks[l];
        e->fib[l].uc_stack.ss_size = kStack;
        e->fib[l].uc_link = &e->sched;
        makecontext(&e->fib[l], (void (*)())fiber_entry, 0);
#endif
    }
    for (;;) {
        bool any = false;
        for (int l = 0; l < kLanes; ++l) {
            if (e->done[l]) continue;
            any = true;
            e->cur = l;
#if PHE_EMU_ASM_SWITCH
            phe_emu_switch(&e->sched_sp, e->fib_sp[l]);
#else
            swapcontext(&e->sched, &e->fib[l]);
#endif
        }
        if (!any) break;
    }
    for (int l = 0; l < kLanes; ++l) free(e->stacks[l]);
    emu() = prev;
    delete e;
}

inline uint32_t lane_id() { return (uint32_t)emu()->cur; }

// post x, wait for every lane, then read lane `src` (or 0 if src < 0)
template <typename F>
inline uint32_t exchange(uint32_t x, F src_of) {
    Emu* e = emu();
    const int l = e->cur;
    const unsigned ph = e->lane_phase[l]++ & 1u;
    e->box[ph][l] = x;
    yield_all();
    const int src = src_of(l);
    return src < 0 ? 0u : e->box[ph][src];
}

template <int G>
struct Lanes {
    uint32_t lane, g, not_top, not_low;
    explicit Lanes(uint32_t lane_) : lane(lane_), g(lane_ & (G - 1)) {
        not_top = (g == G - 1) ? 0u : 0xffffffffu;
        not_low = (g == 0) ? 0u : 0xffffffffu;
    }
};
template <int G>
inline uint32_t grp_down1(uint32_t x, const Lanes<G>&) {
    return exchange(x, [](int l) { return (l & (G - 1)) == G - 1 ? -1 : l + 1; });
}
// the top lane's result is unspecified on the device: hand back junk so that a caller forgetting its mask fails
template <int G>
inline uint32_t grp_down1_raw(uint32_t x) {
    return exchange(x ^ 0u, [](int l) { return l + 1 < kLanes ? l + 1 : l; }) | 0u;
}
template <int G>
inline uint32_t grp_up1(uint32_t x, const Lanes<G>&) {
    return exchange(x, [](int l) { return (l & (G - 1)) == 0 ? -1 : l - 1; });
}
template <int G>
inline uint32_t grp_bcast0(uint32_t x, const Lanes<G>&) {
    return exchange(x, [](int l) { return l & ~(G - 1); });
}
inline uint64_t ballot(bool p) {
    Emu* e = emu();
    const int l = e->cur;
    const unsigned ph = e->lane_phase[l]++ & 1u;
    e->box[ph][l] = p ? 1u : 0u;
    yield_all();
    uint64_t m = 0;
    for (int i = 0; i < kLanes; ++i) m |= (uint64_t)(e->box[ph][i] & 1u) << i;
    return m;
}
// Workgroup barrier between waves that run in DIFFERENT host threads (run_block below): every lane of the wave reaches this
// point (one yield, as in lds_fence), then lane 0 — always the first lane to be resumed — waits for the other waves' lane 0
// on the shared pthread barrier; the remaining lanes resume only after it has passed.
inline pthread_barrier_t*& block_barrier_object() {
    static pthread_barrier_t* b = nullptr;
    return b;
}
inline void block_barrier() {
    Emu* e = emu();
    e->lane_phase[e->cur]++;
    yield_all();
    if (e->cur == 0 && block_barrier_object()) pthread_barrier_wait(block_barrier_object());
}

// every mad64 is one v_mad_u64_u32 lane-operation on the device: counted here so that tools/count_executed_mads.py can
// state EXACTLY how many multiply-adds a kernel issues per element (bench.py's roofline.executed), not a hand model
// (per host thread; run_block adds what its wave threads counted to the caller's)
inline uint64_t& mad_counter() {
    static thread_local uint64_t count = 0;
    return count;
}

// run `body(wave, lane)` for the waves of one workgroup, one host thread per wave; block_barrier() joins them
inline void run_block(int n_waves, const std::function<void(uint32_t, uint32_t)>& body) {
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, (unsigned)n_waves);
    block_barrier_object() = &bar;
    struct Arg {
        const std::function<void(uint32_t, uint32_t)>* body;
        uint32_t wave;
        uint64_t mads;
    };
    std::vector<pthread_t> th((size_t)n_waves);
    std::vector<Arg> args((size_t)n_waves);
    for (int w = 0; w < n_waves; ++w) {
        args[(size_t)w] = Arg{&body, (uint32_t)w, 0};
        pthread_create(&th[(size_t)w], nullptr, [](void* p) -> void* {
            Arg* a = (Arg*)p;
            const uint32_t wv = a->wave;
            const auto* fn = a->body;
            run_wave([&](uint32_t lane) { (*fn)(wv, lane); });
            a->mads = mad_counter();
            return nullptr;
        }, &args[(size_t)w]);
    }
    for (int w = 0; w < n_waves; ++w) {
        pthread_join(th[(size_t)w], nullptr);
        mad_counter() += args[(size_t)w].mads;
    }
    block_barrier_object() = nullptr;
    pthread_barrier_destroy(&bar);
}

inline void lds_fence() {
    Emu* e = emu();
    e->lane_phase[e->cur]++;
    yield_all();
}

// LDS-DMA stand-in: lane l's 16 bytes land at lds_wave_base + 16*l (copied at once: any completion time before the wait
// is a legal behaviour of the device instruction)
inline void async_copy16_to_lds(const uint32_t* gsrc, uint32_t* lds_wave_base, bool active) {
    if (active) {
        uint32_t* dst = lds_wave_base + 4 * lane_id();
        for (int i = 0; i < 4; ++i) dst[i] = gsrc[i];
    }
}
inline void wait_async_copies() { lds_fence(); }

template <int N>
struct ScalarRow {
    uint32_t w[N];
    void request(const uint32_t* p) {
        for (int k = 0; k < N; ++k) w[k] = p[k];
    }
    template <int OFFW>
    void request_at(const uint32_t* p) { request(p + OFFW); }
    uint32_t word(int k) const { return w[k]; }
};
struct DigitPair {
    uint32_t w[2];
    template <int ROW>
    void request(const uint32_t* lds_column) {
        w[0] = lds_column[ROW * 64];
        w[1] = lds_column[(ROW + 1) * 64];
    }
    uint32_t word(int u) const { return w[u]; }
};
template <int N, int GD>
inline void arrived(ScalarRow<N> (&)[GD], DigitPair (&)[GD / 2]) {}
inline void order_fence() {}
inline void set_priority(int) {}
inline void load16(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d, const uint32_t* p) { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
inline uint32_t uniform(uint32_t x) { return x; }
template <int N>
inline void scalar_words(uint32_t (&c)[N], const uint32_t* p) {
    for (int k = 0; k < N; ++k) c[k] = p[k];
}
inline const uint32_t* reread_ptr(const uint32_t* p) { return p; }
template <class T>
inline T* reread_vptr(T* p) { return p; }
inline uint32_t reread(uint32_t x) { return x; }  // see wave_gfx950.h: an optimisation barrier on the device, nothing here
inline uint64_t reread64(uint64_t x) { return x; }
inline void lds_or(uint32_t* p, uint32_t v) { __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }  // (the waves of a workgroup are host threads)
typedef uint32_t lds_u32;  // wave_gfx950.h: an LDS-address-space pointer on the device
inline lds_u32* as_lds(uint32_t* p) { return p; }
inline lds_u32* reread_lds(uint32_t* p) { return p; }
inline const lds_u32* reread_lds(const uint32_t* p) { return p; }
inline void lds_store2(lds_u32* p, uint32_t a, uint32_t b) { p[0] = a; p[1] = b; }
inline void lds_store4(lds_u32* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }

inline uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
    ++mad_counter();
    return (uint64_t)a * b + c;
}

}  // namespace wave
