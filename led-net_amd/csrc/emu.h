// emu.h -- CPU emulation of the HIP execution model.  TEST INFRASTRUCTURE ONLY.
//
// Compiled only into tests/emu/libledn_emu.so (-DLEDN_CPU_EMU, host clang, ASan
// friendly).  It lets the kernel sources under csrc/ run on the CPU of a box
// without a GPU so that indexing, barriers, wave shuffles and MFMA fragment
// layouts can be debugged (and sanitised) before GPU minutes are spent.  The
// product library libledn_hip.so never contains this file, and the product
// package only ever loads libledn_hip.so.
//
// Model: every HIP thread of a workgroup is a cooperative fiber of one OS thread;
// workgroups run one after another (so `__shared__` maps to `static`);
// __syncthreads() and the wave collectives (shuffles, ballot, MFMA) are
// rendez-vous points at which a fiber yields.  Wave = 64.  Atomics are plain
// read-modify-writes (single OS thread).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <map>
#include <string>
#include <vector>

#ifdef LEDN_EMU_DEFINE_SWITCH
namespace ledn_emu_census {
static std::map<std::pair<std::string, int>, long>* g_sites = nullptr;
static int g_on = -1;
}
extern "C" void ledn_emu_atomic_note(const char* file, int line) {
    using namespace ledn_emu_census;
    if (g_on < 0) g_on = getenv("LEDN_EMU_ATOMIC_CENSUS") ? 1 : 0;
    if (!g_on) return;
    if (!g_sites) g_sites = new std::map<std::pair<std::string, int>, long>();
    const char* b = strrchr(file, '/');
    ++(*g_sites)[{std::string(b ? b + 1 : file), line}];
}
// (python: lib.cdll.ledn_emu_atomic_census(path, reset)) writes "file:line count" lines and optionally clears the table
extern "C" int ledn_emu_atomic_census(const char* path, int reset) {
    using namespace ledn_emu_census;
    g_on = 1;
    FILE* f = path ? fopen(path, "w") : nullptr;
    int n = 0;
    if (g_sites) {
        for (auto& kv : *g_sites) {
            if (f) fprintf(f, "%s:%d %ld\n", kv.first.first.c_str(), kv.first.second, kv.second);
            ++n;
        }
        if (reset) g_sites->clear();
    }
    if (f) fclose(f);
    return n;
}
#endif

namespace emu {

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

// Cooperative fibers: all HIP threads of a workgroup are fibers of ONE OS thread,
// resumed round-robin; a barrier is "yield until everybody alive has arrived".
// Deterministic, no locks, ~10 ns per switch (hand-rolled x86-64 context switch).
extern "C" void ledn_emu_switch(void** save_sp, void* load_sp);
#ifdef LEDN_EMU_DEFINE_SWITCH
__asm__(R"(
.text
.globl ledn_emu_switch
.type ledn_emu_switch,@function
ledn_emu_switch:
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
.size ledn_emu_switch,.-ledn_emu_switch
)");
#endif

struct Barrier {
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    void reset(int n) { alive = n; arrived = 0; }
};

struct Ctx {
    dim3 tid, bid, bdim, gdim;
    int lane = 0, wave = 0;
    Barrier* block_bar = nullptr;
    Barrier* wave_bar = nullptr;
    unsigned char* xchg = nullptr;  // this wave's exchange area: 64 lanes x 64 B
    unsigned char* mx = nullptr;    // matrix-instruction operands: TWO such areas, used alternately (one barrier per MFMA)
    int mphase = 0;
    void* sp = nullptr;             // saved stack pointer of this fiber
    bool done = false;
};

struct Sched {
    void* main_sp = nullptr;
    Ctx* cur = nullptr;
    void (*entry)(void*) = nullptr;
    void* entry_arg = nullptr;
};
inline Sched g_sched;
inline unsigned char* g_dyn_smem = nullptr;
#define LEDN_EMU_CUR (emu::g_sched.cur)

inline void yield_to_main() { ledn_emu_switch(&g_sched.cur->sp, g_sched.main_sp); }

inline void barrier_wait(Barrier* b) {
    const unsigned g = b->gen;
    if (++b->arrived >= b->alive) {
        b->arrived = 0;
        ++b->gen;
        return;
    }
    while (b->gen == g) yield_to_main();
}
inline void barrier_drop(Barrier* b) {
    --b->alive;
    if (b->alive > 0 && b->arrived >= b->alive) {
        b->arrived = 0;
        ++b->gen;
    }
}

inline void fiber_trampoline() {
    Ctx* c = g_sched.cur;
    g_sched.entry(g_sched.entry_arg);
    barrier_drop(c->wave_bar);
    barrier_drop(c->block_bar);
    c->done = true;
    yield_to_main();
    std::abort();  // never resumed
}

template <class F>
void launch(dim3 grid, dim3 block, size_t dyn_smem, F body) {
    const int nt = block.x * block.y * block.z;
    const int nw = (nt + 63) / 64;
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nt <= 0 || nblocks <= 0) return;
    constexpr size_t STACK = 96 * 1024;
    Barrier block_bar;
    std::vector<Barrier> wave_bars(nw);
    std::vector<unsigned char> xchg((size_t)nw * 64 * 64 * 3);
    std::vector<unsigned char> smem(dyn_smem + 64);
    std::vector<unsigned char> stacks((size_t)nt * STACK + 64);
    std::vector<Ctx> fibers(nt);
    g_dyn_smem = smem.data();
    F* bodyp = &body;
    g_sched.entry = [](void* p) { (*static_cast<F*>(p))(); };
    g_sched.entry_arg = bodyp;
    for (long b = 0; b < nblocks; ++b) {
        block_bar.reset(nt);
        for (int w = 0; w < nw; ++w) wave_bars[w].reset(std::min(64, nt - w * 64));
        for (int t = 0; t < nt; ++t) {
            Ctx& c = fibers[t];
            c.bdim = block;
            c.gdim = grid;
            c.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            c.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / ((long)grid.x * grid.y));
            c.lane = t % 64;
            c.wave = t / 64;
            c.block_bar = &block_bar;
            c.wave_bar = &wave_bars[c.wave];
            c.xchg = xchg.data() + (size_t)c.wave * 64 * 64 * 3;
            c.mx = c.xchg + 64 * 64;
            c.mphase = 0;
            c.done = false;
            // initial frame: 6 callee-saved slots + return address (16-B ABI alignment at entry)
            uintptr_t top = (uintptr_t)(stacks.data() + (size_t)(t + 1) * STACK);
            top &= ~(uintptr_t)15;
            void** sp = (void**)(top - 8);
            *--sp = (void*)&fiber_trampoline;
            for (int i = 0; i < 6; ++i) *--sp = nullptr;
            c.sp = sp;
        }
        int remaining = nt;
        while (remaining > 0) {
            for (int t = 0; t < nt; ++t) {
                Ctx& c = fibers[t];
                if (c.done) continue;
                g_sched.cur = &c;
                ledn_emu_switch(&g_sched.main_sp, c.sp);
                if (c.done) --remaining;
            }
        }
    }
    g_sched.cur = nullptr;
    g_dyn_smem = nullptr;
}

// ---- wave collectives ------------------------------------------------------
template <class T>
inline T shfl_from(T v, int src) {
    static_assert(sizeof(T) <= 64, "exchange slot is 64 B");
    Ctx& c = *g_sched.cur;
    std::memcpy(c.xchg + c.lane * 64, &v, sizeof(T));
    barrier_wait(c.wave_bar);
    T r;
    std::memcpy(&r, c.xchg + (src & 63) * 64, sizeof(T));
    barrier_wait(c.wave_bar);
    return r;
}

}  // namespace emu

// ---- HIP surface -------------------------------------------------------------
using emu::dim3;
using std::min;
using std::max;
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define threadIdx (LEDN_EMU_CUR->tid)
#define blockIdx (LEDN_EMU_CUR->bid)
#define blockDim (LEDN_EMU_CUR->bdim)
#define gridDim (LEDN_EMU_CUR->gdim)
#define LEDN_DYN_SMEM (emu::g_dyn_smem)

inline void __syncthreads() { emu::barrier_wait(LEDN_EMU_CUR->block_bar); }
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    std::memset(p, v, n);
    return 0;
}

template <class T> inline T __shfl(T v, int src, int = 64) { return emu::shfl_from(v, src); }
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return emu::shfl_from(v, LEDN_EMU_CUR->lane ^ m); }
template <class T> inline T __shfl_down(T v, int d, int = 64) {
    int s = LEDN_EMU_CUR->lane + d;
    return emu::shfl_from(v, s < 64 ? s : LEDN_EMU_CUR->lane);
}
inline unsigned long long __ballot(int pred) {
    unsigned long long r = 0;
    for (int l = 0; l < 64; ++l) {  // 64 rounds: simple, test-only
        int p = emu::shfl_from(pred, l);
        if (p) r |= 1ull << l;
    }
    return r;
}

inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __fdividef(float a, float b) { return a / b; }

// census of the FLOAT atomic sites a run goes through (LEDN_EMU_ATOMIC_CENSUS=<file>: "source:line count" per site at
// exit): how tests/test_deterministic.py proves that the deterministic mode reaches no float atomic at all
extern "C" void ledn_emu_atomic_note(const char* file, int line);
inline float atomicAdd(float* p, float v, const char* file = __builtin_FILE(), int line = __builtin_LINE()) {
    ledn_emu_atomic_note(file, line);
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do {
        f = __uint_as_float(old) + v;
        nw = __float_as_uint(f);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return __uint_as_float(old);
}
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline unsigned atomicMin(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---- vector types + MFMA (layouts: cdna_hip_programming.md section 3) -------
typedef short emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

namespace emu {
inline float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// v_mfma_f32_32x32x16_bf16: lane l (r=l&31,h=l>>5) holds A[r][8h+j], B[8h+j][r];
// D: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5).
inline emu_f32x16 mfma_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
    Ctx& cx = *g_sched.cur;
    unsigned char* const X = cx.mx + (size_t)cx.mphase * 64 * 64;     // alternate areas: a lane that runs ahead into the
    cx.mphase ^= 1;                                                   // next MFMA writes the OTHER area (no trailing barrier)
    std::memcpy(X + cx.lane * 64, &a, 16);
    std::memcpy(X + cx.lane * 64 + 16, &b, 16);
    barrier_wait(cx.wave_bar);
    const int col = cx.lane & 31, hi = cx.lane >> 5;
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
        float acc = c[reg];
        for (int k = 0; k < 16; ++k) {
            unsigned short av, bv;
            std::memcpy(&av, X + (row + 32 * (k >> 3)) * 64 + 2 * (k & 7), 2);
            std::memcpy(&bv, X + (col + 32 * (k >> 3)) * 64 + 16 + 2 * (k & 7), 2);
            acc = fmaf(bf16_bits_to_f32(av), bf16_bits_to_f32(bv), acc);
        }
        c[reg] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x32_bf16: lane l holds A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15];
// D: col=l&15, row=(l>>4)*4+reg.
inline emu_f32x4 mfma_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
    Ctx& cx = *g_sched.cur;
    unsigned char* const X = cx.mx + (size_t)cx.mphase * 64 * 64;     // alternate areas: a lane that runs ahead into the
    cx.mphase ^= 1;                                                   // next MFMA writes the OTHER area (no trailing barrier)
    std::memcpy(X + cx.lane * 64, &a, 16);
    std::memcpy(X + cx.lane * 64 + 16, &b, 16);
    barrier_wait(cx.wave_bar);
    const int col = cx.lane & 15, q = cx.lane >> 4;
    for (int reg = 0; reg < 4; ++reg) {
        const int row = q * 4 + reg;
        float acc = c[reg];
        for (int k = 0; k < 32; ++k) {
            unsigned short av, bv;
            std::memcpy(&av, X + (row + 16 * (k >> 3)) * 64 + 2 * (k & 7), 2);
            std::memcpy(&bv, X + (col + 16 * (k >> 3)) * 64 + 16 + 2 * (k & 7), 2);
            acc = fmaf(bf16_bits_to_f32(av), bf16_bits_to_f32(bv), acc);
        }
        c[reg] = acc;
    }
    return c;
}

// v_mfma_f32_32x32x2_f32: lane l holds A[l&31][l>>5], B[l>>5][l&31]; D as 32x32 above.
inline emu_f32x16 mfma_32x32x2_f32(float a, float b, emu_f32x16 c) {
    Ctx& cx = *g_sched.cur;
    unsigned char* const X = cx.mx + (size_t)cx.mphase * 64 * 64;     // alternate areas: a lane that runs ahead into the
    cx.mphase ^= 1;                                                   // next MFMA writes the OTHER area (no trailing barrier)
    std::memcpy(X + cx.lane * 64, &a, 4);
    std::memcpy(X + cx.lane * 64 + 4, &b, 4);
    barrier_wait(cx.wave_bar);
    const int col = cx.lane & 31, hi = cx.lane >> 5;
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
        float acc = c[reg];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            std::memcpy(&av, X + (row + 32 * k) * 64, 4);
            std::memcpy(&bv, X + (col + 32 * k) * 64 + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        c[reg] = acc;
    }
    return c;
}
}  // namespace emu

typedef short emu_s4 __attribute__((ext_vector_type(4)));
namespace emu {
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group, lane 4q+p supplies the
// address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows (row q in element q).
inline emu_s4 lds_read_tr16(const void* p) {
    Ctx& cx = *g_sched.cur;
    std::memcpy(cx.xchg + cx.lane * 64, &p, sizeof(p));
    barrier_wait(cx.wave_bar);
    const int g = cx.lane >> 4, i = cx.lane & 15;
    emu_s4 r;
    for (int q = 0; q < 4; ++q) {
        const unsigned char* base;
        std::memcpy(&base, cx.xchg + (g * 16 + 4 * q + (i >> 2)) * 64, sizeof(base));
        short v;
        std::memcpy(&v, base + (i & 3) * 2, 2);
        r[q] = v;
    }
    barrier_wait(cx.wave_bar);
    return r;
}
}  // namespace emu

#define LEDN_LAUNCH(kernel, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), (smem), [=] { kernel(__VA_ARGS__); })
