// HIP-on-CPU SIMT emulator.  TEST INFRASTRUCTURE ONLY (never linked into the product
// library): lets the build container — which has no GPU — execute the very same
// kernel sources that hipcc compiles for gfx950, so wave-level logic (MFMA fragment
// layouts, cross-lane shuffles, LDS tiling, barriers) is checked against the oracle
// before GPU minutes are spent.  Each GPU thread is a ucontext fiber; blocks run one
// after another; wave- and block-collectives rendezvous through exchange buffers.
//
// MFMA lane maps follow /opt/skills/guides/cdna_hip_programming.md §3 ("Fragment
// layout"); the first GPU run re-validates them on hardware (tests/test_gpu_kernels.py).
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
#if defined(__x86_64__)
#define EMU_FAST_SWITCH 1
// Minimal SysV x86-64 context switch (callee-saved registers + stack pointer).  glibc's swapcontext makes two
// rt_sigprocmask system calls per switch, which dominated the emulator's run time.
__attribute__((naked, noinline)) static void emu_switch(void** /*save_sp*/, void* /*load_sp*/) {
    asm volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret");
}
#else
#define EMU_FAST_SWITCH 0
#endif

struct Fiber {
    ucontext_t ctx;
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
};
struct State {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
    ucontext_t sched;
    void* sched_sp = nullptr;
    std::vector<Fiber> fibers;
    int cur = -1;
    int nthreads = 0;
    // block barrier
    int blk_arrived = 0, blk_alive = 0;
    unsigned blk_gen = 0;
    // per-wave rendezvous
    int wave_arrived[32], wave_alive[32];
    unsigned wave_gen[32];
    alignas(16) unsigned char xbuf[32][64][256];
    std::function<void()> body;
};
inline State& S() { static State s; return s; }

inline void yield() {
    State& s = S();
    Fiber& f = s.fibers[s.cur];
#if EMU_FAST_SWITCH
    emu_switch(&f.sp, s.sched_sp);
#else
    swapcontext(&f.ctx, &s.sched);
#endif
}
inline int lane_id() { State& s = S(); return s.cur & 63; }
inline int wave_id() { State& s = S(); return s.cur >> 6; }

inline void block_barrier() {
    State& s = S();
    unsigned g = s.blk_gen;
    if (++s.blk_arrived >= s.blk_alive) { s.blk_arrived = 0; s.blk_gen++; return; }
    while (s.blk_gen == g) yield();
}
inline void wave_barrier() {
    State& s = S();
    int w = wave_id();
    unsigned g = s.wave_gen[w];
    if (++s.wave_arrived[w] >= s.wave_alive[w]) { s.wave_arrived[w] = 0; s.wave_gen[w]++; return; }
    while (s.wave_gen[w] == g) yield();
}
inline void on_exit_thread() {
    State& s = S();
    int w = wave_id();
    s.blk_alive--; s.wave_alive[w]--;
    if (s.blk_alive > 0 && s.blk_arrived >= s.blk_alive) { s.blk_arrived = 0; s.blk_gen++; }
    if (s.wave_alive[w] > 0 && s.wave_arrived[w] >= s.wave_alive[w]) { s.wave_arrived[w] = 0; s.wave_gen[w]++; }
}
inline void trampoline() {
    State& s = S();
    s.body();
    s.fibers[s.cur].done = true;
    on_exit_thread();
#if EMU_FAST_SWITCH
    emu_switch(&s.fibers[s.cur].sp, s.sched_sp);
    __builtin_trap();                 // a finished fiber is never resumed
#else
    swapcontext(&s.fibers[s.cur].ctx, &s.sched);
#endif
}

template <class F>
void launch(dim3 grid, dim3 block, F&& f) {
    State& s = S();
    const int nt = block.x * block.y * block.z;
    const size_t STK = 256 * 1024;
    s.blockDim = block; s.gridDim = grid; s.nthreads = nt;
    s.body = f;
    if ((int)s.fibers.size() < nt) {
        size_t old = s.fibers.size();
        s.fibers.resize(nt);
        for (size_t i = old; i < (size_t)nt; i++) s.fibers[i].stack = (char*)malloc(STK);
    }
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        s.blockIdx = dim3(bx, by, bz);
        s.blk_arrived = 0; s.blk_alive = nt; s.blk_gen = 0;
        for (int w = 0; w < 32; w++) { s.wave_arrived[w] = 0; s.wave_gen[w] = 0; s.wave_alive[w] = 0; }
        for (int t = 0; t < nt; t++) {
            Fiber& fb = s.fibers[t];
            fb.done = false;
            fb.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            s.wave_alive[t >> 6]++;
#if EMU_FAST_SWITCH
            // initial frame: six callee-saved slots, then the entry point as emu_switch's return address; the stack is
            // 16-byte aligned + 8 at trampoline's first instruction, as after a call
            void** top = (void**)(((uintptr_t)fb.stack + STK) & ~(uintptr_t)15);
            *--top = nullptr;                            // fake return address of trampoline (never used)
            *--top = (void*)trampoline;                  // emu_switch's `ret` lands here
            for (int r = 0; r < 6; r++) *--top = nullptr;
            fb.sp = top;
#else
            getcontext(&fb.ctx);
            fb.ctx.uc_stack.ss_sp = fb.stack;
            fb.ctx.uc_stack.ss_size = STK;
            fb.ctx.uc_link = nullptr;
            makecontext(&fb.ctx, (void (*)())trampoline, 0);
#endif
        }
        int remaining = nt;
        while (remaining > 0) {
            for (int t = 0; t < nt; t++) {
                Fiber& fb = s.fibers[t];
                if (fb.done) continue;
                s.cur = t; s.threadIdx = fb.tid;
#if EMU_FAST_SWITCH
                emu_switch(&s.sched_sp, fb.sp);
#else
                swapcontext(&s.sched, &fb.ctx);
#endif
                if (fb.done) remaining--;
            }
        }
    }
}

// generic wave exchange: every lane publishes `n` bytes, then reads any lane's
template <class T>
inline T shfl(T v, int src) {
    State& s = S();
    int w = wave_id(), l = lane_id();
    memcpy(s.xbuf[w][l], &v, sizeof(T));
    wave_barrier();
    T r; memcpy(&r, s.xbuf[w][src & 63], sizeof(T));
    wave_barrier();
    return r;
}
}  // namespace emu

#define threadIdx (emu::S().threadIdx)
#define blockIdx (emu::S().blockIdx)
#define blockDim (emu::S().blockDim)
#define gridDim (emu::S().gridDim)

inline void __syncthreads() { emu::block_barrier(); }
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { (void)width; return emu::shfl(v, emu::lane_id() ^ m); }
// wave vote: true in every lane if the predicate holds in any lane
inline int __any(int pred) {
    emu::State& s = emu::S();
    int w = emu::wave_id(), l = emu::lane_id();
    int v = pred != 0;
    memcpy(s.xbuf[w][l], &v, sizeof(int));
    emu::wave_barrier();
    int r = 0;
    for (int i = 0; i < 64; i++) { int t; memcpy(&t, s.xbuf[w][i], sizeof(int)); r |= t; }
    emu::wave_barrier();
    return r;
}
template <class T> inline T __shfl(T v, int src, int width = 64) { (void)width; return emu::shfl(v, src); }
template <class T> inline T __shfl_down(T v, int d, int width = 64) { (void)width; int l = emu::lane_id(); return emu::shfl(v, (l + d) < 64 ? l + d : l); }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline float __expf(float x) { return expf(x); }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __logf(float x) { return logf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }

typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef short emu_s16x8 __attribute__((ext_vector_type(8)));

namespace emu {
inline float bf16_to_f32(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// D = A(16x32) * B(32x16) + C.   lane l: A[row l&15][k 8*(l>>4)+j], B[k 8*(l>>4)+j][col l&15],
// C/D[row 4*(l>>4)+r][col l&15]
template <float (*DEC)(unsigned short)> inline emu_f32x4 mfma_16x16x32_b16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c) {
    State& s = S();
    int w = wave_id(), l = lane_id();
    unsigned char* mine = s.xbuf[w][l];
    memcpy(mine, &a, 16); memcpy(mine + 16, &b, 16);
    wave_barrier();
    emu_f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; r++) {
        const int row = 4 * (l >> 4) + r;
        float acc = d[r];
        for (int g = 0; g < 4; g++) {
            const unsigned short* pa = (const unsigned short*)(s.xbuf[w][g * 16 + row]);
            const unsigned short* pb = (const unsigned short*)(s.xbuf[w][g * 16 + col] + 16);
            for (int j = 0; j < 8; j++) acc = fmaf(DEC(pa[j]), DEC(pb[j]), acc);
        }
        d[r] = acc;
    }
    wave_barrier();
    return d;
}
// D = A(16x4) * B(4x16) + C, f32 inputs.  lane l: A[l&15][l>>4], B[l>>4][l&15]
inline emu_f32x4 mfma_16x16x4_f32(float a, float b, emu_f32x4 c) {
    State& s = S();
    int w = wave_id(), l = lane_id();
    unsigned char* mine = s.xbuf[w][l];
    memcpy(mine, &a, 4); memcpy(mine + 4, &b, 4);
    wave_barrier();
    emu_f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; r++) {
        const int row = 4 * (l >> 4) + r;
        float acc = d[r];
        for (int g = 0; g < 4; g++) {
            float av, bv;
            memcpy(&av, s.xbuf[w][g * 16 + row], 4);
            memcpy(&bv, s.xbuf[w][g * 16 + col] + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    wave_barrier();
    return d;
}
// D = A(32x16) * B(16x32) + C.  lane l: A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31],
// C/D[row (r&3)+8*(r>>2)+4*(l>>5)][col l&31]
template <float (*DEC)(unsigned short)> inline emu_f32x16 mfma_32x32x16_b16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c) {
    State& s = S();
    int w = wave_id(), l = lane_id();
    unsigned char* mine = s.xbuf[w][l];
    memcpy(mine, &a, 16); memcpy(mine + 16, &b, 16);
    wave_barrier();
    emu_f32x16 d = c;
    const int col = l & 31;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        for (int g = 0; g < 2; g++) {
            const unsigned short* pa = (const unsigned short*)(s.xbuf[w][g * 32 + row]);
            const unsigned short* pb = (const unsigned short*)(s.xbuf[w][g * 32 + col] + 16);
            for (int j = 0; j < 8; j++) acc = fmaf(DEC(pa[j]), DEC(pb[j]), acc);
        }
        d[r] = acc;
    }
    wave_barrier();
    return d;
}
// IEEE half <-> float through the host compiler's _Float16
inline float f16_to_f32(unsigned short h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
inline unsigned short f32_to_f16(float f) { _Float16 x = (_Float16)f; unsigned short h; memcpy(&h, &x, 2); return h; }
inline emu_f32x4 mfma_16x16x32_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c) { return mfma_16x16x32_b16<bf16_to_f32>(a, b, c); }
inline emu_f32x4 mfma_16x16x32_f16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c) { return mfma_16x16x32_b16<f16_to_f32>(a, b, c); }
inline emu_f32x16 mfma_32x32x16_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c) { return mfma_32x32x16_b16<bf16_to_f32>(a, b, c); }
inline emu_f32x16 mfma_32x32x16_f16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c) { return mfma_32x32x16_b16<f16_to_f32>(a, b, c); }
}  // namespace emu

namespace emu {
typedef short s16x4_t __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16 as measured on gfx950 (tools/probes/tr_probe.hip): in each 16-lane group lane i
// gets element (i&3) of the chunks supplied by lanes (i>>2) + 4j, j = 0..3.
inline s16x4_t ds_read_tr16_b64(const unsigned short* p) {
    State& s = S();
    int w = wave_id(), l = lane_id();
    memcpy(s.xbuf[w][l], &p, sizeof(p));
    wave_barrier();
    s16x4_t r;
    const int grp = l & ~15, i = l & 15;
    for (int j = 0; j < 4; j++) {
        const unsigned short* q;
        memcpy(&q, s.xbuf[w][grp + (i >> 2) + 4 * j], sizeof(q));
        r[j] = (short)q[i & 3];
    }
    wave_barrier();
    return r;
}
}  // namespace emu

namespace emu {
// global_load_lds_dwordx4: lane i's 16 bytes land at (lane 0's LDS pointer) + 16*i
inline void global_load_lds16(const void* g, void* lds) {
    State& s = S();
    int w = wave_id(), l = lane_id();
    memcpy(s.xbuf[w][l], &lds, sizeof(lds));
    wave_barrier();
    int first = 0;
    for (int i = 0; i < 64; i++) if ((w * 64 + i) < s.nthreads && !s.fibers[w * 64 + i].done) { first = i; break; }
    unsigned char* base;
    memcpy(&base, s.xbuf[w][first], sizeof(base));
    memcpy(base + 16 * (l - first), g, 16);
    wave_barrier();
}
}  // namespace emu

#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma_16x16x32_bf16(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16_bf16(a, b, c)

#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define OD_LAUNCH(kern, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })
#define OD_LAUNCH_DYN(kern, grid, block, smem, stream, ...) \
    emu::launch((grid), (block), [&]() { kern(__VA_ARGS__); })
namespace emu { inline unsigned char* dyn_smem() { alignas(16) static unsigned char buf[160 * 1024]; return buf; } }
#define OD_DYN_SMEM(name) unsigned char* name = emu::dyn_smem()
