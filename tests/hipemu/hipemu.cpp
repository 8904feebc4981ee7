// hipemu.cpp — fiber scheduler of the CPU kernel emulator (see hipemu.h). Test infrastructure.
#include "hipemu.h"

#include <sys/mman.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

extern "C" void hipemu_switch(void** save_sp, void* next_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch, .-hipemu_switch
)");

namespace hipemu {

enum { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
static const size_t STACK_BYTES = 256 * 1024;
static const int MAX_THREADS = 1024;

struct Fiber {
    void* sp;
    int state;
};
static Fiber fibers[MAX_THREADS];
static char* stacks = nullptr;
static void* sched_sp;
static int cur = 0;
static int nthreads = 0;
static const std::function<void()>* cur_body = nullptr;
static int live_block = 0, arrived_block = 0;
static int live_wave[MAX_THREADS / 64], arrived_wave[MAX_THREADS / 64];
static char wave_bufs[MAX_THREADS / 64][64 * 64] __attribute__((aligned(64)));
static char* dyn_smem_store = nullptr;
char* dyn_smem = nullptr;

int lane_id() { return cur & 63; }
void* wave_buf() { return wave_bufs[cur >> 6]; }

static void yield_to_sched() { hipemu_switch(&fibers[cur].sp, sched_sp); }

static void release_block_if_complete() {
    if (live_block > 0 && arrived_block == live_block) {
        for (int t = 0; t < nthreads; ++t)
            if (fibers[t].state == WAIT_BLOCK) fibers[t].state = RUN;
        arrived_block = 0;
    }
}
static void release_wave_if_complete(int w) {
    if (live_wave[w] > 0 && arrived_wave[w] == live_wave[w]) {
        int end = (w + 1) * 64 < nthreads ? (w + 1) * 64 : nthreads;
        for (int t = w * 64; t < end; ++t)
            if (fibers[t].state == WAIT_WAVE) fibers[t].state = RUN;
        arrived_wave[w] = 0;
    }
}

void block_barrier() {
    fibers[cur].state = WAIT_BLOCK;
    ++arrived_block;
    yield_to_sched();
}
void wave_barrier() {
    fibers[cur].state = WAIT_WAVE;
    ++arrived_wave[cur >> 6];
    yield_to_sched();
}

static void fiber_entry() {
    (*cur_body)();
    fibers[cur].state = DONE;
    --live_block;
    --live_wave[cur >> 6];
    for (;;) yield_to_sched();
}

static void init_fiber(int t) {
    char* top = stacks + (size_t)(t + 1) * STACK_BYTES;
    uintptr_t T = (uintptr_t)top & ~(uintptr_t)15;
    void** s = (void**)T;
    s[-1] = nullptr;              // fake return address of fiber_entry (never returns)
    s[-2] = (void*)&fiber_entry;  // popped by `ret` in hipemu_switch
    for (int i = 3; i <= 8; ++i) s[-i] = nullptr;  // r15 r14 r13 r12 rbx rbp
    fibers[t].sp = (void*)(s - 8);
    fibers[t].state = RUN;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > MAX_THREADS) {
        fprintf(stderr, "hipemu: bad block size %d\n", nthreads);
        abort();
    }
    if (!stacks) {
        stacks = (char*)mmap(nullptr, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE,
                             MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == (char*)MAP_FAILED) abort();
        dyn_smem_store = (char*)aligned_alloc(64, 160 * 1024);
    }
    if (shmem > 160 * 1024) {
        fprintf(stderr, "hipemu: dynamic LDS %zu > 160 KiB\n", shmem);
        abort();
    }
    dyn_smem = dyn_smem_store;
    blockDim = block;
    gridDim = grid;
    cur_body = &body;
    int nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = dim3(bx, by, bz);
                memset(dyn_smem, 0xCD, shmem);  // poison: LDS is uninitialised on hardware
                for (int t = 0; t < nthreads; ++t) init_fiber(t);
                live_block = nthreads;
                arrived_block = 0;
                for (int w = 0; w < nwaves; ++w) {
                    int end = (w + 1) * 64 < nthreads ? (w + 1) * 64 : nthreads;
                    live_wave[w] = end - w * 64;
                    arrived_wave[w] = 0;
                }
                while (live_block > 0) {
                    bool progressed = false;
                    for (int t = 0; t < nthreads; ++t) {
                        if (fibers[t].state != RUN) continue;
                        cur = t;
                        threadIdx.x = t % block.x;
                        threadIdx.y = (t / block.x) % block.y;
                        threadIdx.z = t / (block.x * block.y);
                        hipemu_switch(&sched_sp, fibers[t].sp);
                        progressed = true;
                        release_wave_if_complete(t >> 6);
                        release_block_if_complete();
                    }
                    if (!progressed) {
                        fprintf(stderr, "hipemu: deadlock (divergent barrier?) block (%u,%u,%u)\n", bx, by, bz);
                        abort();
                    }
                }
            }
    cur_body = nullptr;
}

}  // namespace hipemu
