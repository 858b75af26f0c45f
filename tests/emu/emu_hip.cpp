// TEST INFRASTRUCTURE -- see emu_hip.h.  Workgroups run one after another; the lanes of a
// workgroup are ucontext fibers scheduled round-robin, yielding at __syncthreads().
#include "emu_hip.h"

#include <stdio.h>
#include <time.h>
#include <ucontext.h>

#include <mutex>
#include <vector>

namespace emu {

ThreadCtx* cur = nullptr;
Dim3 g_block_dim = {1, 1, 1}, g_grid_dim = {1, 1, 1};
char* g_dyn_smem = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    ThreadCtx tc;
    bool done;
};
std::vector<Fiber> fibers;
std::vector<char> stacks;
ucontext_t sched_ctx;
const std::function<void()>* g_body = nullptr;
Fiber* g_running = nullptr;

void trampoline() {
    (*g_body)();
    g_running->done = true;
    swapcontext(&g_running->ctx, &sched_ctx);
}
}  // namespace

static void yield() {
    Fiber* f = g_running;
    swapcontext(&f->ctx, &sched_ctx);
    cur = &f->tc;
}

// counting barrier over the fibers of the workgroup that are still alive (a lane that has returned no longer takes part,
// as on the hardware)
static unsigned g_alive = 0, g_arrived = 0, g_gen = 0;
static void barrier_release_if_complete() {
    if (g_arrived > 0 && g_arrived >= g_alive) {
        g_arrived = 0;
        g_gen++;
    }
}
void syncthreads() {
    const unsigned gen = g_gen;
    g_arrived++;
    barrier_release_if_complete();
    while (g_gen == gen) yield();
}

// per-wave rendezvous with a one-word payload per lane
namespace {
struct WaveState {
    uint32_t in[64], out[64];
    unsigned arrived, gen;
};
std::vector<WaveState> waves;
}  // namespace
const uint32_t* wave_exchange(uint32_t v) {
    const unsigned tid = g_running->tc.tid.x, w = tid / 64, lane = tid % 64;
    WaveState& ws = waves[w];
    const unsigned lanes = g_block_dim.x - 64 * w < 64 ? g_block_dim.x - 64 * w : 64;
    const unsigned gen = ws.gen;
    ws.in[lane] = v;
    if (++ws.arrived == lanes) {
        for (unsigned i = 0; i < 64; i++) ws.out[i] = i < lanes ? ws.in[i] : 0;
        ws.arrived = 0;
        ws.gen++;
    } else {
        while (ws.gen == gen) yield();
    }
    return ws.out;
}

static int g_count_acc = 0;
int syncthreads_count(int pred) {
    g_count_acc += pred;
    syncthreads();
    const int r = g_count_acc;
    syncthreads();
    g_count_acc = 0;  // every fiber resets after the second barrier; all have read r by then
    syncthreads();
    return r;
}

// one kernel at a time: the fiber scheduler and the `__shared__` statics are process-wide, and the multi-device entry
// points of the library launch from one host thread per (emulated) device
static std::mutex g_launch_mu;
void launch(unsigned grid, unsigned block, size_t shmem, const std::function<void()>& body) {
    if (grid == 0 || block == 0) return;
    std::lock_guard<std::mutex> lk(g_launch_mu);
    g_block_dim = {block, 1, 1};
    g_grid_dim = {grid, 1, 1};
    std::vector<char> smem(shmem + 64);
    g_dyn_smem = smem.data() + ((64 - ((uintptr_t)smem.data() & 63)) & 63);
    if (fibers.size() < block) fibers.resize(block);
    if (waves.size() < (block + 63) / 64) waves.resize((block + 63) / 64);
    if (stacks.size() < (size_t)block * kStack) stacks.resize((size_t)block * kStack);
    g_body = &body;
    for (unsigned b = 0; b < grid; b++) {
        for (unsigned t = 0; t < block; t++) {
            Fiber& f = fibers[t];
            f.done = false;
            f.tc.tid = {t, 0, 0};
            f.tc.bid = {b, 0, 0};
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = stacks.data() + (size_t)t * kStack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
        unsigned remaining = block;
        g_alive = block;
        g_arrived = 0;
        for (auto& ws : waves) ws.arrived = 0;
        while (remaining) {
            for (unsigned t = 0; t < block; t++) {
                Fiber& f = fibers[t];
                if (f.done) continue;
                g_running = &f;
                cur = &f.tc;
                swapcontext(&sched_ctx, &f.ctx);
                if (f.done) {
                    remaining--;
                    g_alive--;
                    barrier_release_if_complete();   // the lanes still at a barrier no longer wait for this one
                }
            }
        }
    }
    cur = nullptr;
    g_dyn_smem = nullptr;
}
}  // namespace emu

struct emuEvent {
    double t;
};
static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// ZK_EMU_DEVICES=k makes the emulator report k "devices" (all of them this host's memory): the multi-device host logic
// of the library (per-device contexts, window fan-out, partial-sum addition) is exercised by the CPU test tier
hipError_t hipGetDeviceCount(int* n) {
    const char* e = getenv("ZK_EMU_DEVICES");
    const int v = e ? atoi(e) : 1;
    *n = v >= 1 && v <= 16 ? v : 1;
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "cpu-emulator (tests only)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
    *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) {
    *st = (hipStream_t)1;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "emu"; }
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new emuEvent{0};
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = now_ms();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = (float)(b->t - a->t);
    return hipSuccess;
}
