// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Fiber scheduler of the host emulation
// (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>

thread_local uint3_ threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

struct hostsim_event { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hostsim_event; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

namespace hostsim {

namespace {
constexpr size_t STACK = 256 * 1024;

// Switch between fibers of one OS thread: callee-saved integer registers and the stack pointer only (x86-64 System V; the
// kernels change neither the x87 control word nor MXCSR).  ucontext's swapcontext makes a signal-mask system call per
// switch, which was most of the emulation's run time: a lane shift is two rendezvous of 64 fibers.
#if !defined(__x86_64__)
#error "the host emulation's fiber switch is written for x86-64"
#endif
struct Ctx { void* sp = nullptr; };
extern "C" void hostsim_swap(Ctx* from, Ctx* to);
asm(R"(
    .text
    .globl hostsim_swap
    .type hostsim_swap,@function
hostsim_swap:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hostsim_swap,.-hostsim_swap
)");

struct Fiber {
    Ctx ctx;
    bool done = false;
    uint3_ tid;
};
struct BlockRun {
    std::vector<Fiber> fibers;
    Ctx sched;
    int current = -1;
    const std::function<void()>* body = nullptr;
    std::vector<double> xchg;
    // true barriers (threads of a block / lanes of a 64-wide wave may take different paths between two rendezvous,
    // e.g. wave-specialised kernels): arrivals are counted against the threads that have not returned yet
    int alive = 0;
    long bar_gen = 0;
    int bar_arrived = 0;
    std::vector<int> wave_alive, wave_arrived;
    std::vector<long> wave_gen;
};
thread_local BlockRun* g_run = nullptr;

// the stacks of a worker thread live as long as the thread: a launch reuses them
struct StackPool {
    std::vector<char*> s;
    ~StackPool() { for (char* p : s) std::free(p); }
    char* get(size_t i)
    {
        while (s.size() <= i) s.push_back((char*)std::malloc(STACK));
        return s[i];
    }
};
thread_local StackPool g_stacks;

void fiber_entry()
{
    BlockRun* r = g_run;
    (*r->body)();
    r->fibers[r->current].done = true;
    hostsim_swap(&r->fibers[r->current].ctx, &r->sched);
    __builtin_trap();   // a finished fiber is never resumed
}

void fiber_prepare(Fiber& f, char* stack)
{
    // as hostsim_swap leaves it: six saved registers, then the address it returns to; the entry sees the alignment of a call
    uintptr_t top = ((uintptr_t)stack + STACK) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;                 // the entry's (unused) return address
    *--sp = (void*)&fiber_entry;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.ctx.sp = sp;
}

void yield_to_sched()
{
    BlockRun* r = g_run;
    hostsim_swap(&r->fibers[r->current].ctx, &r->sched);
    threadIdx = r->fibers[r->current].tid;   // restored by the scheduler as well
}
}  // namespace

void syncthreads()
{
    BlockRun* r = g_run;
    const long gen = r->bar_gen;
    r->bar_arrived++;
    while (r->bar_gen == gen) {
        if (r->bar_arrived >= r->alive) { r->bar_arrived = 0; r->bar_gen++; break; }
        yield_to_sched();
    }
}

namespace {
void wave_barrier(BlockRun* r, int w)
{
    const long gen = r->wave_gen[w];
    r->wave_arrived[w]++;
    while (r->wave_gen[w] == gen) {
        if (r->wave_arrived[w] >= r->wave_alive[w]) { r->wave_arrived[w] = 0; r->wave_gen[w]++; break; }
        yield_to_sched();
    }
}
}  // namespace

// exchange between the lanes of ONE 64-wide wave (consecutive linear thread ids); src is a linear thread id of the block
double shfl_exchange(double v, int src)
{
    BlockRun* r = g_run;
    const int me = r->current, w = me / 64;
    r->xchg[me] = v;
    wave_barrier(r, w);          // every live lane of the wave has published
    const double out = r->xchg[src];
    wave_barrier(r, w);          // every live lane has read before the next publish
    return out;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
    const int nthreads = (int)(block.x * block.y * block.z);
    const long nblocks = (long)grid.x * grid.y * grid.z;
#pragma omp parallel
    {
        BlockRun run;
        run.fibers.resize(nthreads);
        run.xchg.resize(nthreads);
        run.body = &body;
        g_run = &run;
#pragma omp for schedule(dynamic)
        for (long bi = 0; bi < nblocks; ++bi) {
            blockDim = block;
            gridDim = grid;
            blockIdx.x = (unsigned)(bi % grid.x);
            blockIdx.y = (unsigned)((bi / grid.x) % grid.y);
            blockIdx.z = (unsigned)(bi / ((long)grid.x * grid.y));
            for (int t = 0; t < nthreads; ++t) {
                Fiber& f = run.fibers[t];
                f.done = false;
                f.tid.x = t % block.x;
                f.tid.y = (t / block.x) % block.y;
                f.tid.z = t / (block.x * block.y);
                fiber_prepare(f, g_stacks.get((size_t)t));
            }
            const int nwaves = (nthreads + 63) / 64;
            run.alive = nthreads;
            run.bar_arrived = 0;
            run.wave_alive.assign(nwaves, 0);
            run.wave_arrived.assign(nwaves, 0);
            run.wave_gen.assign(nwaves, 0);
            for (int t = 0; t < nthreads; ++t) run.wave_alive[t / 64]++;
            while (run.alive > 0) {
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = run.fibers[t];
                    if (f.done) continue;
                    run.current = t;
                    threadIdx = f.tid;
                    hostsim_swap(&run.sched, &f.ctx);
                    if (f.done) { run.alive--; run.wave_alive[t / 64]--; }
                }
            }
        }
        g_run = nullptr;
    }
}

}  // namespace hostsim
