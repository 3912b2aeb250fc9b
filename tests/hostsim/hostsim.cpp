// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Fiber scheduler of the host emulation
// (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

#include <chrono>

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
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    uint3_ tid;
};
struct BlockRun {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    int current = -1;
    const std::function<void()>* body = nullptr;
    std::vector<double> xchg;
};
thread_local BlockRun* g_run = nullptr;

void fiber_entry()
{
    BlockRun* r = g_run;
    (*r->body)();
    r->fibers[r->current].done = true;
    swapcontext(&r->fibers[r->current].ctx, &r->sched);
}

void yield_to_sched()
{
    BlockRun* r = g_run;
    swapcontext(&r->fibers[r->current].ctx, &r->sched);
    threadIdx = r->fibers[r->current].tid;   // restored by the scheduler as well
}
}  // namespace

void syncthreads() { yield_to_sched(); }

double shfl_exchange(double v, int src)
{
    BlockRun* r = g_run;
    const int me = r->current;
    r->xchg[me] = v;
    yield_to_sched();            // everybody has published
    const double out = r->xchg[src];
    yield_to_sched();            // everybody has read before the next publish
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
        for (auto& f : run.fibers) f.stack = (char*)std::malloc(STACK);
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
                getcontext(&f.ctx);
                f.ctx.uc_stack.ss_sp = f.stack;
                f.ctx.uc_stack.ss_size = STACK;
                f.ctx.uc_link = nullptr;
                makecontext(&f.ctx, fiber_entry, 0);
            }
            int alive = nthreads;
            while (alive > 0) {
                alive = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = run.fibers[t];
                    if (f.done) continue;
                    run.current = t;
                    threadIdx = f.tid;
                    swapcontext(&run.sched, &f.ctx);
                    if (!f.done) ++alive;
                }
            }
        }
        for (auto& f : run.fibers) std::free(f.stack);
        g_run = nullptr;
    }
}

}  // namespace hostsim
