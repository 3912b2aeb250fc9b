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
    // true barriers (threads of a block / lanes of a 64-wide wave may take different paths between two rendezvous,
    // e.g. wave-specialised kernels): arrivals are counted against the threads that have not returned yet
    int alive = 0;
    long bar_gen = 0;
    int bar_arrived = 0;
    std::vector<int> wave_alive, wave_arrived;
    std::vector<long> wave_gen;
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
                    swapcontext(&run.sched, &f.ctx);
                    if (f.done) { run.alive--; run.wave_alive[t / 64]--; }
                }
            }
        }
        for (auto& f : run.fibers) std::free(f.stack);
        g_run = nullptr;
    }
}

}  // namespace hostsim
