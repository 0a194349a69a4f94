// hostemu.cpp — fiber scheduler behind hostemu.h.  TEST INFRASTRUCTURE ONLY (see hostemu.h).
#include "hostemu.h"

#include <sys/mman.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void hostemu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hostemu_switch
.type hostemu_switch,@function
hostemu_switch:
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
.size hostemu_switch,.-hostemu_switch
)");

namespace hostemu {

thread_local ThreadCtx g_ctx;

namespace {
constexpr size_t kStack = 64 * 1024;
constexpr int kMaxThreads = 1024;
enum State : int { RUNNABLE = 0, WAIT_WAVE = 1, WAIT_BLOCK = 2, DONE = 3 };

struct Fiber {
    void* sp = nullptr;
    State state = DONE;
    dim3 tid;
    int linear = 0;
};
struct Wave {
    alignas(16) unsigned char slot[64][64];
};
struct Worker {
    char* stacks = nullptr;
    Fiber fibers[kMaxThreads];
    Wave waves[kMaxThreads / 64];
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    int nthreads = 0;
};
thread_local Worker* tl_worker = nullptr;

Worker* worker() {
    if (!tl_worker) {
        tl_worker = new Worker();
        void* m = mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { std::perror("hostemu mmap"); std::abort(); }
        tl_worker->stacks = static_cast<char*>(m);
    }
    return tl_worker;
}

void yield_to_scheduler() {
    Worker* w = tl_worker;
    Fiber* f = w->cur;
    hostemu_switch(&f->sp, w->sched_sp);
}

void fiber_entry() {
    Worker* w = tl_worker;
    (*w->body)();
    w->cur->state = DONE;
    yield_to_scheduler();
    std::abort();  // a finished fiber is never resumed
}

void resume(Worker* w, Fiber* f) {
    w->cur = f;
    g_ctx.tid = f->tid;
    hostemu_switch(&w->sched_sp, f->sp);
}

void run_block(Worker* w, dim3 block) {
    const int n = int(block.x * block.y * block.z);
    if (n > kMaxThreads) { std::fprintf(stderr, "hostemu: block too large\n"); std::abort(); }
    w->nthreads = n;
    for (int t = 0; t < n; ++t) {
        Fiber& f = w->fibers[t];
        f.linear = t;
        f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        f.state = RUNNABLE;
        char* top = w->stacks + size_t(t + 1) * kStack;           // 16-byte aligned (kStack multiple of 16)
        void** sp = reinterpret_cast<void**>(top - 16);
        sp[0] = reinterpret_cast<void*>(&fiber_entry);             // `ret` target; rsp == 8 (mod 16) on entry
        sp -= 6;                                                   // rbp rbx r12 r13 r14 r15
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        f.sp = sp;
    }
    const int nwaves = (n + 63) / 64;
    int done = 0;
    while (done < n) {
        bool progress = false;
        for (int wv = 0; wv < nwaves; ++wv) {
            const int lo = wv * 64, hi = (lo + 64 < n) ? lo + 64 : n;
            for (;;) {
                bool ran = false;
                for (int t = lo; t < hi; ++t) {
                    if (w->fibers[t].state == RUNNABLE) {
                        resume(w, &w->fibers[t]);
                        ran = true;
                        if (w->fibers[t].state == DONE) ++done;
                    }
                }
                progress |= ran;
                int waiting = 0, alive = 0;
                for (int t = lo; t < hi; ++t) {
                    if (w->fibers[t].state != DONE) ++alive;
                    if (w->fibers[t].state == WAIT_WAVE) ++waiting;
                }
                if (waiting > 0 && waiting == alive) {
                    for (int t = lo; t < hi; ++t)
                        if (w->fibers[t].state == WAIT_WAVE) w->fibers[t].state = RUNNABLE;
                    progress = true;
                    continue;
                }
                if (waiting > 0) {
                    std::fprintf(stderr, "hostemu: divergent wave collective (wave %d: %d of %d lanes waiting)\n", wv, waiting, alive);
                    std::abort();
                }
                break;
            }
        }
        int at_barrier = 0;
        for (int t = 0; t < n; ++t)
            if (w->fibers[t].state == WAIT_BLOCK) ++at_barrier;
        if (at_barrier > 0) {
            // exited threads do not take part in the barrier (as waves that have ended do not on the hardware)
            if (at_barrier != n - done) { std::fprintf(stderr, "hostemu: inconsistent __syncthreads\n"); std::abort(); }
            for (int t = 0; t < n; ++t)
                if (w->fibers[t].state == WAIT_BLOCK) w->fibers[t].state = RUNNABLE;
            progress = true;
        }
        if (!progress && done < n) { std::fprintf(stderr, "hostemu: deadlock\n"); std::abort(); }
    }
}
}  // namespace

void block_sync() {
    tl_worker->cur->state = WAIT_BLOCK;
    yield_to_scheduler();
}
int lane_id() { return tl_worker->cur->linear & 63; }
void wave_deposit(const void* src, int bytes) {
    Worker* w = tl_worker;
    Fiber* f = w->cur;
    std::memcpy(w->waves[f->linear >> 6].slot[f->linear & 63], src, size_t(bytes));
    f->state = WAIT_WAVE;
    yield_to_scheduler();
}
const void* wave_slot(int lane) {
    Worker* w = tl_worker;
    return w->waves[w->cur->linear >> 6].slot[lane & 63];
}
void wave_release() {
    // second rendezvous: nobody may overwrite its slot before every lane has read what it needs
    tl_worker->cur->state = WAIT_WAVE;
    yield_to_scheduler();
}

namespace {
// persistent pool: fibers' stacks and the kernels' `static thread_local` LDS stay warm across launches
struct Pool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    const std::function<void()>* body = nullptr;
    dim3 grid, block;
    long total = 0;
    std::atomic<long> next{0};
    unsigned long generation = 0;
    int active = 0;
    int nthreads = 0;

    void worker_main() {
        unsigned long seen = 0;
        Worker* w = worker();
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return generation != seen; });
                seen = generation;
            }
            w->body = body;
            g_ctx.bdim = block;
            g_ctx.gdim = grid;
            for (;;) {
                long b = next.fetch_add(1);
                if (b >= total) break;
                g_ctx.bid = dim3(unsigned(b % grid.x), unsigned((b / grid.x) % grid.y), unsigned(b / (long(grid.x) * grid.y)));
                run_block(w, block);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--active == 0) cv_done.notify_all();
            }
        }
    }
    void ensure_started() {
        if (nthreads) return;
        unsigned hw = std::thread::hardware_concurrency();
        nthreads = int(hw ? hw : 4);
        if (const char* e = std::getenv("HOSTEMU_THREADS")) nthreads = std::atoi(e) > 0 ? std::atoi(e) : nthreads;
        for (int i = 0; i < nthreads; ++i) {
            threads.emplace_back([this] { worker_main(); });
            threads.back().detach();
        }
    }
    void run(dim3 g, dim3 b, const std::function<void()>& fn) {
        ensure_started();
        std::unique_lock<std::mutex> lk(mu);
        body = &fn;
        grid = g;
        block = b;
        total = long(g.x) * g.y * g.z;
        next.store(0);
        active = nthreads;
        ++generation;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
    }
};
Pool& pool() { static Pool* p = new Pool(); return *p; }
std::mutex g_launch_mu;
}  // namespace

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    if (long(grid.x) * grid.y * grid.z == 0) return;
    std::lock_guard<std::mutex> lk(g_launch_mu);   // one kernel at a time (a stream, in effect)
    pool().run(grid, block, body);
}

}  // namespace hostemu
