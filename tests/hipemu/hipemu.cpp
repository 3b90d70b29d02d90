// hipemu.cpp -- fiber scheduler of the developer-only HIP kernel-logic simulator (see hipemu.h).
#include "hipemu.h"
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <atomic>

namespace hipemu {

Block*& tls_block() {
    static thread_local Block* b = nullptr;
    return b;
}

struct Worker {
    Block blk;
    std::vector<Fiber> fibers;
    unsigned char* stacks = nullptr;
    unsigned char* smem = nullptr;
    size_t smem_cap = 0;
    Worker() {
        fibers.resize(kMaxThreads);
        stacks = (unsigned char*)mmap(nullptr, kStackBytes * kMaxThreads, PROT_READ | PROT_WRITE,
                                      MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == (unsigned char*)MAP_FAILED) { perror("hipemu mmap"); abort(); }
        for (int i = 0; i < kMaxThreads; ++i) fibers[i].stack = stacks + (size_t)i * kStackBytes;
    }
    ~Worker() { munmap(stacks, kStackBytes * kMaxThreads); free(smem); }
};

static void fiber_entry() {
    Block* b = tls_block();
    Fiber* f = &b->fibers[b->cur];
    (*b->body)();
    f->done = true;
    swapcontext(&f->ctx, &b->sched);
}

static void run_block(Worker& w, dim3 bid, dim3 grid, dim3 bdim, size_t smem_bytes, const std::function<void()>& body) {
    Block& b = w.blk;
    int n = (int)(bdim.x * bdim.y * bdim.z);
    if (n > kMaxThreads) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
    if (smem_bytes + 16 > w.smem_cap) {
        free(w.smem);
        w.smem_cap = smem_bytes + 4096;
        w.smem = (unsigned char*)aligned_alloc(256, (w.smem_cap + 255) / 256 * 256);
    }
    // poison LDS so that reads of never-written LDS show up as NaN / garbage, as on hardware
    memset(w.smem, 0xFF, smem_bytes + 16);
    b.bid = bid; b.gdim = grid; b.bdim = bdim; b.smem = w.smem; b.nthreads = n; b.fibers = w.fibers.data();
    b.bar_count = 0; b.bar_gen = 0; b.body = &body;
    int nw = (n + kWave - 1) / kWave;
    for (int i = 0; i < nw; ++i) {
        b.waves[i].count = 0; b.waves[i].gen = 0;
        b.waves[i].nlanes = (i == nw - 1) ? n - i * kWave : kWave;
    }
    tls_block() = &b;
    for (int i = 0; i < n; ++i) {
        Fiber& f = b.fibers[i];
        f.lin = i; f.lane = i % kWave; f.wave = i / kWave;
        f.tid = dim3(i % bdim.x, (i / bdim.x) % bdim.y, i / (bdim.x * bdim.y));
        f.done = false; f.wait_gen = nullptr; f.wave_ops = 0;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int remaining = n;
    while (remaining > 0) {
        bool progressed = false;
        for (int i = 0; i < n; ++i) {
            Fiber& f = b.fibers[i];
            if (f.done) continue;
            if (f.wait_gen && *f.wait_gen == f.wait_val) continue;
            b.cur = i;
            swapcontext(&b.sched, &f.ctx);
            progressed = true;
            if (f.done) --remaining;
        }
        if (!progressed && remaining > 0) {
            fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): %d threads blocked at a barrier / wave collective "
                            "that the rest of the workgroup/wave never reaches\n", bid.x, bid.y, bid.z, remaining);
            abort();
        }
    }
    tls_block() = nullptr;
}

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    size_t total = (size_t)grid.x * grid.y * grid.z;
    if (total == 0) return;
    int nthreads = 8;
    if (const char* e = getenv("HIPEMU_THREADS")) nthreads = atoi(e);
    if ((size_t)nthreads > total) nthreads = (int)total;
    if (nthreads < 1) nthreads = 1;
    std::atomic<size_t> next(0);
    static std::vector<Worker*> pool;            // launches are serialised by the caller (one stream)
    while ((int)pool.size() < nthreads) pool.push_back(new Worker());
    auto work = [&](int t) {
        Worker* w = pool[t];
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= total) break;
            dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((size_t)grid.x * grid.y)));
            run_block(*w, bid, grid, block, smem_bytes, body);
        }
    };
    if (nthreads == 1) { work(0); return; }
    std::vector<std::thread> ts;
    for (int t = 0; t < nthreads; ++t) ts.emplace_back(work, t);
    for (auto& t : ts) t.join();
}

}  // namespace hipemu
