// TEST INFRASTRUCTURE ONLY — fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
// One ucontext fiber per GPU thread of the workgroup being emulated; workgroups run one
// after another.  Rendezvous points: __syncthreads (whole workgroup) and wave collectives
// (the 64 lanes of a wavefront: shuffles, ballots, MFMA operand exchange).
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <vector>

emu_uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
constexpr size_t kStack = 256 * 1024;
enum Wait { RUN = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = RUN;
    unsigned gen = 0;  // generation being waited for
    emu_uint3 tid;
    int flat = 0;
};
struct Wave {
    int live = 0;
    int arrived = 0;
    unsigned gen = 0;
    unsigned char slot[2][64][64];
};

ucontext_t g_main;
std::vector<Fiber> g_fibers;
std::vector<Wave> g_waves;
int g_cur = -1;
int g_block_live = 0, g_block_arrived = 0;
unsigned g_block_gen = 0;
void (*g_thunk)(void*) = nullptr;
void* g_ctx = nullptr;

void fiber_entry() {
    g_thunk(g_ctx);
    Fiber& f = g_fibers[g_cur];
    f.state = DONE;
    g_block_live--;
    g_waves[f.flat / 64].live--;
    // a thread that exits releases rendezvous that were only waiting for it
    if (g_block_live > 0 && g_block_arrived == g_block_live) { g_block_arrived = 0; g_block_gen++; }
    Wave& w = g_waves[f.flat / 64];
    if (w.live > 0 && w.arrived == w.live) { w.arrived = 0; w.gen++; }
    swapcontext(&f.ctx, &g_main);
}

void yield_to_main() {
    Fiber& f = g_fibers[g_cur];
    swapcontext(&f.ctx, &g_main);
}
}  // namespace

int emu_lane_id() { return g_fibers[g_cur].flat & 63; }

void emu_sync_block() {
    Fiber& f = g_fibers[g_cur];
    unsigned my = g_block_gen;
    g_block_arrived++;
    if (g_block_arrived == g_block_live) {
        g_block_arrived = 0;
        g_block_gen++;
        return;
    }
    f.state = WAIT_BLOCK;
    f.gen = my;
    yield_to_main();
}

void emu_wave_exchange(const void* mine, unsigned bytes, void* all64) {
    Fiber& f = g_fibers[g_cur];
    Wave& w = g_waves[f.flat / 64];
    unsigned my = w.gen;
    int lane = f.flat & 63;
    if (bytes > 64) { fprintf(stderr, "emu: exchange too wide\n"); abort(); }
    memcpy(w.slot[my & 1][lane], mine, bytes);
    w.arrived++;
    if (w.arrived == w.live) {
        w.arrived = 0;
        w.gen++;
    } else {
        f.state = WAIT_WAVE;
        f.gen = my;
        yield_to_main();
    }
    unsigned char* out = (unsigned char*)all64;
    for (int i = 0; i < 64; ++i) memcpy(out + (size_t)i * bytes, w.slot[my & 1][i], bytes);
}

void emu_run_grid(dim3 grid, dim3 block, void (*thunk)(void*), void* ctx) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "emu: bad block size %d\n", nthreads); abort(); }
    gridDim = grid;
    blockDim = block;
    g_thunk = thunk;
    g_ctx = ctx;
    if ((int)g_fibers.size() < nthreads) {
        size_t old = g_fibers.size();
        g_fibers.resize(nthreads);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (char*)malloc(kStack);
    }
    const int nwaves = (nthreads + 63) / 64;
    g_waves.resize(nwaves);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx = {bx, by, bz};
                g_block_live = nthreads;
                g_block_arrived = 0;
                g_block_gen = 0;
                for (int w = 0; w < nwaves; ++w) {
                    g_waves[w].live = std::min(64, nthreads - 64 * w);
                    g_waves[w].arrived = 0;
                    g_waves[w].gen = 0;
                }
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    f.state = RUN;
                    f.flat = t;
                    f.tid.x = t % block.x;
                    f.tid.y = (t / block.x) % block.y;
                    f.tid.z = t / (block.x * block.y);
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = &g_main;
                    makecontext(&f.ctx, fiber_entry, 0);
                }
                int done = 0;
                while (done < nthreads) {
                    bool progressed = false;
                    done = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = g_fibers[t];
                        if (f.state == DONE) { done++; continue; }
                        if (f.state == WAIT_BLOCK && g_block_gen == f.gen) continue;
                        if (f.state == WAIT_WAVE && g_waves[t / 64].gen == f.gen) continue;
                        f.state = RUN;
                        g_cur = t;
                        threadIdx = f.tid;
                        swapcontext(&g_main, &f.ctx);
                        progressed = true;
                        if (f.state == DONE) done++;
                    }
                    if (!progressed && done < nthreads) {
                        fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier/collective\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g_cur = -1;
}
