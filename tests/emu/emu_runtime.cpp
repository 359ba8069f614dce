// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
//
// One workgroup at a time; its threads are cooperative fibers on the calling OS thread, resumed
// round-robin.  __syncthreads and the wave-collective builtins are rendezvous points.  A round in
// which no fiber makes progress is reported as a deadlock (e.g. a barrier inside divergent code).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch, .-emu_switch
)");

namespace emu {

Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
alignas(64) static unsigned char g_smem_buf[160 * 1024];
unsigned char* g_dyn_smem = g_smem_buf;

namespace {
constexpr size_t kStack = 96 * 1024;
struct Fiber { void* sp = nullptr; bool done = false; unsigned char* stack = nullptr; };
std::vector<Fiber> g_fibers;
void* g_sched_sp = nullptr;
int g_cur = -1, g_nt = 0, g_alive = 0;
const std::function<void()>* g_body = nullptr;
bool g_progress = false;
const char* g_error = nullptr;

// block barrier
int g_bar_count = 0; unsigned g_bar_gen = 0;
// wave rendezvous
struct WaveState { int count = 0; unsigned gen = 0; float a[2][64], b[2][64]; const void* gp[2][64]; void* lp[2][64]; float fa[2][64][8], fb[2][64][8]; int alive = 0; };
std::vector<WaveState> g_waves;

void yield() { emu_switch(&g_fibers[g_cur].sp, g_sched_sp); }

void fiber_entry() {
    (*g_body)();
    Fiber& f = g_fibers[g_cur];
    f.done = true;
    g_progress = true;
    --g_alive;
    --g_waves[g_cur / 64].alive;
    // a thread that exits releases barriers the remaining threads are waiting on
    if (g_alive > 0 && g_bar_count == g_alive) { g_bar_count = 0; ++g_bar_gen; }
    WaveState& w = g_waves[g_cur / 64];
    if (w.alive > 0 && w.count == w.alive) { w.count = 0; ++w.gen; }
    emu_switch(&f.sp, g_sched_sp);
    std::abort();  // never resumed
}

void wave_rendezvous(WaveState& w) {
    const unsigned gen = w.gen;
    g_progress = true;
    if (++w.count == w.alive) { w.count = 0; ++w.gen; return; }
    while (w.gen == gen) yield();
}
}  // namespace

void syncthreads() {
    const unsigned gen = g_bar_gen;
    g_progress = true;
    if (++g_bar_count == g_alive) { g_bar_count = 0; ++g_bar_gen; return; }
    while (g_bar_gen == gen) yield();
}

void wave_sync() { wave_rendezvous(g_waves[g_cur / 64]); }

float shfl_xor(float v, int mask) {
    WaveState& w = g_waves[g_cur / 64];
    const int lane = g_cur & 63, par = w.gen & 1;
    w.a[par][lane] = v;
    wave_rendezvous(w);
    return w.a[par][(lane ^ mask) & 63];
}

f32x16_t mfma_32x32x2(float a, float b, f32x16_t c) {
    WaveState& w = g_waves[g_cur / 64];
    const int lane = g_cur & 63, par = w.gen & 1;
    w.a[par][lane] = a;
    w.b[par][lane] = b;
    wave_rendezvous(w);
    // A[i][k] is held by lane i+32k, B[k][j] by lane j+32k; D: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    const int j = lane & 31, hi = lane >> 5;
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float t = d[r];
        t = fmaf(w.a[par][i], w.b[par][j], t);
        t = fmaf(w.a[par][i + 32], w.b[par][j + 32], t);
        d[r] = t;
    }
    return d;
}

// v_mfma_f32_32x32x16_{bf16,f16}: lane l holds A[i = l&31][k = 8*(l>>5) + 0..7] and B[k = 8*(l>>5) + 0..7][j = l&31]; products are exact in
// fp32 (8 x 8 / 11 x 11 significand bits), accumulated in k order (the hardware's internal order may differ: same error class).  Every lane
// widens its own fragments to fp32 BEFORE the rendezvous, so the 256 fused multiply-adds per lane run on plain floats.
static f32x16_t mfma_16bit(const float (&fa)[8], const float (&fb)[8], f32x16_t c) {
    WaveState& w = g_waves[g_cur / 64];
    const int lane = g_cur & 63, par = w.gen & 1;
    memcpy(w.fa[par][lane], fa, sizeof fa);
    memcpy(w.fb[par][lane], fb, sizeof fb);
    wave_rendezvous(w);
    const int j = lane & 31, hi = lane >> 5;
    const float* b0 = w.fb[par][j];
    const float* b1 = w.fb[par][j + 32];
    f32x16_t d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float* a0 = w.fa[par][i];
        const float* a1 = w.fa[par][i + 32];
        float t = d[r];
        for (int k = 0; k < 8; ++k) t = fmaf(a0[k], b0[k], t);
        for (int k = 0; k < 8; ++k) t = fmaf(a1[k], b1[k], t);
        d[r] = t;
    }
    return d;
}

f32x16_t mfma_bf16_32x32x16(const void* a16, const void* b16, f32x16_t c) {
    unsigned short ha[8], hb[8];
    memcpy(ha, a16, 16); memcpy(hb, b16, 16);
    float fa[8], fb[8];
    for (int k = 0; k < 8; ++k) {
        unsigned ua = (unsigned)ha[k] << 16, ub = (unsigned)hb[k] << 16;
        memcpy(&fa[k], &ua, 4); memcpy(&fb[k], &ub, 4);
    }
    return mfma_16bit(fa, fb, c);
}

f32x16_t mfma_f16_32x32x16(const void* a16, const void* b16, f32x16_t c) {
    _Float16 ha[8], hb[8];
    memcpy(ha, a16, 16); memcpy(hb, b16, 16);
    float fa[8], fb[8];
    for (int k = 0; k < 8; ++k) { fa[k] = (float)ha[k]; fb[k] = (float)hb[k]; }
    return mfma_16bit(fa, fb, c);
}

void buf_dma16(const unsigned char* base, unsigned bytes, unsigned voff, unsigned soff, unsigned char* lds) {
    // raw-buffer LDS-DMA: range check on voffset against num_records - soffset (gfx9 raw-buffer rule);
    // destination = lane 0's LDS address + 16*lane (M0 semantics)
    WaveState& w = g_waves[g_cur / 64];
    const int lane = g_cur & 63, par = w.gen & 1;
    w.lp[par][lane] = lds;
    if (w.alive != 64) throw std::runtime_error("emu: buffer LDS-DMA needs a full, converged wave");
    wave_rendezvous(w);
    unsigned char* dst = static_cast<unsigned char*>(w.lp[par][0]) + (size_t)lane * 16;
    const bool oob = soff > bytes || (unsigned long long)voff + 16 > (unsigned long long)(bytes - soff);
    if (oob) memset(dst, 0, 16); else memcpy(dst, base + (size_t)voff + soff, 16);
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || nt > 1024) throw std::runtime_error("emu: bad block size");
    if (block.y != 1 || block.z != 1) throw std::runtime_error("emu: only 1-D blocks are emulated");
    if (shmem > sizeof g_smem_buf) throw std::runtime_error("emu: dynamic LDS request exceeds 160 KiB");
    if ((int)g_fibers.size() < nt) {
        const size_t old = g_fibers.size();
        g_fibers.resize(nt);
        for (size_t i = old; i < g_fibers.size(); ++i) g_fibers[i].stack = (unsigned char*)aligned_alloc(64, kStack);
    }
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    g_body = &body;
    g_nt = nt;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = {bx, by, bz};
                g_alive = nt;
                g_bar_count = 0;
                g_waves.assign((nt + 63) / 64, WaveState());
                for (int t = 0; t < nt; ++t) {
                    Fiber& f = g_fibers[t];
                    f.done = false;
                    g_waves[t / 64].alive++;
                    void** top = reinterpret_cast<void**>(f.stack + kStack);   // 16-byte aligned
                    top[-1] = nullptr;                                  // fake return address slot
                    top[-2] = reinterpret_cast<void*>(&fiber_entry);    // ret target
                    for (int k = 3; k <= 8; ++k) top[-k] = nullptr;     // rbp rbx r12..r15
                    f.sp = &top[-8];
                }
                while (g_alive > 0) {
                    g_progress = false;
                    for (int t = 0; t < nt; ++t) {
                        if (g_fibers[t].done) continue;
                        g_cur = t;
                        g_threadIdx = {(unsigned)t, 0, 0};
                        emu_switch(&g_sched_sp, g_fibers[t].sp);
                    }
                    if (!g_progress && g_alive > 0) throw std::runtime_error("emu: deadlock (divergent barrier or wave collective)");
                }
            }
    g_body = nullptr;
}

}  // namespace emu
