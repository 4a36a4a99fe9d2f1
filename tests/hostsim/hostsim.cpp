// Fiber scheduler behind tests/hostsim/hip/hip_runtime.h (test infrastructure only).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

namespace hostsim {

constexpr size_t kStack = 256 * 1024;

struct Fiber {
  ucontext_t uc;
  Ctx ctx;
  char* stack = nullptr;
  bool done = false;
  int wave = 0, lane = 0;
};

struct Wave {
  int live = 0;
  int arrived = 0;
  unsigned gen = 0;
  uint32_t slot[64];
  float fa[64], fb[64];
  unsigned long long ballot_acc = 0, ballot_res = 0;
};

struct Worker {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  Fiber* cur = nullptr;
  int live = 0;
  int barrier_arrived = 0;
  unsigned barrier_gen = 0;
  std::vector<char> smem;
  const std::function<void()>* body = nullptr;
};

static thread_local Worker* tw = nullptr;

Ctx& cur() { return tw->cur->ctx; }
char* dyn_smem() { return tw->smem.data(); }
int lane_id() { return tw->cur->lane; }

static void yield() { swapcontext(&tw->cur->uc, &tw->sched); }
void yield_now() { std::this_thread::yield(); }

void syncthreads() {
  Worker* w = tw;
  const unsigned g = w->barrier_gen;
  if (++w->barrier_arrived >= w->live) {
    w->barrier_arrived = 0;
    ++w->barrier_gen;
    return;
  }
  while (w->barrier_gen == g) yield();
}

static void wave_barrier(Wave& wv) {
  const unsigned g = wv.gen;
  if (++wv.arrived >= wv.live) {
    wv.arrived = 0;
    ++wv.gen;
    return;
  }
  while (wv.gen == g) yield();
}

uint32_t wave_exchange(uint32_t v, int src_lane) {
  Fiber* f = tw->cur;
  Wave& wv = tw->waves[f->wave];
  wv.slot[f->lane] = v;
  wave_barrier(wv);
  const uint32_t r = (src_lane >= 0 && src_lane < 64) ? wv.slot[src_lane] : v;
  wave_barrier(wv);
  return r;
}

void wave_gather2(float a, float b, float* A64, float* B64) {
  Fiber* f = tw->cur;
  Wave& wv = tw->waves[f->wave];
  wv.fa[f->lane] = a;
  wv.fb[f->lane] = b;
  wave_barrier(wv);
  for (int i = 0; i < 64; ++i) { A64[i] = wv.fa[i]; B64[i] = wv.fb[i]; }
  wave_barrier(wv);
}

unsigned long long wave_ballot(bool p) {
  Fiber* f = tw->cur;
  Wave& wv = tw->waves[f->wave];
  if (p) wv.ballot_acc |= (1ull << f->lane);
  const unsigned g = wv.gen;
  if (++wv.arrived >= wv.live) {
    wv.ballot_res = wv.ballot_acc;
    wv.ballot_acc = 0;
    wv.arrived = 0;
    ++wv.gen;
  } else {
    while (wv.gen == g) yield();
  }
  const unsigned long long r = wv.ballot_res;
  wave_barrier(wv);
  return r;
}

static void trampoline() {
  Worker* w = tw;
  Fiber* f = w->cur;
  (*w->body)();
  f->done = true;
  --w->live;
  Wave& wv = w->waves[f->wave];
  --wv.live;
  // a finished work-item no longer counts at rendezvous points it never reached
  if (wv.live > 0 && wv.arrived >= wv.live) { wv.arrived = 0; ++wv.gen; }
  if (w->live > 0 && w->barrier_arrived >= w->live) { w->barrier_arrived = 0; ++w->barrier_gen; }
  swapcontext(&f->uc, &w->sched);
}

static void run_block(Worker& w, dim3 grid, dim3 block, dim3 bid, size_t smem, const std::function<void()>& body) {
  const int n = (int)(block.x * block.y * block.z);
  if ((int)w.fibers.size() < n) {
    const size_t old = w.fibers.size();
    w.fibers.resize(n);
    for (size_t i = 0; i < (size_t)n; ++i) {
      if (i < old && w.fibers[i].stack) continue;
      w.fibers[i].stack = (char*)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    }
  }
  const int nw = (n + 63) / 64;
  w.waves.assign(nw, Wave());
  w.smem.assign(smem + 64, 0);
  w.live = n;
  w.barrier_arrived = 0;
  w.body = &body;
  for (int i = 0; i < n; ++i) {
    Fiber& f = w.fibers[i];
    f.done = false;
    f.wave = i / 64;
    f.lane = i % 64;
    w.waves[f.wave].live++;
    f.ctx.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    f.ctx.bid = bid;
    f.ctx.bdim = block;
    f.ctx.gdim = grid;
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = f.stack;
    f.uc.uc_stack.ss_size = kStack;
    f.uc.uc_link = nullptr;
    makecontext(&f.uc, (void (*)())trampoline, 0);
  }
  while (w.live > 0) {
    for (int i = 0; i < n; ++i) {
      Fiber& f = w.fibers[i];
      if (f.done) continue;
      w.cur = &f;
      swapcontext(&w.sched, &f.uc);
    }
  }
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
  const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  unsigned nthreads = std::thread::hardware_concurrency();
  if (const char* e = getenv("HOSTSIM_THREADS")) nthreads = (unsigned)atoi(e);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > nblocks) nthreads = (unsigned)nblocks;
  std::atomic<size_t> next{0};
  auto work = [&]() {
    Worker w;
    tw = &w;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
      run_block(w, grid, block, bid, smem, body);
    }
    for (auto& f : w.fibers)
      if (f.stack) munmap(f.stack, kStack);
    tw = nullptr;
  };
  if (nthreads == 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nthreads; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
}

}  // namespace hostsim

// The RCCL wrapper (csrc/comm.hip) has no kernels and is not emulated: its entry points exist so that the binding loads, and fail.
extern "C" {
const char* hifihr_comm_last_error(void) { return "hostsim: no RCCL"; }
int hifihr_comm_get_unique_id(void*) { return -2; }
int hifihr_comm_init(void**, int, int, const void*) { return -2; }
int hifihr_comm_allreduce_f32(void*, float*, size_t, void*) { return -2; }
int hifihr_comm_broadcast_f32(void*, float*, size_t, int, void*) { return -2; }
int hifihr_comm_destroy(void*) { return 0; }
}
