// katamx_fibers.cpp — see katamx_fibers.h. SURVEY 8 row f2: the caller side of the leaf batcher.
//
// Three small pieces:
//   1. a cooperative scheduler per OS thread over ucontext fibers (FIFO: fresh fibers, parked fibers with their ticket, fibers that
//      yielded). A parked fiber at the head of the queue is what the OS thread blocks on - device batches complete in launch order,
//      so the oldest ticket is the next one to be ready;
//   2. a process-wide pool of carrier OS threads (created on demand, parked on a condition variable between searches);
//   3. performTaskOnFibers, the --wrap replacement of Search::performTaskWithThreads (searchmultithreadhelpers.cpp:77-92): the
//      same contract - task(i) runs exactly once for every i in [0, min(capThreads, numThreads)), the call returns when all have
//      returned - with K logical threads per OS thread.
// Nothing here knows about the search: the only search-side fact used is that evaluate() is called without a lock held
// (searchnnhelpers.cpp:61-135), so parking inside it cannot leave a mutex locked on a descheduled fiber.
#include "katamx_fibers.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "core/global.h"
#include "katamx_leaf.h"
#include "search/search.h"

namespace {

// The descent is recursive (one frame per ply, search.cpp:1440, with Board copies and per-move arrays in it) and normally runs on an
// 8 MiB thread stack: a fiber gets the same (KATAMX_FIBER_STACK_MB overrides, 1..64). The mapping is MAP_NORESERVE and pages are
// committed on touch, so the size costs address space only. Below it sits a guard page: running into it is a SIGSEGV, not silent
// corruption of a neighbour's stack - and a guard that cannot be installed is an error, not a warning.
constexpr size_t GUARD_BYTES = 1u << 12;
size_t stackBytes() {
  static const size_t bytes = [] {
    const char* e = getenv("KATAMX_FIBER_STACK_MB");
    long mb = e ? atol(e) : 8;
    mb = mb < 1 ? 1 : mb > 64 ? 64 : mb;
    return (size_t)mb << 20;
  }();
  return bytes;
}

std::atomic<uint64_t> gFibersRun{0}, gParks{0}, gBlockingWaits{0};

// ---- stacks: mapped once, reused by later searches ---------------------------------------------------------------------
struct StackPool {
  std::mutex mutex;
  std::vector<char*> free;
  char* get() {
    {
      std::lock_guard<std::mutex> lock(mutex);
      if(!free.empty()) {
        char* s = free.back();
        free.pop_back();
        return s;
      }
    }
    void* p = mmap(NULL, stackBytes() + GUARD_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK | MAP_NORESERVE, -1, 0);
    if(p == MAP_FAILED)
      throw StringError("katamx fibers: cannot map a fiber stack");
    if(mprotect(p, GUARD_BYTES, PROT_NONE) != 0) {  // the stack grows down towards it
      munmap(p, stackBytes() + GUARD_BYTES);
      throw StringError("katamx fibers: cannot protect the guard page of a fiber stack");
    }
    return (char*)p;
  }
  void put(char* s) {
    std::lock_guard<std::mutex> lock(mutex);
    free.push_back(s);
  }
};
StackPool& stackPool() {
  static StackPool* pool = new StackPool();  // never destroyed: carrier threads may outlive static destruction
  return *pool;
}

struct Sched;
struct Fiber {
  ucontext_t ctx;
  char* stack = NULL;
  Sched* sched = NULL;
  int logicalIdx = 0;
  bool done = false;
  bool yielded = false;    // sits in the queue because it yielded (waitForNextNNEvalIfAny), not because it is new or parked on a ticket
  bool hasTicket = false;
  KatamxLeaf::Port* port = NULL;
  uint64_t ticket = 0;
  std::exception_ptr waitError, taskError;
};
struct Sched {
  ucontext_t mainCtx;
  std::function<void(int)>* task = NULL;
  std::deque<Fiber*> queue;
  Fiber* current = NULL;
};
thread_local Sched* tlsSched = NULL;

void trampoline(unsigned lo, unsigned hi) {
  Fiber* f = (Fiber*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
  try {
    (*f->sched->task)(f->logicalIdx);
  }
  catch(...) {
    f->taskError = std::current_exception();
  }
  f->done = true;
  swapcontext(&f->ctx, &f->sched->mainCtx);
  std::abort();  // a finished fiber is never resumed
}

// ---- carrier threads -------------------------------------------------------------------------------------------------------
struct CarrierPool {
  std::mutex mutex;
  std::condition_variable wake;
  std::deque<std::function<void()>*> jobs;
  int idle = 0, threads = 0;
  void worker() {
    std::unique_lock<std::mutex> lock(mutex);
    while(true) {
      while(jobs.empty()) {
        idle++;
        wake.wait(lock);
        idle--;
      }
      std::function<void()>* job = jobs.front();
      jobs.pop_front();
      lock.unlock();
      (*job)();
      lock.lock();
    }
  }
  void post(std::function<void()>* job) {
    std::lock_guard<std::mutex> lock(mutex);
    jobs.push_back(job);
    if((int)jobs.size() > idle) {
      std::thread(&CarrierPool::worker, this).detach();
      threads++;
    }
    wake.notify_one();
  }
};
CarrierPool& carriers() {
  static CarrierPool* pool = new CarrierPool();
  return *pool;
}

}  // namespace

int KatamxFibers::leavesPerThread() {
  static const int k = [] {
    const char* e = getenv("KATAMX_LEAVES_PER_THREAD");
    const int v = e ? atoi(e) : 1;
    if(getenv("KATAMX_FIBER_STATS") != NULL)  // one line on stderr at exit: what the fibers did (tests)
      atexit([] {
        fprintf(stderr, "katamx fibers: %llu fibers run, %llu parks, %llu blocking waits, %d carrier threads\n",
                (unsigned long long)gFibersRun.load(), (unsigned long long)gParks.load(), (unsigned long long)gBlockingWaits.load(), carriers().threads);
      });
    return v < 1 ? 1 : v > 64 ? 64 : v;
  }();
  return k;
}

bool KatamxFibers::onFiber() {
  const Sched* s = tlsSched;
  return s != NULL && s->current != NULL;
}

bool KatamxFibers::park(KatamxLeaf::Port* port, uint64_t ticket) {
  Sched* s = tlsSched;
  if(s == NULL || s->current == NULL)
    return false;
  Fiber* f = s->current;
  f->port = port;
  f->ticket = ticket;
  f->hasTicket = true;
  s->queue.push_back(f);
  gParks.fetch_add(1, std::memory_order_relaxed);
  swapcontext(&f->ctx, &s->mainCtx);
  // resumed: the scheduler has collected the ticket
  if(f->waitError) {
    std::exception_ptr e = f->waitError;
    f->waitError = nullptr;
    std::rethrow_exception(e);
  }
  return true;
}

bool KatamxFibers::yieldToOthers() {
  Sched* s = tlsSched;
  if(s == NULL || s->current == NULL || s->queue.empty())
    return false;
  // Someone to yield TO: a fiber that has not started yet or one parked on a ticket. When every queued fiber is itself a yielder,
  // they are all waiting for evaluations of OTHER OS threads: passing the thread round among them would spin hot - the caller
  // then sleeps on the evaluator's completion count instead (as the reference's waitForNextNNEvalIfAny does).
  bool useful = false;
  for(const Fiber* q : s->queue)
    useful = useful || !q->yielded;
  if(!useful)
    return false;
  Fiber* f = s->current;
  f->yielded = true;
  s->queue.push_back(f);
  swapcontext(&f->ctx, &s->mainCtx);
  f->yielded = false;
  return true;
}

void KatamxFibers::runOnFibers(std::function<void(int)>* task, const int* indices, int count) {
  if(count <= 0)
    return;
  if(tlsSched != NULL && tlsSched->current != NULL) {
    // a task that fans out again from inside a fiber: run the inner indices in place
    for(int i = 0; i < count; i++)
      (*task)(indices[i]);
    return;
  }
  Sched sched;
  sched.task = task;
  std::vector<std::unique_ptr<Fiber>> fibers;
  for(int i = 0; i < count; i++) {
    std::unique_ptr<Fiber> f(new Fiber());
    f->sched = &sched;
    f->logicalIdx = indices[i];
    f->stack = stackPool().get();
    getcontext(&f->ctx);
    f->ctx.uc_stack.ss_sp = f->stack + GUARD_BYTES;
    f->ctx.uc_stack.ss_size = stackBytes();
    f->ctx.uc_link = NULL;
    const uintptr_t p = (uintptr_t)f.get();
    makecontext(&f->ctx, (void (*)())trampoline, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    sched.queue.push_back(f.get());
    fibers.push_back(std::move(f));
  }
  gFibersRun.fetch_add((uint64_t)count, std::memory_order_relaxed);
  Sched* const outer = tlsSched;
  tlsSched = &sched;
  while(!sched.queue.empty()) {
    Fiber* f = sched.queue.front();
    sched.queue.pop_front();
    if(f->hasTicket) {
      // every other fiber of this thread is behind f in the queue: parked later, or not able to run before f has
      try {
        KatamxLeaf::wait(f->port, f->ticket);
      }
      catch(...) {
        f->waitError = std::current_exception();
      }
      f->hasTicket = false;
      gBlockingWaits.fetch_add(1, std::memory_order_relaxed);
    }
    sched.current = f;
    swapcontext(&sched.mainCtx, &f->ctx);
    sched.current = NULL;
    if(f->done) {
      stackPool().put(f->stack);
      f->stack = NULL;
    }
  }
  tlsSched = outer;
  for(const std::unique_ptr<Fiber>& f : fibers)
    if(f->taskError)
      std::rethrow_exception(f->taskError);
}

void KatamxFibers::counters(uint64_t& fibersRun, uint64_t& parks, uint64_t& blockingWaits) {
  fibersRun = gFibersRun.load();
  parks = gParks.load();
  blockingWaits = gBlockingWaits.load();
}

// ---- the replacement of Search::performTaskWithThreads ---------------------------------------------------------------------
// (member function: `this` arrives as the first argument under the Itanium C++ ABI)
extern "C" void __real__ZN6Search22performTaskWithThreadsEPSt8functionIFviEEi(Search* self, std::function<void(int)>* task, int capThreads);

extern "C" void __wrap__ZN6Search22performTaskWithThreadsEPSt8functionIFviEEi(Search* self, std::function<void(int)>* task, int capThreads) {
  const int k = KatamxFibers::leavesPerThread();
  const int n = std::max(1, std::min(capThreads, self->searchParams.numThreads));  // logical threads, as searchmultithreadhelpers.cpp:79
  if(k <= 1 || n <= 1) {
    __real__ZN6Search22performTaskWithThreadsEPSt8functionIFviEEi(self, task, capThreads);
    return;
  }
  const int carriersWanted = (n + k - 1) / k;
  std::vector<std::vector<int>> share(carriersWanted);
  for(int i = 0; i < n; i++)
    share[i / k].push_back(i);  // logical thread 0 stays on the calling thread, as in the reference

  std::mutex doneMutex;
  std::condition_variable doneCv;
  int remaining = carriersWanted - 1;
  std::exception_ptr failure;
  std::vector<std::function<void()>> jobs(carriersWanted);
  for(int c = 1; c < carriersWanted; c++) {
    jobs[c] = [&, c]() {
      std::exception_ptr err;
      try {
        KatamxFibers::runOnFibers(task, share[c].data(), (int)share[c].size());
      }
      catch(...) {
        err = std::current_exception();
      }
      std::lock_guard<std::mutex> lock(doneMutex);
      if(err && !failure)
        failure = err;
      if(--remaining == 0)
        doneCv.notify_all();
    };
    carriers().post(&jobs[c]);
  }
  std::exception_ptr mine;
  try {
    KatamxFibers::runOnFibers(task, share[0].data(), (int)share[0].size());
  }
  catch(...) {
    mine = std::current_exception();
  }
  {
    std::unique_lock<std::mutex> lock(doneMutex);
    doneCv.wait(lock, [&] { return remaining == 0; });
  }
  if(mine)
    std::rethrow_exception(mine);
  if(failure)
    std::rethrow_exception(failure);
}
