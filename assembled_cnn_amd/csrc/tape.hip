// Launch tape: a training step written down once as the list of its device operations -- kernel launches with their
// arguments, cross-stream joins, the two fills of the strided 1x1 input gradient -- and issued again by ONE C call per step.
//
// Why (round 4).  The eager step costs the host 14 - 15 ms of Python per step (930 launches) against 25 ms of GPU time.  That
// head-room disappears on a busy host: the GPU boxes of this pool are shared (256 hardware threads, load averages of 40 -
// 65), and when the Python thread gets half a core the step becomes HOST-bound -- 30 ms with one competitor on the core,
// 48 ms with two (tools/debug/host_contention.sh); single-stream steps, which need fewer host calls, degrade later, which
// is what rounds 3 - 4 first took for a stream-placement problem.  ROCm 7's hipGraphLaunch does not help: it spends 11.5 - 22
// ms of host time on the step's nodes.  The tape replays the same launches through hipLaunchKernel in a C loop: ~2 ms.
//
// Recording is per host thread and piggy-backs on the launches themselves (common.h: asm_launch): between asm_tape_begin and
// asm_tape_end every ASM_LAUNCH of this thread appends {kernel, grid, block, LDS bytes, stream, argument copy} before it
// launches as usual, and every asm_stream_join appends {waiting stream, signalling stream, event}.  The host layer records
// while it captures the step into a HIP graph (train.Trainer.capture): the capture supplies what a replay needs from the
// allocator -- a private pool whose blocks keep their addresses and are not handed out again across streams -- and the
// tape supplies the launches.  asm_tape_mark cuts the tape into segments so that the host can do something between them
// (hand a gradient bucket to RCCL).
#include "common.h"

#include <string.h>

#include <memory>
#include <mutex>
#include <vector>

namespace {

enum NodeKind { N_LAUNCH = 0, N_JOIN = 1, N_MEMSET = 2, N_MEMCPY = 3 };

struct Node {
  int kind;
  const void* fn;
  dim3 grid, block;
  unsigned shmem;
  hipStream_t stream;   // launch / fill: where it runs; join: the stream that waits
  hipStream_t src;      // join: the stream waited for
  hipEvent_t ev;        // join
  size_t arg0;          // launch: first entry of this launch in Tape::argptr
  int nargs;
  void* dst;            // fills
  const void* from;
  size_t bytes;
  int value;
};

struct Tape {
  std::vector<Node> nodes;
  std::vector<char> blob;        // argument copies, each at its natural alignment
  std::vector<size_t> argoff;    // per argument: offset into blob while recording ...
  std::vector<void*> argptr;     // ... pointer into blob once the tape is closed
  std::vector<size_t> seg;       // seg[s] = first node of segment s; seg.back() = nodes.size() once closed
  size_t launches = 0, joins = 0, fills = 0;
  bool closed = false;
  int device = -1;               // the device that was current while the tape was recorded: its streams and events live there
  void destroy_events() {     // asm_tape_free only: never from a static destructor, where the HIP runtime may be gone already
    for (Node& n : nodes)
      if (n.kind == N_JOIN && n.ev) {
        (void)hipEventDestroy(n.ev);
        n.ev = nullptr;
      }
  }
};

std::mutex g_mu;
// tape id = index + 1; a freed slot holds nullptr.  Heap-allocated and never destroyed: tapes that are still alive at
// process exit are simply abandoned.
std::vector<std::unique_ptr<Tape>>& g_tapes = *new std::vector<std::unique_ptr<Tape>>();
thread_local Tape* t_rec = nullptr;

// events of the eager asm_stream_join (outside a recording): a wait refers to the record that precedes it, so re-recording
// an event that an earlier, still pending wait used is harmless -- a small ring only bounds the number of live events
// One ring per DEVICE (events belong to the device that was current when they were created; recording one on a stream of
// another device fails with an invalid handle): a thread that drives two GPUs gets two rings.
constexpr int RING = 64;
struct EventRing {
  hipEvent_t ev[RING];
  int n = 0, i = 0;
};
thread_local EventRing t_rings[ASM_MAX_DEVICES];

Tape* find(int id) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (id < 1 || (size_t)id > g_tapes.size()) return nullptr;
  return g_tapes[id - 1].get();
}

}  // namespace

void asm_tape_add_launch(const void* fn, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, void** args,
                         const size_t* sizes, const size_t* aligns, int nargs) {
  Tape* t = t_rec;
  if (!t) return;
  Node n = {};
  n.kind = N_LAUNCH;
  n.fn = fn;
  n.grid = grid;
  n.block = block;
  n.shmem = shmem;
  n.stream = stream;
  n.arg0 = t->argoff.size();
  n.nargs = nargs;
  for (int i = 0; i < nargs; ++i) {
    const size_t al = aligns[i] ? aligns[i] : 1;
    const size_t off = (t->blob.size() + al - 1) / al * al;
    t->blob.resize(off + sizes[i]);
    memcpy(t->blob.data() + off, args[i], sizes[i]);
    t->argoff.push_back(off);
  }
  t->nodes.push_back(n);
  ++t->launches;
}

// The fills of the strided 1x1 input gradient (conv_igemm.hip) go through here so that a tape sees them.
hipError_t asm_fill_async(void* dst, const void* from, int value, size_t bytes, hipStream_t stream) {
  if (Tape* t = t_rec) {
    Node n = {};
    n.kind = from ? N_MEMCPY : N_MEMSET;
    n.stream = stream;
    n.dst = dst;
    n.from = from;
    n.bytes = bytes;
    n.value = value;
    t->nodes.push_back(n);
    ++t->fills;
  }
  return from ? hipMemcpyAsync(dst, from, bytes, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(dst, value, bytes, stream);
}

extern "C" int asm_memcpy_async(void* dst, const void* src, size_t bytes, void* stream) {
  ASM_REQUIRE(dst && src, "memcpy_async: null pointer");
  if (bytes == 0) return ASM_OK;
  if (hipError_t e = asm_fill_async(dst, src, 0, bytes, (hipStream_t)stream); e != hipSuccess)
    ASM_FAIL(ASM_EHIP, "memcpy_async: %s", hipGetErrorString(e));
  return ASM_OK;
}

extern "C" int asm_stream_join(void* dst, void* src) {
  if (dst == src) return ASM_OK;
  hipEvent_t ev = nullptr;
  Tape* t = t_rec;
  if (t) {
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ASM_FAIL(ASM_EHIP, "stream_join: hipEventCreate failed");
    Node n = {};
    n.kind = N_JOIN;
    n.stream = (hipStream_t)dst;
    n.src = (hipStream_t)src;
    n.ev = ev;
    t->nodes.push_back(n);
    ++t->joins;
  } else {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ASM_MAX_DEVICES) ASM_FAIL(ASM_EHIP, "stream_join: no current device");
    EventRing& r = t_rings[dev];
    if (r.n < RING) {
      if (hipEventCreateWithFlags(&r.ev[r.n], hipEventDisableTiming) != hipSuccess)
        ASM_FAIL(ASM_EHIP, "stream_join: hipEventCreate failed");
      ++r.n;
    }
    ev = r.ev[r.i];
    r.i = (r.i + 1) % r.n;
  }
  hipError_t e = hipEventRecord(ev, (hipStream_t)src);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)dst, ev, 0);
  if (e != hipSuccess) ASM_FAIL(ASM_EHIP, "stream_join: %s", hipGetErrorString(e));
  return ASM_OK;
}

extern "C" int asm_tape_begin(void) {
  if (t_rec) ASM_FAIL(ASM_EINVAL, "tape_begin: this thread is already recording a tape");
  std::lock_guard<std::mutex> lk(g_mu);
  g_tapes.emplace_back(new Tape());
  t_rec = g_tapes.back().get();
  t_rec->seg.push_back(0);
  if (hipGetDevice(&t_rec->device) != hipSuccess) t_rec->device = -1;
  (void)hipGetLastError();
  asm_tape_on = true;
  return (int)g_tapes.size();
}

extern "C" int asm_tape_mark(void) {
  if (!t_rec) ASM_FAIL(ASM_EINVAL, "tape_mark: no tape is being recorded by this thread");
  t_rec->seg.push_back(t_rec->nodes.size());
  return (int)t_rec->seg.size() - 1;
}

extern "C" int asm_tape_end(void) {
  Tape* t = t_rec;
  if (!t) ASM_FAIL(ASM_EINVAL, "tape_end: no tape is being recorded by this thread");
  t_rec = nullptr;
  asm_tape_on = false;
  t->argptr.resize(t->argoff.size());
  for (size_t i = 0; i < t->argoff.size(); ++i) t->argptr[i] = t->blob.data() + t->argoff[i];
  t->seg.push_back(t->nodes.size());
  t->closed = true;
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < g_tapes.size(); ++i)
    if (g_tapes[i].get() == t) return (int)i + 1;
  ASM_FAIL(ASM_EINVAL, "tape_end: the tape being recorded was freed");
}

extern "C" int asm_tape_info(int tape, int64_t info[6]) {
  Tape* t = find(tape);
  if (!t || !info) ASM_FAIL(ASM_EINVAL, "tape_info: no such tape (%d)", tape);
  info[0] = (int64_t)t->nodes.size();
  info[1] = (int64_t)t->launches;
  info[2] = (int64_t)t->joins;
  info[3] = (int64_t)t->fills;
  info[4] = (int64_t)t->seg.size() - (t->closed ? 1 : 0);
  info[5] = (int64_t)t->blob.size();
  return ASM_OK;
}

extern "C" int asm_tape_replay(int tape, int segment) {
  Tape* t = find(tape);
  if (!t || !t->closed) ASM_FAIL(ASM_EINVAL, "tape_replay: no such closed tape (%d)", tape);
  if (t == t_rec) ASM_FAIL(ASM_EINVAL, "tape_replay: the tape is being recorded");
  const int nseg = (int)t->seg.size() - 1;
  if (segment >= nseg) ASM_FAIL(ASM_EINVAL, "tape_replay: segment %d of %d", segment, nseg);
  if (t->device >= 0) {       // (no device while it was recorded: an empty host-only tape)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != t->device)
      ASM_FAIL(ASM_EINVAL, "tape_replay: tape %d was recorded on device %d, the current device is %d", tape, t->device, dev);
  }
  const size_t i0 = segment < 0 ? 0 : t->seg[segment], i1 = segment < 0 ? t->nodes.size() : t->seg[segment + 1];
  (void)hipGetLastError();
  for (size_t i = i0; i < i1; ++i) {
    const Node& n = t->nodes[i];
    hipError_t e;
    switch (n.kind) {
      case N_LAUNCH:
        asm_count_launch();
        e = hipLaunchKernel(n.fn, n.grid, n.block, t->argptr.data() + n.arg0, n.shmem, n.stream);
        break;
      case N_JOIN:
        e = hipEventRecord(n.ev, n.src);
        if (e == hipSuccess) e = hipStreamWaitEvent(n.stream, n.ev, 0);
        break;
      case N_MEMSET:
        e = hipMemsetAsync(n.dst, n.value, n.bytes, n.stream);
        break;
      default:
        e = hipMemcpyAsync(n.dst, n.from, n.bytes, hipMemcpyDeviceToDevice, n.stream);
        break;
    }
    if (e != hipSuccess) ASM_FAIL(ASM_EHIP, "tape_replay: node %zu of tape %d: %s", i, tape, hipGetErrorString(e));
  }
  return ASM_OK;
}

extern "C" int asm_tape_free(int tape) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (tape < 1 || (size_t)tape > g_tapes.size() || !g_tapes[tape - 1]) ASM_FAIL(ASM_EINVAL, "tape_free: no such tape (%d)", tape);
  if (g_tapes[tape - 1].get() == t_rec) {
    t_rec = nullptr;
    asm_tape_on = false;
  }
  g_tapes[tape - 1]->destroy_events();
  g_tapes[tape - 1].reset();
  return ASM_OK;
}
