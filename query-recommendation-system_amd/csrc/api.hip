// api.hip -- library-wide plumbing of libqrlsh: version, thread-local error text, and an
// optional per-kernel profiler built on HIP events recorded on the launch stream.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void qrlsh_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

QRLSH_EXPORT int qrlsh_version(void) { return 1; }
QRLSH_EXPORT const char *qrlsh_last_error(void) { return g_err; }
QRLSH_EXPORT uint64_t qrlsh_mix64_host(uint64_t x) { return qr_mix64(x); }

// ---- profiler ---------------------------------------------------------------------------
// When enabled, every kernel launch of the library is bracketed by two events on its own
// stream; qrlsh_prof_report() synchronises them and sums hipEventElapsedTime per label.
namespace {
struct Rec { const char *label; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

int qr_prof_begin(const char *label, hipStream_t st) {
  if (!g_on) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{label, get_event(), get_event()};
  if (!r.a || !r.b) return -1;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void qr_prof_end(int slot, hipStream_t st) {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].b, st);
}

bool qr_prof_active() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_on;
}

// ---- auxiliary stream for intra-call overlap ------------------------------------------------
// A few entry points run independent pieces of their work (band groups of the bucket path) on a second stream
// beside the caller's, so that a latency-bound kernel of one piece shares the device with a bandwidth-bound
// kernel of the next.  Fork: the auxiliary stream waits for everything already queued on the caller's
// stream; join: the caller's stream waits for the auxiliary one -- to the caller the entry point is still one
// asynchronous call on its stream.  One auxiliary stream and two events PER DEVICE, created on first use: the overlap is
// per device, not per caller stream -- concurrent callers on one device (the library's contract is one host thread
// per device) would be serialised on it and see each other's work as dependencies; correct, never faster.
namespace {
struct Aux { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false, tried = false; };
Aux g_aux[16];
int g_overlap = -1;  // -1: read QRLSH_OVERLAP on first use (default on)
}  // namespace

QRLSH_EXPORT int qrlsh_set_overlap(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_overlap = on != 0;
  return QRLSH_OK;
}

hipStream_t qr_aux_fork(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_overlap < 0) {
    const char *e = getenv("QRLSH_OVERLAP");
    g_overlap = !(e && e[0] == '0');
  }
  if (!g_overlap || g_on) return nullptr;  // (profiled steps run serially: per-kernel times stay per-kernel)
  Aux &a = g_aux[dev];
  if (!a.tried) {
    a.tried = true;
    a.ok = hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&a.join, hipEventDisableTiming) == hipSuccess;
  }
  if (!a.ok) return nullptr;
  if (hipEventRecord(a.fork, st) != hipSuccess || hipStreamWaitEvent(a.s, a.fork, 0) != hipSuccess) return nullptr;
  return a.s;
}

// 0, or QRLSH_EHIP when the caller's stream could not be ordered after the auxiliary one: the auxiliary stream is then
// drained on the host (whatever ran there is complete before the entry point returns) and the error is reported --
// the call never returns OK with work of its own still unordered.
int qr_aux_join(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return QRLSH_EHIP;
  std::lock_guard<std::mutex> lk(g_mu);
  Aux &a = g_aux[dev];
  if (!a.ok) return QRLSH_OK;   // never forked
  if (hipEventRecord(a.join, a.s) == hipSuccess && hipStreamWaitEvent(st, a.join, 0) == hipSuccess) return QRLSH_OK;
  (void)hipStreamSynchronize(a.s);
  qrlsh_set_error("auxiliary stream could not be joined to the caller's stream");
  return QRLSH_EHIP;
}

QRLSH_EXPORT int qrlsh_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &r : g_recs) {
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_on = on != 0;
  return QRLSH_OK;
}

// stop (1) / resume (0) bracketing without discarding what has been recorded: lets a caller
// sample some iterations of a timed loop instead of slowing all of them down
QRLSH_EXPORT int qrlsh_prof_pause(int paused) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = paused == 0;
  return QRLSH_OK;
}

// Writes "label count total_ms\n" lines (one per label) into buf; returns the number of
// labels, or a negative error.  Blocks until the recorded events have completed.
QRLSH_EXPORT int qrlsh_prof_report(char *buf, size_t buflen) {
  std::lock_guard<std::mutex> lk(g_mu);
  struct Acc { const char *label; long count; double ms; };
  std::vector<Acc> acc;
  for (auto &r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) {
      qrlsh_set_error("qrlsh_prof_report: hipEventSynchronize failed");
      return QRLSH_EHIP;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    bool found = false;
    for (auto &a : acc)
      if (strcmp(a.label, r.label) == 0) {
        a.count++;
        a.ms += ms;
        found = true;
        break;
      }
    if (!found) acc.push_back(Acc{r.label, 1, ms});
  }
  std::string out;
  char line[160];
  for (auto &a : acc) {
    snprintf(line, sizeof(line), "%s %ld %.6f\n", a.label, a.count, a.ms);
    out += line;
  }
  if (buf && buflen) {
    strncpy(buf, out.c_str(), buflen - 1);
    buf[buflen - 1] = 0;
  }
  return (int)acc.size();
}
