// Host-side runtime of libmvoc_hip.so: thread-local error text, launch checks, optional per-family
// HIP-event timing (used by bench.py's roofline leg; events are recorded on the launch stream).
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {
thread_local char g_err[512] = "";

struct ProfRec {
  hipEvent_t e0, e1;
  int fam;
  double work;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec*> g_prof_recs;
}  // namespace

void mvoc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mvoc_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mvoc_set_error("%s: %s", what, hipGetErrorString(e));
    return -3;
  }
  return 0;
}

MvocProfScope::MvocProfScope(int fam_, hipStream_t s, double work) : fam(fam_), stream(s), rec(nullptr) {
  if (!g_prof_on) return;
  ProfRec* r = new ProfRec;
  r->fam = fam_;
  r->work = work;
  if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) {
    delete r;
    return;
  }
  (void)hipEventRecord(r->e0, s);
  rec = r;
}

MvocProfScope::~MvocProfScope() {
  if (!rec) return;
  ProfRec* r = (ProfRec*)rec;
  (void)hipEventRecord(r->e1, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(r);
}

extern "C" int mvoc_version(void) { return MVOC_VERSION; }
extern "C" const char* mvoc_last_error(void) { return g_err; }

extern "C" int mvoc_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

extern "C" int mvoc_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (ProfRec* r : g_prof_recs) {
    (void)hipEventDestroy(r->e0);
    (void)hipEventDestroy(r->e1);
    delete r;
  }
  g_prof_recs.clear();
  return 0;
}

// Caller must have synchronised the stream(s) first.
extern "C" int mvoc_prof_collect(double* ms, int64_t* launches, double* work) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < MVOC_FAM_COUNT; ++i) {
    if (ms) ms[i] = 0;
    if (launches) launches[i] = 0;
    if (work) work[i] = 0;
  }
  for (ProfRec* r : g_prof_recs) {
    float t = 0;
    if (hipEventElapsedTime(&t, r->e0, r->e1) != hipSuccess) {
      mvoc_set_error("prof_collect: event not complete (synchronise the stream first)");
      return -3;
    }
    if (ms) ms[r->fam] += t;
    if (launches) launches[r->fam] += 1;
    if (work) work[r->fam] += r->work;
  }
  return 0;
}
