// Optional per-launch timing of the dominant kernels with HIP events recorded on the launch stream
// (bench.py's roofline leg).  Disabled by default: no events are created or recorded.
#include <vector>

#include "dk_kernels.h"

namespace {
struct Rec { hipEvent_t a, b; int cls; double work; };
std::vector<Rec> g_pool;
size_t g_used = 0;
bool g_on = false;
bool g_open = false;
size_t g_dropped = 0;  // launches that found no event pair (creation failed or the hard cap below): dk_profile_read fails then
const size_t kPool = size_t(1) << 22;  // grows on demand (a 20-image FLUX replay records ~18 000 launches); the cap only bounds a runaway
}  // namespace

void dk_prof_begin(int cls, double work, hipStream_t st) {
  if (!g_on) return;
  if (g_used >= kPool) { ++g_dropped; return; }
  if (g_pool.size() <= g_used) {
    Rec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) { ++g_dropped; return; }
    g_pool.push_back(r);
  }
  g_pool[g_used].cls = cls;
  g_pool[g_used].work = work;
  (void)hipEventRecord(g_pool[g_used].a, st);
  g_open = true;
}
void dk_prof_end(hipStream_t st) {
  if (!g_open) return;
  (void)hipEventRecord(g_pool[g_used].b, st);
  ++g_used;
  g_open = false;
}

extern "C" int dk_profile_enable(int32_t on) {
  g_on = on != 0;
  g_used = 0;
  g_open = false;
  g_dropped = 0;
  return 0;
}
extern "C" int dk_profile_read(int32_t cls, double* total_ms, double* total_work, int64_t* launches) {
  DK_REQUIRE(total_ms && total_work && launches, "null argument");
  DK_REQUIRE(g_dropped == 0, "profile: launches were not recorded (event pool exhausted); the totals would under-report");
  double ms = 0.0, work = 0.0;
  int64_t n = 0;
  for (size_t i = 0; i < g_used; ++i) {
    if (g_pool[i].cls != cls) continue;
    DK_CHECK_HIP(hipEventSynchronize(g_pool[i].b));
    float e = 0.f;
    DK_CHECK_HIP(hipEventElapsedTime(&e, g_pool[i].a, g_pool[i].b));
    ms += e;
    work += g_pool[i].work;
    ++n;
  }
  *total_ms = ms;
  *total_work = work;
  *launches = n;
  return 0;
}
