// Launch lists behind the C ABI (include/ctunet_hip.h: ctu_plan_create / ctu_plan_run / ctu_plan_destroy).
//
// A plan is a recorded sequence of calls of this library's own entry points - the argument blocks of one module's forward
// (or backward) pass, e.g. the nine launches of a ResNet bottleneck (reference: networks/resnet.py:106-126) - replayed by ONE
// call from the host language instead of one FFI crossing per kernel.  The host records once (hybrid-ctunet_amd/_plan.py)
// and hands over, per replay, a table of "slot" values: the device pointers of this call's tensors (and the odd integer).
// Replay = patch the slots into the argument words, then call the entry points in order.  No kernel is launched
// differently from a direct call; options read per launch (ctu_set_option) keep working.
//
// Commands carry a stream index into the table of HIP streams given to ctu_plan_run; EVENT_RECORD / STREAM_WAIT commands on
// events owned by the plan order streams against each other (weight-gradient kernels on a companion stream behind the
// data-gradient chain), which keeps the multi-stream schedule a HIP graph replay loses on ROCm 7.2.
#include <string.h>

#include <vector>

#include "common.h"

namespace {
struct Patch {
  uint64_t pos, slot, off;
};
struct Plan {
  std::vector<uint64_t> words;
  std::vector<Patch> patches;
  std::vector<hipEvent_t> events;
  int nslots;
};
constexpr uint64_t OP_EVENT_RECORD = 1000, OP_STREAM_WAIT = 1001;
inline float plan_f32(uint64_t w) {
  const uint32_t b = (uint32_t)w;
  float f;
  memcpy(&f, &b, 4);
  return f;
}
static_assert(sizeof(ctu_geom) == 80 && sizeof(ctu_epilogue) == 112 && sizeof(ctu_attn_geom) == 36,
              "inline struct sizes are part of the plan format (hybrid-ctunet_amd/_plan.py STRUCT_WORDS)");
}  // namespace

extern "C" int ctu_plan_create(const uint64_t* words, int64_t nwords, const uint64_t* patches, int64_t npatches,
                               int32_t nevents, int32_t nslots, void** handle) {
  CTU_REQUIRE(words && handle && nwords > 0 && npatches >= 0 && nevents >= 0 && nslots >= 0, "plan_create: bad arguments");
  Plan* p = new Plan();
  p->words.assign(words, words + nwords);
  p->nslots = nslots;
  for (int64_t i = 0; i < npatches; ++i) {
    Patch q{patches[3 * i], patches[3 * i + 1], patches[3 * i + 2]};
    if (q.pos >= (uint64_t)nwords || q.slot >= (uint64_t)nslots) {
      delete p;
      ctu_set_error("plan_create: patch %lld out of range", (long long)i);
      return CTU_ERR_ARG;
    }
    p->patches.push_back(q);
  }
  // walk the command stream once: every command must lie inside the blob
  for (int64_t i = 0; i < nwords;) {
    if (i + 3 > nwords || i + 3 + (int64_t)p->words[i + 2] > nwords) {
      delete p;
      ctu_set_error("plan_create: truncated command at word %lld", (long long)i);
      return CTU_ERR_ARG;
    }
    i += 3 + (int64_t)p->words[i + 2];
  }
  for (int i = 0; i < nevents; ++i) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      for (hipEvent_t d : p->events) (void)hipEventDestroy(d);
      delete p;
      ctu_set_error("plan_create: hipEventCreate failed");
      return CTU_ERR_LAUNCH;
    }
    p->events.push_back(e);
  }
  *handle = p;
  return CTU_OK;
}

extern "C" int ctu_plan_destroy(void* handle) {
  Plan* p = reinterpret_cast<Plan*>(handle);
  if (!p) return CTU_OK;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  delete p;
  return CTU_OK;
}

extern "C" int ctu_plan_run(void* handle, const uint64_t* slots, int32_t nslots, void* const* streams, int32_t nstreams) {
  Plan* p = reinterpret_cast<Plan*>(handle);
  CTU_REQUIRE(p && slots && streams && nslots == p->nslots && nstreams > 0, "plan_run: bad arguments");
  uint64_t* W = p->words.data();
  for (const Patch& q : p->patches) W[q.pos] = slots[q.slot] + q.off;
  const int64_t n = (int64_t)p->words.size();
  int cmd = 0;
  for (int64_t i = 0; i < n; ++cmd) {
    const uint64_t op = W[i], sidx = W[i + 1], na = W[i + 2];
    const uint64_t* w = W + i + 3;
    i += 3 + (int64_t)na;
    if (sidx >= (uint64_t)nstreams) {
      ctu_set_error("plan_run: command %d wants stream %llu of %d", cmd, (unsigned long long)sidx, nstreams);
      return CTU_ERR_ARG;
    }
    ctu_stream_t st = streams[sidx];
    int rc = CTU_OK;
    const char* what = "?";
    switch (op) {
      case OP_EVENT_RECORD:
        if (w[0] >= p->events.size() || hipEventRecord(p->events[w[0]], (hipStream_t)st) != hipSuccess) rc = CTU_ERR_LAUNCH;
        what = "event_record";
        break;
      case OP_STREAM_WAIT:
        if (w[0] >= p->events.size() || hipStreamWaitEvent((hipStream_t)st, p->events[w[0]], 0) != hipSuccess) rc = CTU_ERR_LAUNCH;
        what = "stream_wait";
        break;
#include "plan_dispatch.inc"
      default:
        ctu_set_error("plan_run: unknown opcode %llu at command %d", (unsigned long long)op, cmd);
        return CTU_ERR_ARG;
    }
    if (rc != CTU_OK) {
      // (the entry point has set its own message; prefix where in the list it happened)
      char msg[400];
      snprintf(msg, sizeof(msg), "%s", ctu_last_error());
      ctu_set_error("plan command %d (%s): %s", cmd, what, msg);
      return rc;
    }
  }
  return CTU_OK;
}
