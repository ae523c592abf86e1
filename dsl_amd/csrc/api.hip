// libdsl_hip.so: error reporting, op-list executor, hardware probes.
#include <stdarg.h>
#include <stdlib.h>

#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "common.hpp"

static thread_local char g_err[512] = "";

void dsl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dsl_last_error(void) { return g_err; }
extern "C" int dsl_version(void) { return 100; }

// ---- library options (dsl_set_option): the library reads no environment variable; a host (dsl_amd/tuning.py reads DSL_TUNE once)
// sets these before the first launch.  Values are read where the work is planned, so an option set later applies to later plans.
namespace {
struct Option { const char* name; int value; };
Option g_options[] = {
    {"wgrad_slots", 128},    // workgroup budget of a weight-gradient launch that runs beside the caller's chain (dsl_wgrad_desc.slots = 0)
    {"stream_probe", 1},     // 0: the library takes its streams as the runtime deals them instead of probing for distinct hardware queues
    {"debug_sync", 0},       // 1: drain the device after every op of dsl_run_ops and name it on stderr
    {"skip_kinds", 0},       // step-level ablation (tools/step_ablation.sh): bit mask of op kinds dsl_run_ops does not launch - timing only
    {"rla_tail_form", 0},    // dsl_rla_tail_fwd: 0 = tile form by shape, 1 / 2 = force the 14 x 14 / the 6 x 6 K-split form (measurement)
    {"comm_queue", 1},       // which hardware queue the communication stream (dsl_side_stream(5)) is PLACED on (round 6; read at side_init):
                             // 0 = any unused candidate (rounds 3 - 5), 1 = the weight-gradient stream's queue, 2 = the second chain's,
                             // 3 = the frozen prefix's, 4 = the caller's.  A fifth busy queue collapses the step, so the
                             // collectives must share one of the four.  Default 1 (round 6, profiles/r06_comm_queue_sweep.txt): with the late
                             // exchange the weight-gradient queue is idle when the collectives and updates run (under the next forward pass):
                             // 5.7 % of the step for the one-GPU proxy against 13 - 18 % on the other three; DESIGN section 6
};
}  // namespace
int dsl_option(const char* name) {
  for (const Option& o : g_options)
    if (!strcmp(o.name, name)) return o.value;
  return 0;
}
extern "C" int dsl_set_option(const char* name, int value) {
  DSL_CHECK(name != nullptr, "dsl_set_option: null name");
  for (Option& o : g_options)
    if (!strcmp(o.name, name)) { o.value = value; return 0; }
  dsl_set_error("dsl_set_option: unknown option '%s'", name);
  return -1;
}
extern "C" int dsl_get_option(const char* name, int* value) {
  DSL_CHECK(name != nullptr && value != nullptr, "dsl_get_option: null argument");
  for (const Option& o : g_options)
    if (!strcmp(o.name, name)) { *value = o.value; return 0; }
  dsl_set_error("dsl_get_option: unknown option '%s'", name);
  return -1;
}

namespace {
// The library's streams + a ring of events per device, created at the first use of any of them (host objects, no device memory).
//
// Which of a process's streams share one of the device's hardware queues decides up to 20 % of the step: the runtime
// deals streams out over FOUR queues per priority class, a fifth busy queue collapses the step (stream_prio experiments, round 5:
// 235 - 270 instead of 440 img/s with the library's streams in a class of their own), and a host framework's own streams (a
// DataLoader's copy stream, an evaluation hook - created before or after the model) take part in the deal.  The library therefore
// owns every stream it uses and PICKS them: it creates candidates and measures, with a 300 us spin kernel on one stream and an empty
// kernel on the other, which of them run concurrently with the caller's stream and with each other; the three streams that carry
// the step's concurrent work - weight gradients, second chain, frozen prefix - are three candidates on three different queues,
// none of them the caller's, whatever else the process has created.
//   ids 1..3  side streams of dsl_run_ops: 1 = weight gradients (+ the optimizer's per-bucket updates), 2 = second head tower,
//             3 = second image chain / FPN branch.  2 and 3 are ONE stream: their work never overlaps in time (+ 1 %, round 5)
//   ids 4..6  role streams handed to the host side (dsl_side_stream): 4 = pipelined frozen prefix, 5 = communication,
//             6 = asynchronous teacher sweep
constexpr int kEvRing = 256, kSide = 3, kStreams = 6;
hipStream_t g_side[16][kStreams] = {};
hipEvent_t g_ev[16][kEvRing] = {};
int g_evpos[16] = {};
bool g_init[16] = {};
int g_probe_result[16] = {};            // distinct-queue streams found by the probe (3 = all), -1 = probe off
int g_comm_queue[16] = {};              // where the communication stream was placed (option comm_queue's value; 0 = as the runtime dealt it)
hipEvent_t g_named[16][16] = {};        // DSL_OP_RECORD / DSL_OP_WAIT slots
bool g_named_set[16][16] = {};

__global__ void spin_kernel(long long ticks) {       // wall-clock ticks (100 MHz): bounded, touches no memory
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// Do streams a and b run concurrently (different hardware queues)?  b's empty kernel finishes while a's spin is still running.
bool streams_concurrent(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb) {
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 30000LL);      // 300 us
  hipEventRecord(ea, a);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 0LL);
  hipEventRecord(eb, b);
  hipEventSynchronize(eb);
  const bool a_running = hipEventQuery(ea) == hipErrorNotReady;
  hipEventSynchronize(ea);
  (void)hipGetLastError();
  return a_running;
}
void side_init(int dev, hipStream_t caller) {
  if (g_init[dev]) return;
  for (int i = 0; i < kEvRing; ++i) hipEventCreateWithFlags(&g_ev[dev][i], hipEventDisableTiming);
  for (int i = 0; i < 16; ++i) hipEventCreateWithFlags(&g_named[dev][i], hipEventDisableTiming);
  constexpr int kCand = 12, kNeed = 3;
  hipStream_t cand[kCand] = {};
  bool used[kCand] = {};
  for (int i = 0; i < kCand; ++i) {
    hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking);
    hipEventRecord(g_ev[dev][i], cand[i]);         // first use: the candidate takes its hardware queue now
  }
  hipStream_t chosen[kNeed] = {};
  int n = 0;
  if (dsl_option("stream_probe") != 0) {
    hipEvent_t ea = g_ev[dev][kEvRing - 1], eb = g_ev[dev][kEvRing - 2];
    for (int i = 0; i < kCand && n < kNeed; ++i) {
      bool ok = streams_concurrent(caller, cand[i], ea, eb);
      for (int j = 0; j < n && ok; ++j) ok = streams_concurrent(chosen[j], cand[i], ea, eb);
      if (ok) { chosen[n++] = cand[i]; used[i] = true; }
    }
    g_probe_result[dev] = n;
    if (n < kNeed)       // (say so - the layout then is whatever the runtime dealt, worth up to 20 % of a step)
      fprintf(stderr, "[libdsl_hip] stream probe: only %d of %d side streams found a hardware queue of their own (device %d); "
                      "dsl_streams_init reports the count\n", n, kNeed, dev);
  } else {
    g_probe_result[dev] = -1;
  }
  auto take = [&]() -> hipStream_t {            // any candidate not handed out yet
    for (int i = 0; i < kCand; ++i)
      if (!used[i]) { used[i] = true; return cand[i]; }
    return cand[0];
  };
  while (n < kNeed) chosen[n++] = take();        // (probe off, or fewer than three free queues: whatever the runtime dealt)
  g_side[dev][0] = chosen[0];                    // weight gradients
  g_side[dev][1] = chosen[1];                    // second tower ...
  g_side[dev][2] = chosen[1];                    // ... and second image chain: one stream
  g_side[dev][3] = chosen[2];                    // frozen prefix
  // Communication stream: on a hardware queue the library CHOOSES (option comm_queue) - a candidate that does NOT run concurrently
  // with that queue's owner.  With four queues taken by the caller and the three streams above, every further stream shares one of
  // them; which one the runtime's round-robin deals was left to chance until round 6
  hipStream_t comm = nullptr;
  g_comm_queue[dev] = 0;
  const int want = dsl_option("comm_queue");
  if (g_probe_result[dev] == kNeed && want >= 1 && want <= 4) {
    hipEvent_t ea = g_ev[dev][kEvRing - 1], eb = g_ev[dev][kEvRing - 2];
    hipStream_t owner = want == 4 ? caller : chosen[want - 1];
    for (int i = 0; i < kCand && !comm; ++i)
      if (!used[i] && !streams_concurrent(owner, cand[i], ea, eb)) { comm = cand[i]; used[i] = true; g_comm_queue[dev] = want; }
  }
  g_side[dev][4] = comm ? comm : take();         // communication
  g_side[dev][5] = take();                       // teacher sweep
  for (int i = 0; i < kCand; ++i)
    if (!used[i]) hipStreamDestroy(cand[i]);
  g_init[dev] = true;
}
hipEvent_t next_event(int dev) {
  g_evpos[dev] = (g_evpos[dev] + 1) % kEvRing;
  return g_ev[dev][g_evpos[dev]];
}
}  // namespace

void dsl_prof_phase(int cls, int end, long long flops_bits, long long bytes_bits, hipStream_t st);

extern "C" int dsl_run_ops(const dsl_op* ops, int n_ops, void* stream) {
  DSL_CHECK(ops || n_ops == 0, "dsl_run_ops: null op list");
  hipStream_t main_st = (hipStream_t)stream;
  int dev = 0;
  hipGetDevice(&dev);
  DSL_CHECK(dev >= 0 && dev < 16, "dsl_run_ops: device index %d out of range", dev);
  auto pick = [&](int id) -> hipStream_t { return id <= 0 ? main_st : g_side[dev][(id - 1) % kSide]; };
  for (int k = 0; k < n_ops; ++k) {
    const dsl_op& o = ops[k];
    int rc = 0;
    hipStream_t st = main_st;
    const bool dbg_pre = dsl_option("debug_sync") == 1;
    if (dbg_pre) {
      fprintf(stderr, "[dsl_run_ops] start op %d/%d kind %d stream %d desc %p p0 %p p1 %p\n", k, n_ops, o.kind, o.i[6], o.desc, o.p[0], o.p[1]);
      if (o.kind == DSL_OP_CONV && o.desc) {
        const dsl_conv_desc* d = (const dsl_conv_desc*)o.desc;
        fprintf(stderr, "    conv src %p wgt %p dst %p addend %p mask %p n %d nseg %d g %dx%d s %dx%d cs %d cd %d lds %d ldd %d lda %d k %d stride %d mode %d flags %d ws %p\n",
                d->src, d->wgt, d->dst, d->addend, d->mask, d->n, d->nseg, d->gh[0], d->gw[0], d->sh[0], d->sw[0], d->cs, d->cd, d->lds, d->ldd, d->lda, d->kh, d->stride, d->mode, d->flags, d->workspace);
      }
      fflush(stderr);
    }
    if (o.kind == DSL_OP_PROF) {
      dsl_prof_phase(o.i[0], o.i[1], o.l[0], o.l[1], pick(o.i[6]));
      continue;
    }
    // Step-level ablation (tools/step_ablation.sh; results are WRONG, only the clock is read): option skip_kinds = bit mask of op kinds
    // that are not launched (ordering ops are never skipped) - "what would the step gain if this component cost nothing?", measured
    // under the step's real contention instead of estimated from standalone kernel times.  Bit 30: only the ops of side stream 1.
    static const unsigned skip_kinds = (unsigned)dsl_option("skip_kinds");     // (read once: set it before the first dsl_run_ops)
    if (skip_kinds && o.kind < 30 && (skip_kinds >> o.kind & 1u) && o.kind != DSL_OP_FORK && o.kind != DSL_OP_JOIN &&
        o.kind != DSL_OP_RECORD && o.kind != DSL_OP_WAIT && (!(skip_kinds >> 30 & 1u) || o.i[6] == 1))
      continue;
    if (o.kind == DSL_OP_RECORD || o.kind == DSL_OP_WAIT) {
      side_init(dev, main_st);
      DSL_CHECK(o.i[1] >= 0 && o.i[1] < 16, "dsl_run_ops: event slot %d out of range", o.i[1]);
      if (o.kind == DSL_OP_RECORD) {
        hipEventRecord(g_named[dev][o.i[1]], pick(o.i[0]));
        g_named_set[dev][o.i[1]] = true;
      } else if (g_named_set[dev][o.i[1]]) {
        hipStreamWaitEvent(pick(o.i[0]), g_named[dev][o.i[1]], 0);
      }
      continue;
    }
    if (o.i[6] > 0 || o.kind == DSL_OP_FORK || o.kind == DSL_OP_JOIN) {
      side_init(dev, main_st);
      if (o.kind == DSL_OP_FORK || o.kind == DSL_OP_JOIN) {
        const int side_id = o.i[0] > 0 ? o.i[0] : 1, other = o.i[1];
        hipStream_t from = o.kind == DSL_OP_FORK ? pick(other) : pick(side_id);
        hipStream_t to = o.kind == DSL_OP_FORK ? pick(side_id) : pick(other);
        hipEvent_t e = next_event(dev);
        hipEventRecord(e, from);
        hipStreamWaitEvent(to, e, 0);
        continue;
      }
      st = pick(o.i[6]);
    }
    void* stream = (void*)st;      // shadows the argument: the op goes to the selected stream
    switch (o.kind) {
      case DSL_OP_CONV: rc = dsl_conv2d((const dsl_conv_desc*)o.desc, stream); break;
      case DSL_OP_WGRAD: rc = dsl_conv2d_wgrad((const dsl_wgrad_desc*)o.desc, stream); break;
      case DSL_OP_WGRAD_GROUP: rc = dsl_conv2d_wgrad_group((const dsl_wgrad_desc*)o.desc, o.i[0], stream); break;
      case DSL_OP_WGRAD_MULTI: rc = dsl_conv2d_wgrad_multi(o.p[0], o.p[1], stream); break;
      case DSL_OP_BNECK: rc = dsl_bottleneck_fwd((const dsl_bneck_desc*)o.desc, stream); break;
      case DSL_OP_GN_FWD: rc = dsl_groupnorm_relu_fwd((const dsl_gn_desc*)o.desc, stream); break;
      case DSL_OP_GN_BWD: rc = dsl_groupnorm_relu_bwd((const dsl_gn_desc*)o.desc, stream); break;
      case DSL_OP_MAXPOOL: rc = dsl_maxpool3x3s2_ld(o.p[0], o.p[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] > 0 ? o.i[4] : o.i[3], stream); break;
      case DSL_OP_RLA: rc = dsl_rla_op((const dsl_rla_desc*)o.desc, stream); break;
      case DSL_OP_SUM2X2: rc = dsl_sum2x2(o.p[0], o.p[1], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], o.i[5], stream); break;
      case DSL_OP_COLSUM: rc = dsl_colsum(o.p[0], (float*)o.p[1], (long)o.l[0], o.i[0], o.i[1], stream); break;
      case DSL_OP_MEMSET:
        if (hipMemsetAsync(o.p[0], o.i[0], (size_t)o.l[0], st) != hipSuccess) {
          dsl_set_error("dsl_run_ops: memset failed at op %d", k);
          rc = -2;
        }
        break;
      case DSL_OP_PACK_DGRAD: rc = dsl_pack_dgrad_batched((const dsl_pack_item*)o.p[0], o.i[0], o.i[1], stream); break;
      case DSL_OP_PACK_IMAGE: rc = dsl_pack_image((const float*)o.p[0], o.p[1], o.i[0], o.i[1], o.i[2], stream); break;
      case DSL_OP_STEM_POOL:
        rc = dsl_stem_pool_half((const float*)o.p[0], o.p[1], (const float*)(intptr_t)o.l[0], (const float*)(intptr_t)o.l[1], o.p[2], o.i[0],
                                o.i[1], o.i[2], o.i[3], o.i[4], stream);
        break;
      case DSL_OP_ASSIGN: rc = dsl_fcos_assign((const dsl_fcos_desc*)o.desc, stream); break;
      case DSL_OP_LOSS: rc = dsl_fcos_loss((const dsl_fcos_desc*)o.desc, stream); break;
      case DSL_OP_QUANT_FP8: {
        float sc;
        const int32_t bits = (int32_t)o.l[1];
        memcpy(&sc, &bits, 4);
        if (o.p[2]) {
          rc = dsl_absmax(o.p[0], (long)o.l[0], o.i[0], o.i[1], (float*)o.p[2], o.i[2], stream);
          if (rc == 0) rc = dsl_quant_fp8_dyn(o.p[0], o.p[1], (long)o.l[0], o.i[0], o.i[1], (const float*)o.p[2], o.i[2], stream);
        } else {
          rc = dsl_quant_fp8(o.p[0], o.p[1], (long)o.l[0], o.i[0], o.i[1], sc, stream);
        }
        break;
      }
      case DSL_OP_FP8_PREP: {
        float mg;
        const int32_t bits = (int32_t)o.l[1];
        memcpy(&mg, &bits, 4);
        rc = dsl_fp8_prep((const dsl_fp8_prep_item*)o.p[0], o.i[0], o.i[1], o.i[2], mg, stream);
        break;
      }
      case DSL_OP_QUANT_FP8_DELAYED:
        rc = dsl_quant_fp8_delayed(o.p[0], o.p[1], (long)o.l[0], o.i[0], o.i[1], (const float*)o.p[3], (float*)o.p[2], o.i[2], stream);
        break;
      case DSL_OP_FP8_COMB: rc = dsl_fp8_comb((const float*)o.p[0], (float*)o.p[1], o.i[0], (const float*)o.p[2], o.i[2], stream); break;
      case DSL_OP_QUANT_FP8_W: {
        float sc;
        const int32_t bits = (int32_t)o.l[1];
        memcpy(&sc, &bits, 4);
        rc = dsl_quant_fp8_weights((const float*)o.p[0], o.p[1], (float*)o.p[2], (const float*)o.p[3], o.i[0], o.i[1], o.i[2], sc, stream);
        break;
      }
      default:
        dsl_set_error("dsl_run_ops: unknown op kind %d at index %d", o.kind, k);
        return -1;
    }
    if (rc != 0) return rc;
    // option debug_sync = 1: drain the device after every op and name it - a faulting launch is then the last line on stderr
    if (dbg_pre) {
      const hipError_t e = hipDeviceSynchronize();
      fprintf(stderr, "[dsl_run_ops] op %d/%d kind %d stream %d -> %s\n", k, n_ops, o.kind, o.i[6], hipGetErrorString(e));
      fflush(stderr);
    }
  }
  return 0;
}

// Make `stream` wait for named event slot `slot` (recorded by a DSL_OP_RECORD of an earlier dsl_run_ops call): the
// communication stream of the data-parallel wrapper waits for "the weight gradients of backward segment s are done"
// without involving the caller's compute stream.  No-op (returns 1) when the slot was never recorded.
extern "C" int dsl_stream_wait_slot(int slot, void* stream) {
  int dev = 0;
  hipGetDevice(&dev);
  DSL_CHECK(dev >= 0 && dev < 16 && slot >= 0 && slot < 16, "dsl_stream_wait_slot: bad device %d / slot %d", dev, slot);
  if (!g_init[dev] || !g_named_set[dev][slot]) return 1;
  DSL_CHECK(hipStreamWaitEvent((hipStream_t)stream, g_named[dev][slot], 0) == hipSuccess, "dsl_stream_wait_slot: hipStreamWaitEvent failed");
  return 0;
}

// Record named event slot `slot` on `stream` (what a DSL_OP_RECORD inside an op list does, for a stream the library does not
// own): the optimizer marks "the head + FPN bucket is updated" on its own stream, and the next forward list waits for exactly
// that slot in front of the FPN (deferred head update).
extern "C" int dsl_stream_record_slot(int slot, void* stream) {
  int dev = 0;
  hipGetDevice(&dev);
  DSL_CHECK(dev >= 0 && dev < 16 && slot >= 0 && slot < 16, "dsl_stream_record_slot: bad device %d / slot %d", dev, slot);
  side_init(dev, (hipStream_t)stream);
  DSL_CHECK(hipEventRecord(g_named[dev][slot], (hipStream_t)stream) == hipSuccess, "dsl_stream_record_slot: hipEventRecord failed");
  g_named_set[dev][slot] = true;
  return 0;
}

// The library's stream `id` of the current device (1..3 side streams of dsl_run_ops, 4 prefix, 5 communication, 6 sweep: side_init),
// for callers that have to queue their own work in order with it or need a stream whose hardware queue is part of the library's layout:
// the optimizer puts the deferred bucket's update on side stream 1, right behind the weight gradients it waits for (a stream of
// its own may share a hardware queue with another of the step's streams and then runs behind THAT stream's backlog).
extern "C" int dsl_side_stream(int id, void** stream_out) {
  int dev = 0;
  hipGetDevice(&dev);
  DSL_CHECK(dev >= 0 && dev < 16 && id >= 1 && id <= kStreams && stream_out, "dsl_side_stream: bad device %d / stream id %d", dev, id);
  side_init(dev, (hipStream_t)0);          // (not initialised yet and no caller stream at hand: probe against the default stream)
  *stream_out = (void*)g_side[dev][id - 1];
  return 0;
}

// Create (and pick, see side_init) the library's streams of the current device now, measured against `caller_stream` - the stream
// the step will be queued on.  Optional: the first dsl_run_ops does it otherwise.  *distinct_out (may be NULL): how many of the three
// concurrently used streams the probe found on hardware queues of their own (3 = all; -1 = probe switched off).
extern "C" int dsl_streams_init(void* caller_stream, int* distinct_out) {
  int dev = 0;
  hipGetDevice(&dev);
  DSL_CHECK(dev >= 0 && dev < 16, "dsl_streams_init: device index %d out of range", dev);
  side_init(dev, (hipStream_t)caller_stream);
  if (distinct_out) *distinct_out = g_probe_result[dev];
  return 0;
}

// Where the communication stream sits: 1 = the weight-gradient stream's hardware queue, 2 = the second chain's, 3 = the frozen prefix's,
// 4 = the caller's, 0 = not placed (probe off / no candidate found: as the runtime dealt it).
extern "C" int dsl_comm_stream_queue(void) {
  int dev = 0;
  hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || !g_init[dev]) return -1;
  return g_comm_queue[dev];
}

// ---- communication proxy (bench.py extra.comm_proxy, tests/test_comm_proxy_gpu.py) -------------------------------------------------
// What a ring all-reduce of `n` floats costs the DEVICE beside the backward pass, without a second GPU: `wgs` workgroups (RCCL: one
// per channel) make `passes` read-modify-write passes over the bucket (x *= 1.0f: the gradients keep their bits), i.e. the HBM
// traffic of reduce-scatter + all-gather and the CUs RCCL's kernels hold while they run.  No xGMI traffic, no peer latency.
namespace {
__global__ __launch_bounds__(512) void comm_proxy_kernel(float* __restrict__ g, long long n4, int passes, float one) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4* p = reinterpret_cast<f4*>(g);
  for (int r = 0; r < passes; ++r) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
      f4 v = __builtin_nontemporal_load(p + i);
      v *= one;          // (a run-time 1.0f: not folded away, and every bit pattern - signed zeros too - survives)
      __builtin_nontemporal_store(v, p + i);
    }
    __threadfence();
  }
}
}  // namespace
extern "C" int dsl_comm_proxy(float* buf, long long n, int wgs, int passes, void* stream) {
  DSL_CHECK(buf != nullptr && n >= 0 && ((uintptr_t)buf & 15) == 0 && wgs >= 1 && wgs <= 1024 && passes >= 1,
            "dsl_comm_proxy: bad arguments (n %lld wgs %d passes %d)", n, wgs, passes);
  if (n < 4) return 0;
  hipLaunchKernelGGL(comm_proxy_kernel, dim3(wgs), dim3(512), 0, (hipStream_t)stream, buf, n / 4, passes, 1.0f);
  DSL_LAUNCH_CHECK("comm_proxy_kernel");
  return 0;
}

// ---- live kernel timing ----------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
constexpr int kMaxRec = 1 << 15;
int g_prof_on = 0;       // 0 off, 1 = only class 0 (the dominant kernel), 2 = every class
ProfRec* g_rec = nullptr;
int g_nrec = 0, g_nevents = 0;
}  // namespace

bool dsl_prof_active() { return g_prof_on != 0; }

int dsl_prof_begin(int cls, double flops, hipStream_t st, double bytes) {
  if (!g_prof_on || g_nrec >= kMaxRec) return -1;
  if (g_prof_on == 1 && cls != 0) return -1;
  if ((g_prof_on == 3) != (cls >= 4)) return -1;          // phases only in mode 3, kernels only in modes 1, 2
  if (!g_rec) g_rec = (ProfRec*)calloc(kMaxRec, sizeof(ProfRec));
  if (g_nrec >= g_nevents) {
    hipEventCreate(&g_rec[g_nrec].a);
    hipEventCreate(&g_rec[g_nrec].b);
    g_nevents = g_nrec + 1;
  }
  g_rec[g_nrec].cls = cls;
  g_rec[g_nrec].flops = flops;
  g_rec[g_nrec].bytes = bytes;
  hipEventRecord(g_rec[g_nrec].a, st);
  return g_nrec++;
}

void dsl_prof_end(int id, hipStream_t st) {
  if (id >= 0) hipEventRecord(g_rec[id].b, st);
}

void dsl_prof_phase(int cls, int end, long long flops_bits, long long bytes_bits, hipStream_t st) {
  static int open_id[DSL_PROF_CLASSES];
  if (g_prof_on != 3 || cls < 4 || cls >= DSL_PROF_CLASSES) return;
  if (!end) {
    double fl, by;
    memcpy(&fl, &flops_bits, 8);
    memcpy(&by, &bytes_bits, 8);
    open_id[cls] = dsl_prof_begin(cls, fl, st, by);
  } else {
    dsl_prof_end(open_id[cls], st);
    open_id[cls] = -1;
  }
}

extern "C" int dsl_prof_enable(int on) {
  g_prof_on = on;
  return 0;
}
extern "C" int dsl_prof_reset(void) {
  g_nrec = 0;
  return 0;
}
extern "C" int dsl_prof_read2(int64_t* launches, double* ms, double* flops, double* bytes) {
  for (int c = 0; c < DSL_PROF_CLASSES; ++c) { launches[c] = 0; ms[c] = 0; flops[c] = 0; if (bytes) bytes[c] = 0; }
  for (int i = 0; i < g_nrec; ++i) {
    hipEventSynchronize(g_rec[i].b);
    float t = 0.f;
    hipEventElapsedTime(&t, g_rec[i].a, g_rec[i].b);
    const int c = g_rec[i].cls;
    if (c >= 0 && c < DSL_PROF_CLASSES) {
      launches[c]++; ms[c] += t; flops[c] += g_rec[i].flops;
      if (bytes) bytes[c] += g_rec[i].bytes;
    }
  }
  return 0;
}
extern "C" int dsl_prof_read(int64_t* launches, double* ms, double* flops) { return dsl_prof_read2(launches, ms, flops, nullptr); }

// ---- probe: semantics of ds_read_b64_tr_b16 (used by tests/test_probe_gpu.py) -------------------
__global__ void probe_tr16_kernel(const uint16_t* img, const int* lane_off, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = img[i];
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(lds + lane_off[threadIdx.x]));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = (uint16_t)v[e];
}

extern "C" int dsl_probe_tr16(const uint16_t* img, const int32_t* lane_off, uint16_t* out, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, img, lane_off, out);
  DSL_LAUNCH_CHECK("probe_tr16_kernel");
  return 0;
}

// ---- probe: XCC id of each workgroup + XCD-local (workgroup-scope, L2-resident) float atomics --------------
// The XCD-aware workgroup mappings of the conv / wgrad kernels assume blocks with equal b % 8 share an XCD (round-robin dispatch); this probe pins
// that (HW_REG_XCC_ID) and that workgroup-scope global_atomic_add_f32 into per-XCD buffers is complete after the
// kernel.
__global__ void probe_xcc_kernel(int* xcc_of_block, float* acc /* [8][256] */) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = (int)x;
  __hip_atomic_fetch_add(acc + (x & 7) * 256 + threadIdx.x, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- probe: which CUs a CU-masked stream runs on -------------------------------------------------------------------
// every workgroup claims a whole CU's LDS for ~20 us and records (XCC id, HW_ID: SE/SH/CU fields)
__global__ void probe_cu_kernel(int* out) {
  extern __shared__ int lds[];
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 32)" : "=s"(h));
  lds[threadIdx.x] = (int)h;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(20);      // 100 MHz constant clock: 20 us
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (int)x;
    out[2 * blockIdx.x + 1] = lds[0];
  }
}

extern "C" int dsl_probe_cu_mask(const uint32_t* mask, int nwords, int32_t* out, int nblocks) {
  hipStream_t st;
  if (mask && nwords > 0) {
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask) != hipSuccess) {
      dsl_set_error("dsl_probe_cu_mask: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(hipGetLastError()));
      return -2;
    }
  } else {
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  }
  const int lds = 160 * 1024;
  hipFuncSetAttribute((const void*)probe_cu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(probe_cu_kernel, dim3(nblocks), dim3(64), lds, st, out);
  hipError_t e = hipStreamSynchronize(st);
  hipStreamDestroy(st);
  DSL_CHECK(e == hipSuccess, "dsl_probe_cu_mask: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int dsl_probe_xcc(int32_t* xcc_of_block, float* acc, int nblocks, void* stream) {
  hipLaunchKernelGGL(probe_xcc_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, xcc_of_block, acc);
  DSL_LAUNCH_CHECK("probe_xcc_kernel");
  return 0;
}
