// common.hpp -- context, error reporting and workspace management shared by the
// translation units of libkimchi_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <initializer_list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kimchi_hip.h"

namespace kh {

void set_error(const char* fmt, ...);

#define KH_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            kh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return KH_E_DEVICE;                                                              \
        }                                                                                    \
    } while (0)

#define KH_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            kh::set_error(__VA_ARGS__);       \
            return KH_E_INVALID;              \
        }                                     \
    } while (0)

// A grow-only device buffer (workspace), owning its allocation (move-only; freed with its owner).  Not thread-safe
// by itself: the owner (Context / SRS handle) serialises users with its mutex.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~DevBuf() { release(); }
    // bumped by every (re)allocation / release of any workspace buffer: part of the hipGraph key, so that a captured launch sequence can
    // never be replayed across a reallocation -- not even one that hands the same address back (defence in depth: every pointer the
    // launches bake in is in the key as well)
    static std::atomic<uint64_t>& generation() { static std::atomic<uint64_t> g{1}; return g; }
    bool graph_keyed = true;                  // false: a buffer no captured launch sequence ever reads (the host-pointer transforms' transfer buffers)
    int reserve(size_t bytes) {
        if (bytes <= cap) return KH_OK;
        if (graph_keyed) generation()++;
        if (p) { hipError_t e = hipFree(p); (void)e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); p = nullptr; return KH_E_NOMEM; }
        cap = want;
        return KH_OK;
    }
    void release() { if (p) { hipError_t e = hipFree(p); (void)e; if (graph_keyed) generation()++; } p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct PhaseTimer {
    static constexpr int MAXP = 24;
    hipEvent_t k0 = nullptr, k1 = nullptr;    // start / stop of ONE kernel of interest (hipExtLaunchKernelGGL): its own duration,
    const char* kname = nullptr;              // without the barrier / dispatch gaps an event pair around the launch includes
    hipEvent_t ev[MAXP + 1];
    const char* names[MAXP];
    int n = 0;
    bool created = false;
    bool enabled = false;                     // kh_set_phase_timers: an event between two kernels costs the stream 6-10 us of idle time
    int init() {
        if (created) return KH_OK;
        for (int i = 0; i <= MAXP; i++) KH_HIP(hipEventCreate(&ev[i]));
        KH_HIP(hipEventCreate(&k0)); KH_HIP(hipEventCreate(&k1));
        created = true;
        return KH_OK;
    }
    void begin(hipStream_t s) { n = 0; kname = nullptr; if (enabled && created) (void)hipEventRecord(ev[0], s); }
    void mark(const char* name, hipStream_t s) {
        if (!enabled || !created || n >= MAXP) return;
        names[n] = name; n++;
        (void)hipEventRecord(ev[n], s);
    }
};

// One MSM pipeline slot: its own stream, workspace and result staging, so that several MSMs can be in
// flight (kh_msm_submit / kh_msm_wait).  k_accumulate is held to 112 VGPRs so that the sort kernels
// (4-16 VGPRs) of the NEXT job fit beside its 4 waves per SIMD: with three slots the sort of job i+2
// and the latency-bound tail of job i both run underneath the accumulation of job i+1.
struct MsmSlot {
    hipStream_t stream = nullptr;
    PhaseTimer timer;
    DevBuf ws_scalars, ws_digits, ws_hist, ws_cnt, ws_off, ws_ntask, ws_toff, ws_entries, ws_partial,
        ws_buckets, ws_seg, ws_out, ws_scan_tmp, ws_biglist, ws_points, ws_order, ws_chunks, ws_handed, ws_sync, ws_mid,
        ws_b29, ws_a1, ws_a2, ws_xlist;                              // wide windows (c = 20): lazy buckets, first-level chunk sums, the 2 x 2^lo marginals
    void* pinned = nullptr; size_t pinned_cap = 0;      // host staging of the group sums (XYZZ)
    hipEvent_t done = nullptr;
    // pending job (set by enqueue, consumed by finish)
    bool busy = false;
    std::thread::id owner;             // the host thread that queued the pending job (acquire_slot: another thread's ticket will be waited for)
    uint64_t ticket = 0;
    int curve = 0, W = 0, c = 0, precomp = 0, planes = 0, plane_shift[2] = {0, 0};
    int wide_lo = 0, g_wide_lo = 0;    // wide windows: bits of the low bucket digit (0: the narrow path); msm_finish folds two marginal groups per MSM
    size_t k = 0, ngroups = 0;
    // hipGraph of one MSM's launch sequence (the opening rounds repeat the same MSM -- same basis, scalar buffer, sizes --
    // 16 times: ~22 launches per round replayed as one graph).  Keyed by every pointer and size the launches bake in.
    hipGraphExec_t gexec = nullptr;
    uint64_t gkey = 0, gseen = 0;
    const void* gscalars = nullptr;   // the scalar buffer the captured launch sequence reads (an opening's: its SRS handle's ipa_sc)
    size_t gnout = 0;
    int g_W = 0, g_c = 0, g_precomp = 0, g_planes = 0, g_shift[2] = {0, 0};
    size_t g_ngroups = 0;
    // the one-launch sort (k_sort_fused) spins on grid barriers and needs all its blocks resident at once; if another PROCESS shares the
    // GPU they may never be -- its barriers then give up after a bounded spin, set this host-visible word, and msm_finish re-runs the job
    // with the multi-launch sort (the arguments of the pending job are kept for that) and disables the fused path for the process
    volatile uint32_t* host_abort = nullptr;          // pinned host memory, written by the kernel
    // MSM_SPREAD_SCALARS was a wrong promise: k_bucket_sum_q met a bucket with more task partials than a quad may sum in sequence, stored this word
    // (pinned) and left the bucket empty; msm_finish re-runs the job with the hot-bucket kernels and suspends the hint (Context::spread_suspended)
    volatile uint32_t* spread_abort = nullptr;
    bool spread_used = false, g_spread = false;
    bool flag_unavailable = false;                     // no coherent host allocation for done_flag / pinned: completion by event
    bool pinned_coherent = false;
    // Completion by flag (round 5): when a job's LAST kernel is k_marginal_fin_q (it writes the result into `pinned` itself), the last block of that
    // kernel to finish also stores the slot's launch count into done_flag (pinned).  A synchronous waiter polls that word instead of hipEventQuery:
    // the host has the result ~5 us earlier (tools/latency/launch_latency.hip: 0.2 against 5.3 us after the kernel's last store).
    volatile uint32_t* done_flag = nullptr;           // pinned host memory
    DevBuf ws_done;                                    // [0] blocks finished in the running launch, [1] launches finished
    uint32_t done_expect = 0;                          // launches enqueued with the flag so far (what done_flag shows when the newest one has ended)
    bool done_by_flag = false, g_done_by_flag = false; // the job in flight (the captured graph) ends with the flag store
    bool fused_used = false, g_fused = false;
    struct { const void* pts; const uint8_t* inf; size_t bn, stride, batch_stride; int precomp_c; size_t offset; const uint64_t* scalars; size_t n, k; int mont, curve; bool glv; } retry{};
};
static constexpr int MSM_SLOTS = 4;

struct Context {
    std::mutex mu;            // serialises the host side of the C ABI (GPU waits happen outside it)
    std::condition_variable cv;        // a synchronous caller that finds every slot busy waits here for one to finish
    int sync_inflight = 0;             // slots some thread is currently blocked on (they WILL free up)
    std::multiset<std::thread::id> blocked_owners;   // threads waiting in acquire_slot (deadlock test: see there)
    int device = -1;
    bool ready = false;
    hipStream_t stream = nullptr;      // = slot[0].stream: the library's main stream (NTT / LDE / vector steps / opening folds)
    bool main_dirty = false;           // asynchronous work was queued on the main stream since the last kh_sync
    hipEvent_t order_ev = nullptr;     // orders device-resident producers on the main stream before an MSM on another slot's stream
    int num_cus = 256;
    size_t lds_per_cu = (size_t)160 << 10, lds_per_block = (size_t)64 << 10;    // hipDeviceProp: what holds k_acc_wide29 to a block count per CU
    bool spread_suspended = false;     // an MSM under MSM_SPREAD_SCALARS met a hot bucket: the hint is ignored until the caller's next opening (kh_ipa_begin)
    bool fused_disabled = false;       // a k_sort_fused launch could not get all its blocks resident (shared GPU): multi-launch sort from then on
    PhaseTimer timer;                  // NTT / LDE phases (MSM phases are per slot)
    // last timings (filled after a sync)
    std::vector<std::pair<const char*, float>> last;      // names are string literals (kh_last_timings hands them out)
    MsmSlot slot[MSM_SLOTS];
    uint64_t next_ticket = 1;
    std::chrono::steady_clock::time_point last_sync_msm_arrival{};     // burst detection of the caller coalescing (api.hip: msm_common)
    // NTT workspace
    DevBuf ws_ntt_a, ws_ntt_b;
    void* pinned = nullptr; size_t pinned_cap = 0;
    // Pinned staging ring for the small argument tables of the vector steps (token programs, pointer tables, constants): a copy
    // from pageable memory drains the stream first, one from pinned memory is just another stream operation -- so kh_expr_* /
    // kh_poly_* can be queued back to back like kh_ntt_dev.  A slice is reused only after a full turn of the ring, which waits
    // for the main stream.  Guarded by `mu`.
    char* stage = nullptr; size_t stage_cap = 0, stage_cur = 0;
    // copies the parts back to back into the ring and queues ONE host-to-device copy of them to dst_dev on the main stream
    int stage_upload(void* dst_dev, std::initializer_list<std::pair<const void*, size_t>> parts);
    // asynchronous work was queued on the main stream: a later MSM on another slot's stream waits for this point
    void mark_async() { if (hipEventRecord(order_ev, stream) == hipSuccess) main_dirty = true; }
    // named scratch buffers / one-time flags of the other translation units (what used to be function-local statics:
    // a static is process-wide, these belong to ONE device).  References stay valid (std::map nodes do not move).
    std::mutex scratch_mu;
    std::map<std::string, DevBuf> scratch_bufs;
    std::set<std::string> done_flags;
    DevBuf& scratch(const char* name) { std::lock_guard<std::mutex> lk(scratch_mu); return scratch_bufs[name]; }
    bool once(const char* name) { std::lock_guard<std::mutex> lk(scratch_mu); return done_flags.insert(name).second; }   // true the first time
    void trim_scratch() { std::lock_guard<std::mutex> lk(scratch_mu); for (auto& kv : scratch_bufs) kv.second.release(); }
};
static constexpr int KH_MAX_DEVICES = 16;

// The context of the calling thread's CURRENT device: the one an enclosing DeviceScope selected (entry points that
// take an SRS / opening handle run on the handle's device), else the thread's kh_set_device / kh_init choice, else
// the process default (the first device initialised).
Context& ctx();
// initialises the current device's context on first use and binds the calling thread to it (hipSetDevice is
// per-thread state: every entry point must do this, not just the thread that ran kh_init)
int ensure_init();
struct DeviceScope {               // run the rest of this scope on `device` (no-op for device < 0)
    int prev;
    explicit DeviceScope(int device);
    ~DeviceScope();
};
void collect_timings(Context& c, PhaseTimer& t);
// process-wide event counters behind kh_counter (tests and tools read them: how often a rare path ran).  `name` must be one of COUNTER_NAMES.
enum CounterId { CNT_SPREAD_RETRY = 0, CNT_FUSED_RETRY, CNT_GRAPH_REPLAY, CNT_GRAPH_CAPTURE, CNT_REBASE_LAUNCH, CNT_REBASE_SWITCH, CNT_REBASE_ABANDON, CNT_REBASED_ROUNDS, CNT_COUNT };
static constexpr const char* COUNTER_NAMES[CNT_COUNT] = {"spread_retry", "fused_retry", "graph_replay", "graph_capture", "rebase_launch", "rebase_switch", "rebase_abandon", "rebased_rounds"};
std::atomic<uint64_t>& counter(CounterId id);

// device exclusive scan of n u32 values (in may alias out); tmp is workspace
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, DevBuf& tmp, hipStream_t s);

}  // namespace kh
