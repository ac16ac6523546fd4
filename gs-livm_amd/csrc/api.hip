// api.hip -- the C ABI of libgsraster_hip.so (include/gsraster.h): argument checks, blob carving and
// stage sequencing.  Restates the ORCHESTRATION of CudaRasterizer::Rasterizer::forward / backward
// (reference rasterizer_impl.cu:181-342, :346-457); all device work is in the sibling .hip files.
//
// How a forward is sequenced (gsr_forward):
//   synchronous  (a thread's first, debug, GSR_SYNC_FORWARD)  k_preprocess + depth sort, host waits for num_rendered
//                (mailbox), exact binning blob, one binning chain (enqueue_chain) + blend;
//   speculative  blob sized from the thread's recent frames, everything enqueued at once, counts read on the device,
//                the mailbox read after the last enqueue; overflow -> the chain is enqueued again with the exact size;
//   near / far   (dense frames) two chains over the Gaussians in depth order -- near up to a budget, then only the far
//                Gaussians whose rectangle still holds an unfinished tile;
//   + far-chain speculation (after two split frames that left nothing unfinished): the far chain is not enqueued on
//                the caller's stream -- ASYNCHRONOUS: gated, on the library's own stream behind a stream-side wait for
//                the near blend's decision, the caller's stream waiting for the frame's go word; or HOST-DECIDED: the
//                count of unfinished quads comes back through the mailbox -- and only the near candidates are sorted
//                by depth up front (partial depth sort), the full sort moving into the far chain.
// Per host thread and device: ThreadCtx (mailbox, counters, histogram pair, predictions, the second stream).
#include "../../include/gsraster.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <unordered_map>
#include <vector>

#include "gsr_internal.hpp"

using namespace gsr;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail(GSR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));       \
  } while (0)

// debug != 0: synchronise after the stage so a faulting kernel is reported where it ran
// (the reference's CHECK_CUDA, auxiliary.h:146-154).
#define STAGE(expr)                                                                               \
  do {                                                                                            \
    HIP_TRY(expr);                                                                                \
    if (debug) HIP_TRY(hipStreamSynchronize(stream));                                             \
  } while (0)

// ---- optional event profiler -------------------------------------------------------------------
namespace gsr {
std::atomic<bool> g_prof_on{false};
std::atomic<unsigned long long> g_prof_mask{~0ull};
thread_local bool t_prof_suppress = false;
static std::mutex g_ev_mu;            // guards the pool: the reference calls the rasterizer from several threads
static std::vector<hipEvent_t> g_ev;  // pool: [2*i] start, [2*i+1] stop
static std::vector<int> g_ev_id;
static size_t g_ev_used = 0;
void prof_begin(int id, hipStream_t s) {
  g_ev_mu.lock();  // released by prof_end: a start / stop pair is recorded as a unit
  if (g_ev_used == g_ev_id.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
      g_prof_on = false;
      g_ev_id.push_back(-1);  // keeps prof_end's bookkeeping consistent; never read (id -1)
      g_ev.push_back(nullptr); g_ev.push_back(nullptr);
      return;
    }
    g_ev.push_back(a); g_ev.push_back(b); g_ev_id.push_back(id);
  }
  g_ev_id[g_ev_used] = id;
  if (g_ev[2 * g_ev_used]) (void)hipEventRecord(g_ev[2 * g_ev_used], s);
}
void prof_end(hipStream_t s) {
  if (g_ev_used < g_ev_id.size() && g_ev[2 * g_ev_used + 1]) {
    (void)hipEventRecord(g_ev[2 * g_ev_used + 1], s);
    g_ev_used++;
  }
  g_ev_mu.unlock();
}
}  // namespace gsr
static const char* const kKernelNames[K_COUNT] = {
    "k_preprocess", "k_point_offsets", "k_scan_offsets", "k_emit",
    "k_sort_hist", "k_sort_scan_chunks", "k_sort_scan_top", "k_sort_scatter", "k_tile_ranges", "k_blend_forward",
    "k_blend_backward", "k_compact_touched", "k_gather_records", "k_gaussian_backward", "k_mark_visible", "k_sort_hist[depth]",
    "k_sort_scan_chunks[depth]", "k_sort_scan_top[depth]", "k_sort_scatter[depth]", "k_activate",
    "k_activate_backward", "k_adam", "k_loss_forward", "k_loss_finalize", "k_loss_backward", "k_init_gaussians", "k_pack_ply_rows", "k_model_step", "k_tile_order", "k_live_sat", "k_compact_near"};

extern "C" {

int gsr_kernel_count(void) { return K_COUNT; }
const char* gsr_kernel_name(int id) { return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : ""; }
int gsr_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_ev_mu);
  if (on) g_ev_used = 0;  // a fresh recording; disabling keeps what was recorded for gsr_profile_read
  g_prof_on = on != 0;
  g_prof_mask = ~0ull;
  return GSR_OK;
}
int gsr_profile_enable_only(const int* kernel_ids, int n) {
  if (!kernel_ids && n > 0) return fail(GSR_ERR_INVALID_ARGUMENT, "null kernel id list");
  unsigned long long m = 0;
  for (int i = 0; i < n; i++) {
    if (kernel_ids[i] < 0 || kernel_ids[i] >= K_COUNT) return fail(GSR_ERR_INVALID_ARGUMENT, "bad kernel id");
    m |= 1ull << kernel_ids[i];
  }
  std::lock_guard<std::mutex> lk(g_ev_mu);
  g_ev_used = 0;
  g_prof_mask = m;
  g_prof_on = true;
  return GSR_OK;
}
int gsr_profile_read(int max_ids, double* total_ms, int* launches) {
  const int n = max_ids < K_COUNT ? max_ids : (int)K_COUNT;
  std::lock_guard<std::mutex> lk(g_ev_mu);
  for (int i = 0; i < n; i++) { total_ms[i] = 0.0; launches[i] = 0; }
  for (size_t i = 0; i < g_ev_used; i++) {
    HIP_TRY(hipEventSynchronize(g_ev[2 * i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]));
    const int id = g_ev_id[i];
    if (id >= 0 && id < n) { total_ms[id] += ms; launches[id]++; }
  }
  g_ev_used = 0;
  return n;
}

const char* gsr_last_error(void) { return g_err; }
int gsr_abi_version(void) { return GSR_ABI_VERSION; }

uint32_t gsr_higher_msb(uint32_t n) {
  // smallest b such that n < 2^b, found by bisection like the reference (rasterizer_impl.cu:35-48)
  uint32_t msb = 16, step = 16;
  while (step > 1) {
    step >>= 1;
    msb = (n >> msb) ? msb + step : msb - step;
  }
  if (n >> msb) msb++;
  return msb;
}

// Blob sizes are rounded up to one eighth of their power of two (at most 12 % more, 1 MB at least): a map that grows by
// a per cent every few iterations (gaussian.cu:241-313) then asks its allocator for the SAME size most of the time, and
// a caching allocator (Torch's) can hand the previous blob back instead of going to the driver for a slightly larger
// one -- every such trip is milliseconds, and the blocks it leaves behind fragment the pool.
static size_t round_blob(size_t bytes) {
  if (bytes < ((size_t)1 << 20)) return bytes;
  size_t p2 = (size_t)1 << 20;
  while ((p2 << 1) <= bytes) p2 <<= 1;
  const size_t g = p2 >> 3 > ((size_t)1 << 20) ? p2 >> 3 : ((size_t)1 << 20);
  return (bytes + g - 1) / g * g;
}
// (the same for a binning capacity predicted from history, in instances; capacities a caller or a test hook states
// exactly are left alone)
static uint32_t round_capacity(uint32_t n) {
  if (n < (1u << 20)) return n;
  uint32_t p2 = 1u << 20;
  while (p2 <= (n >> 1)) p2 <<= 1;
  const uint32_t g = p2 >> 4;
  const unsigned long long r = ((unsigned long long)n + g - 1) / g * g;
  return r > 0x7fffffffull ? 0x7fffffffu : (uint32_t)r;
}

size_t gsr_geometry_bytes(int P) {
  size_t b = 0;
  GeomState::carve(nullptr, (size_t)(P > 0 ? P : 0), &b);
  return round_blob(b);
}
size_t gsr_image_bytes(int width, int height) {
  size_t b = 0;
  ImageState::carve(nullptr, width, height, &b);
  return b;
}
size_t gsr_binning_bytes(int R) {
  size_t b = 0;
  BinningState::carve(nullptr, (size_t)(R > 0 ? R : 0), &b);
  return b;
}

// Binning mode (gsraster.h, gsr_set_reference_rects): process-wide, initialised once from GSR_REFERENCE_RECTS.
static std::atomic<int>& reference_rects_flag() {
  static std::atomic<int> flag([] {
    const char* e = getenv("GSR_REFERENCE_RECTS");
    return (e && e[0] && e[0] != '0') ? 1 : 0;
  }());
  return flag;
}

// Per host thread (gsr_set_reference_rects_thread / gsr_set_near_far_thread): -1 = follow the process-wide value.  The
// reference calls the rasterizer from several threads; a thread that needs a mode of its own for one
// call sets it here and never disturbs the forwards of the others.
static thread_local int t_reference_rects = -1;
static thread_local int t_near_far = -1;

static FrameParams make_params(int P, int D, int M, int W, int H, float tan_fovx, float tan_fovy, float scale_mod) {
  FrameParams fp;
  fp.P = P; fp.D = D; fp.M = M; fp.W = W; fp.H = H;
  fp.gx = (W + TILE - 1) / TILE;
  fp.gy = (H + TILE - 1) / TILE;
  fp.tan_fovx = tan_fovx;
  fp.tan_fovy = tan_fovy;
  fp.focal_y = H / (2.0f * tan_fovy);  // rasterizer_impl.cu:210-211
  fp.focal_x = W / (2.0f * tan_fovx);
  fp.scale_modifier = scale_mod;
  fp.ref_rects = t_reference_rects >= 0 ? t_reference_rects : reference_rects_flag().load(std::memory_order_relaxed);
  return fp;
}

// ---- shared host state ---------------------------------------------------------------------------------------
static std::atomic<unsigned long long> g_mailbox_slow_hits{0};  // forwards whose count arrived through the stream query
static std::atomic<unsigned long long> g_speculative_forwards{0}, g_speculation_overflows{0};

// What the slow path saw the last time it fired (gsr_mailbox_slow_path_last): enough to tell a late-visible store
// (the word appears some time after the stream has drained) from a late dispatch (the word is there as soon as a HIP
// call has been made) the next time it occurs naturally.
static std::mutex g_slow_mu;
static gsr_mailbox_event g_slow_last = {0, 0, 0, 0.0, 0, 0, 0, 0, 0.0, 0.0};

// Per host thread and device: the mailbox, the kernel's "workgroups done" word, the two digit-histogram buffers and the
// capacity prediction.  A host thread is inside gsr_forward for one forward at a time, and its forwards must be
// ordered on the device (ONE stream per host thread at a time, as the reference's default-stream use): the done
// word and the histogram pair are re-used by consecutive calls.  A thread that switches streams is detected and made
// safe (the library drains the previous stream before re-using the words) instead of corrupting the counter.
constexpr int MAILBOX_WORDS = 32;

// What a thread predicts from: kept PER VIEW.  The reference's loop renders several keyframes in turn before one
// backward (lioOptimization.cpp:1691-1737), each Camera holding its own world_view_transform tensor for its lifetime
// (camera.cu:36-48): the frame a forward resembles is the previous frame of the SAME camera, not the previous frame of
// the thread.  A view is recognised by the device address of its view matrix and the image size (a guess like every
// other prediction here: a wrong one costs a redo or a far chain, never a result).  A view seen for the first time
// starts from the thread's most recently used history, so a caller that passes a fresh matrix tensor every frame -- or
// renders one camera only -- gets exactly the per-thread behaviour.
struct ViewHist {
  const void* key = nullptr;
  int W = 0, H = 0;
  unsigned long long used = 0;                // LRU stamp (0 = free)
  unsigned generation = 0;                    // bumped when the slot is given to another view
  // speculative binning size: capacity the next forward allocates before it knows its instance count (0 = none)
  uint32_t recent[4] = {0, 0, 0, 0};
  int recent_pos = 0;
  uint32_t recent_far[4] = {0, 0, 0, 0};      // far-phase instance counts of this view's recent near/far frames
  int recent_far_pos = 0;
  bool have_far = false;
  int far_idle_streak = 0;                    // consecutive near/far frames of this view that left no tile live
  // adaptive near budget (budget_feedback): the configured entries per tile x near_scale_q8 / 256
  uint32_t near_scale_q8 = 256;
  int near_hit_run = 0, near_miss_run = 0;
  int split_pause = 0;                        // frames for which this view is binned in one chain again (budget_feedback)
  bool probing = false;                       // the last miss raised the budget: the next outcome says whether it helped
  uint32_t probe_scale_q8 = 256, probe_live = 0;
  int raise_cooldown = 0;                     // misses do not raise the budget while this runs (it did not help)
  bool near_list_too_long = false;            // the last partial depth sort's candidates were most of the scene
  // High-water mark of the predicted binning capacity.  The prediction follows the instance count, which creeps up by a
  // per cent whenever the map grows; asked for a slightly larger blob every few iterations, a caching allocator goes to
  // the driver each time (a multi-GB binning blob: ~10 ms, four views: a 60 ms hiccup).  So a view keeps asking for the
  // capacity it asked for last time until the prediction outgrows it, then jumps by a quarter; a prediction below half
  // of it for 64 frames in a row lets it go.
  uint32_t cap_hw = 0;
  int cap_low_run = 0;
};
constexpr int VIEW_SLOTS = 32;

struct ThreadCtx {
  unsigned long long* mailbox = nullptr;      // host pointer (page-locked, device-mapped)
  unsigned long long* mailbox_dev = nullptr;
  unsigned long long* done_counter = nullptr; // device: (workgroups done << 40 | sum) + 2 x [4][256] digit histograms
  int device = -1;
  uint32_t ticket = 0;
  uint32_t hist_flip = 0;
  hipStream_t last_stream = nullptr;
  bool used = false;
  ViewHist views[VIEW_SLOTS];                 // per-view histories (view_hist), least recently used replaced
  int cur = 0;                                // the view of the thread's current / last forward
  unsigned long long view_clock = 0;
  long long hint_override = -1;               // gsr_set_binning_capacity_hint
  uint32_t last_R = 0;
  long long far_hint_override = -1;           // gsr_set_near_far_hints
  long long near_entries_override = -1;
  uint32_t last_near = 0, last_far = 0;
  bool last_was_near_far = false;
  int far_skip_override = -1;                 // gsr_set_far_speculation: -1 auto, 0 never, 1 the next split forward
  bool last_far_skipped = false;
  // asynchronous near/far frames: a second stream for the far chain, three signal words (0 decide, 1 go), a sequence
  // number, and what the last such frame left to be read from the mailbox once it has got there
  hipStream_t far_stream = nullptr;
  uint32_t* sig_decide = nullptr;             // 2 seq + (far chain needed ? 1 : 0), stored by the near blend
  uint32_t* sig_go = nullptr;                 // seq once the frame is complete
  uint32_t async_seq = 0;
  int async_state = 0;                        // 0 not probed, 1 usable, -1 unavailable
  bool near_count_pending = false;            // the last forward ran a partial depth sort: its candidate count is in the
  uint32_t near_count_ticket = 0;             // mailbox (word 4) -- when it is most of the scene, sort everything again
  int near_count_view = 0;                    // (for that view)
  unsigned near_count_gen = 0;
  // Asynchronous frames return before their far-chain decision is known.  What each left open -- quads unfinished, and
  // if so the far chain's instance count -- arrives in a mailbox slot of the frame's own (words 16 + 2 (seq & 7), + 1),
  // so a thread that runs several forwards ahead of the GPU still learns every outcome (lazy_resolve): up to eight
  // frames pending, oldest first.
  struct Pending { uint32_t ticket, near, total; int w_live, w_far, view; unsigned gen; bool speculated; };
  Pending pending[8];
  int pending_head = 0, pending_n = 0;
  int w_live = 3, w_far = 1;                  // this forward's mailbox words for those two counts (3 / 1 unless lazy)
  uint32_t lazy_seq = 0;                      // frames that returned before their far chain's outcome: slot = seq & 7
  const uint32_t* top_hist = nullptr;         // this forward's [count | tile sum] by top key byte (k_preprocess), or null
  ~ThreadCtx();                               // (a thread that ends gives the asynchronous mechanism back)
};
static thread_local ThreadCtx g_ctx;

static int ctx_prepare(ThreadCtx& c, hipStream_t stream) {
  int cur_device = 0;
  HIP_TRY(hipGetDevice(&cur_device));
  if (!c.mailbox || cur_device != c.device) {  // first call of this thread, or the thread switched GPUs
    void* h = nullptr;
    void* d = nullptr;
    void* k = nullptr;
    if (hipHostMalloc(&h, MAILBOX_WORDS * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&d, h, 0) != hipSuccess || hipMalloc(&k, 64 + 2 * 5120) != hipSuccess ||
        hipMemset(k, 0, 64 + 2 * 5120) != hipSuccess)
      return fail(GSR_ERR_HIP, "cannot allocate the host-mapped mailbox: %s", hipGetErrorString(hipGetLastError()));
    c.mailbox = static_cast<unsigned long long*>(h);  // (a previous device's 256 bytes stay allocated: switching is rare)
    c.mailbox_dev = static_cast<unsigned long long*>(d);
    c.done_counter = static_cast<unsigned long long*>(k);
    c.device = cur_device;
    c.ticket = 0;
    c.hist_flip = 0;
    c.used = false;
    c.async_state = 0;  // (a previous device's stream and signal words stay allocated: switching is rare)
    c.pending_n = 0;
    // word 0: num_rendered; near/far frames: 1 = far count, 2 = near count, 3 = unfinished quads, 4 = near candidates;
    // 16 .. 31: (unfinished quads, far count) of asynchronous frame seq in slot seq & 7
    for (int w = 0; w < MAILBOX_WORDS; w++) c.mailbox[w] = 0;
    (void)hipDeviceSynchronize();  // the counter is zero before any stream uses it
  }
  if (c.used && stream != c.last_stream) {
    // this thread's previous forward ran on another stream: its k_preprocess may still be counting into the words
    // (the caller may have destroyed that stream meanwhile -- work enqueued on a destroyed stream still completes, but
    // its handle cannot be waited on: then the whole device is drained instead)
    if (hipStreamSynchronize(c.last_stream) != hipSuccess) {
      (void)hipGetLastError();
      HIP_TRY(hipDeviceSynchronize());
    }
    if (c.far_stream) HIP_TRY(hipStreamSynchronize(c.far_stream));
  }
  c.last_stream = stream;
  c.used = true;
  c.ticket = c.ticket == 0xFFFFFFFFu ? 1u : c.ticket + 1u;  // never 0: the mailbox starts at ticket 0
  return GSR_OK;
}

// Waits for k_preprocess's (ticket, R) word.  The host polls a page-locked, host-mapped word (no copy engine, no
// completion interrupt, whose wake-up latency was measured at up to 30 ms on virtualised hosts).  Safety net: stream
// queries from 60 us on, every 25 us -- seen on this pool (2 of ~60 bench processes): the word does not become visible
// to the spinning load until a HIP call is made; every such exit is counted and described (gsr_mailbox_slow_path_*).
static int wait_num_rendered(ThreadCtx& c, hipStream_t stream, uint32_t* R_out, int word = 0) {
  using clk = std::chrono::steady_clock;
  const clk::time_point t0 = clk::now();
  const auto query_period = std::chrono::microseconds(25);
  clk::time_point next_query = t0 + std::chrono::microseconds(60);
  // (how the wait is spent, for the slow path's record: see gsr_mailbox_event)
  unsigned queries = 0;
  double longest_query_us = 0.0, longest_gap_us = 0.0;
  clk::time_point t_poll = t0;
  for (;;) {
    const unsigned long long v = __atomic_load_n(c.mailbox + word, __ATOMIC_ACQUIRE);
    if ((uint32_t)(v >> 32) == c.ticket) { *R_out = (uint32_t)v; return GSR_OK; }
    __builtin_ia32_pause();
    const clk::time_point t_now = clk::now();
    longest_gap_us = std::max(longest_gap_us, std::chrono::duration<double, std::micro>(t_now - t_poll).count());
    t_poll = t_now;
    if (t_now < next_query) continue;
    // slow path: notice a faulted or drained stream instead of spinning for ever
    const hipError_t q = hipStreamQuery(stream);
    t_poll = clk::now();
    longest_query_us = std::max(longest_query_us, std::chrono::duration<double, std::micro>(t_poll - t_now).count());
    queries++;
    const unsigned long long v1 = __atomic_load_n(c.mailbox + word, __ATOMIC_ACQUIRE);
    if (q == hipSuccess) {  // everything enqueued has retired, so the word has been stored
      unsigned long long v2 = v1;
      const clk::time_point t1 = clk::now();
      while ((uint32_t)(v2 >> 32) != c.ticket && clk::now() - t1 < std::chrono::milliseconds(2))
        v2 = __atomic_load_n(c.mailbox + word, __ATOMIC_ACQUIRE);  // a store still in flight towards host memory
      {
        std::lock_guard<std::mutex> lk(g_slow_mu);
        g_slow_last.ticket_expected = c.ticket;
        g_slow_last.ticket_seen_before_query = (uint32_t)(v >> 32);
        g_slow_last.ticket_seen_after_query = (uint32_t)(v1 >> 32);
        g_slow_last.elapsed_us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
        g_slow_last.first_query_result = (int)q;
        g_slow_last.visible_at_query = (uint32_t)(v1 >> 32) == c.ticket;
        g_slow_last.queries = queries;
        g_slow_last.longest_query_us = longest_query_us;
        g_slow_last.longest_poll_gap_us = longest_gap_us;
        g_slow_last.count = (unsigned)(g_mailbox_slow_hits.load() + 1);
      }
      const unsigned long long hits = ++g_mailbox_slow_hits;
      if (hits <= 3 || getenv("GSR_HOST_TRACE"))
        fprintf(stderr, "[gsr] mailbox slow path #%llu: ticket %u, seen %u before / %u right after hipStreamQuery (=%d), "
                        "%.1f us after the enqueue (%u queries, the longest took %.1f us; longest gap between two polls "
                        "%.1f us): %s\n", hits, c.ticket, (uint32_t)(v >> 32), (uint32_t)(v1 >> 32), (int)q,
                g_slow_last.elapsed_us, queries, longest_query_us, longest_gap_us, (uint32_t)(v1 >> 32) == c.ticket
                    ? "word present once a HIP call had been made (late dispatch / late visibility to the spinning load)"
                    : "word arrived after the stream had drained (store still in flight)");
      if ((uint32_t)(v2 >> 32) != c.ticket) {
        (void)hipMemset(c.done_counter, 0, 64);  // do not leave a half-counted launch behind
        return fail(GSR_ERR_HIP, "instance count was not published");
      }
      *R_out = (uint32_t)v2;
      return GSR_OK;
    }
    if (q != hipErrorNotReady) return fail(GSR_ERR_HIP, "stream failed: %s", hipGetErrorString(q));
    next_query = clk::now() + query_period;
  }
}

// What gsr_backward -- on whatever host thread autograd runs it -- may know about the forward that filled an image blob:
// ORDER: the backward's tile order (k_tile_order) was computed by that forward; SPLIT: it was a near/far frame (the
// gradient gather then walks the emitted Gaussians' descriptors instead of compacting P flags).  A note exists iff the
// MOST RECENT forward on that blob pointer left it: every forward drops its blob's note on entry.
enum : unsigned { NOTE_ORDER = 1u, NOTE_SPLIT = 2u, NOTE_PRESENT = 4u };
// One entry per image blob with a forward outstanding -- the reference's loop keeps the blobs of several views alive
// until one backward (lioOptimization.cpp:1691-1737), several rendering threads at once -- so the table is a map sized
// by need, not a ring: an entry lives until the next forward on the same blob pointer replaces it.  Blobs that are
// freed and never used again would accumulate, so beyond NOTES_MAX entries the oldest is dropped; a backward that finds
// no entry takes the general path (it sorts its tiles itself and compacts P flags: same results) and is counted
// (gsr_frame_note_misses) -- every forward that reaches its blend leaves an entry, so a miss means an eviction.
constexpr size_t NOTES_MAX = 4096;
struct FrameNote { unsigned flags; unsigned long long age; };
static std::mutex g_note_mu;
static std::unordered_map<const char*, FrameNote> g_notes;
static unsigned long long g_note_clock = 0;
static std::atomic<unsigned long long> g_note_misses{0};
static void order_forget(const char* blob) {
  std::lock_guard<std::mutex> lk(g_note_mu);
  g_notes.erase(blob);
}
static void frame_note(const char* blob, unsigned flags) {
  std::lock_guard<std::mutex> lk(g_note_mu);
  auto it = g_notes.find(blob);
  if (it != g_notes.end()) { it->second.flags |= flags; return; }
  if (g_notes.size() >= NOTES_MAX) {
    auto oldest = g_notes.begin();
    for (auto e = g_notes.begin(); e != g_notes.end(); ++e)
      if (e->second.age < oldest->second.age) oldest = e;
    g_notes.erase(oldest);
  }
  g_notes.emplace(blob, FrameNote{flags, ++g_note_clock});
}
static void order_remember(const char* blob) { frame_note(blob, NOTE_ORDER); }
static unsigned frame_notes(const char* blob) {
  std::lock_guard<std::mutex> lk(g_note_mu);
  auto it = g_notes.find(blob);
  if (it == g_notes.end()) { ++g_note_misses; return 0u; }
  return it->second.flags;
}

// ---- asynchronous near/far frames ------------------------------------------------------------------------------
// Stream-side hand-off instead of a host round trip (hipStreamWaitValue32, where the device offers it): the far chain
// is enqueued on the library's own stream behind a wait for the near blend's decision word, every kernel of it gated
// on "needed", and the caller's stream waits for the frame's go word -- which the near blend itself stores when no
// quad was left unfinished.  The host never waits for the decision.  GSR_ASYNC_FAR=0 switches the mechanism off (the
// host then reads the count of unfinished quads from the mailbox and enqueues the far chain if there are any).
// ONE host thread per process gets the mechanism (the first that asks): a parked far chain waits for a kernel of its own
// frame on another stream, and with several such pairs multiplexed onto the device's few hardware queues by racing
// threads, two chains could end up parked in front of each other's frames.  A single pair is always submitted in an
// order the queues can execute.  Other threads take the host-decided variant.
static std::atomic<const void*> g_async_owner{nullptr};

// A host thread that ends: its far stream is drained and destroyed, its signal words freed, and the mechanism is free
// for another thread.  (Left behind, the stream of an ended thread kept the process from exiting: seen with two
// rendering threads under pytest.)
ThreadCtx::~ThreadCtx() {
  if (far_stream) {
    (void)hipStreamSynchronize(far_stream);  // (every wait on it is satisfied by a frame that has been enqueued in full)
    (void)hipStreamDestroy(far_stream);
    far_stream = nullptr;
  }
  if (sig_decide) (void)hipFree(sig_decide);
  if (sig_go) (void)hipFree(sig_go);
  sig_decide = sig_go = nullptr;
  const void* me = this;
  (void)g_async_owner.compare_exchange_strong(me, nullptr);
  // the thread's mailbox and counter words (hipFree waits for the device: kernels of this thread's last frames may
  // still be counting into them)
  if (done_counter) (void)hipFree(done_counter);
  if (mailbox) (void)hipHostFree(mailbox);
  done_counter = nullptr;
  mailbox = mailbox_dev = nullptr;
  (void)hipGetLastError();
}

static bool async_far_ready(ThreadCtx& c) {
  if (c.async_state) return c.async_state > 0;
  c.async_state = -1;
  const char* e = getenv("GSR_ASYNC_FAR");
  if (e && e[0] == '0') return false;
  const void* none = nullptr;
  if (!g_async_owner.compare_exchange_strong(none, &c) && none != &c) return false;
  int can = 0;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c.device) != hipSuccess || !can) {
    (void)hipGetLastError();
    return false;
  }
  void *a = nullptr, *b = nullptr;
  hipStream_t st = nullptr;
  if (hipExtMallocWithFlags(&a, 8, hipMallocSignalMemory) != hipSuccess ||
      hipExtMallocWithFlags(&b, 8, hipMallocSignalMemory) != hipSuccess || hipMemset(a, 0, 8) != hipSuccess ||
      hipMemset(b, 0, 8) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return false;
  }
  (void)hipDeviceSynchronize();
  // The API is marked Beta: before a caller's stream is ever made to wait on it, the whole hand-off is rehearsed once on
  // two streams of the library's own -- kernel stores `decide`, the far stream's wait passes, its kernel stores `go`,
  // the other stream's wait passes -- and must have completed within three seconds (the first launch loads code objects).  If not, the mechanism stays off for this
  // thread (the rehearsal's streams are released and abandoned).
  {
    hipStream_t probe = nullptr;
    (void)hipGetLastError();  // (a stale error of some earlier call must not be taken for one of these launches')
    int step = 0;
    bool ok = hipStreamCreateWithFlags(&probe, hipStreamNonBlocking) == hipSuccess;
    uint32_t* decide = static_cast<uint32_t*>(a);
    uint32_t* go = static_cast<uint32_t*>(b);
    // (enqueued in the order of the dependencies, as a frame's operations are: streams share the device's few hardware
    // queues, and a wait enqueued ahead of the kernel it waits for would block that kernel if both landed in one queue)
    if (ok) { step = 1; ok = launch_release_go(Count{nullptr, 0}, decide, 2u, probe) == hipSuccess; }
    if (ok) { step = 2; ok = hipStreamWaitValue32(st, decide, 2u, hipStreamWaitValueGte) == hipSuccess; }
    if (ok) { step = 3; ok = launch_release_go(Count{nullptr, 0}, go, 1u, st) == hipSuccess; }
    if (ok) { step = 4; ok = hipStreamWaitValue32(probe, go, 1u, hipStreamWaitValueGte) == hipSuccess; }
    if (ok) {
      step = 5;
      const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      hipError_t q = hipErrorNotReady, q2 = hipErrorNotReady;
      while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(3000)) {
        if (q == hipErrorNotReady) q = hipStreamQuery(probe);
        if (q2 == hipErrorNotReady) q2 = hipStreamQuery(st);
        if (q != hipErrorNotReady && q2 != hipErrorNotReady) break;
      }
      ok = q == hipSuccess && q2 == hipSuccess;
    }
    if (!ok) {
      (void)hipGetLastError();
      fprintf(stderr, "[gsr] stream-side waits did not complete their rehearsal (step %d): far-chain speculation stays on "
                      "the host\n", step);
      // do not leave a parked stream behind (it would keep the process from exiting): both words to a value that
      // satisfies every wait, then the streams are abandoned
      (void)hipMemset(a, 0xFF, 8);
      (void)hipMemset(b, 0xFF, 8);
      // (destroying a stream does not wait: its resources go once its work has completed -- the two signal words stay
      // allocated, 16 bytes, since a wait may still be looking at them) and another thread may try its luck
      if (probe) (void)hipStreamDestroy(probe);
      (void)hipStreamDestroy(st);
      (void)hipGetLastError();
      const void* me = &c;
      (void)g_async_owner.compare_exchange_strong(me, nullptr);
      return false;
    }
    (void)hipStreamDestroy(probe);
    if (hipMemset(a, 0, 8) != hipSuccess || hipMemset(b, 0, 8) != hipSuccess) return false;
    (void)hipDeviceSynchronize();
  }
  c.sig_decide = static_cast<uint32_t*>(a);
  c.sig_go = static_cast<uint32_t*>(b);
  c.far_stream = st;
  c.async_seq = 0;
  c.async_state = 1;
  return true;
}

static bool peek_word(const ThreadCtx& c, int word, uint32_t ticket, uint32_t* out) {
  const unsigned long long v = __atomic_load_n(c.mailbox + word, __ATOMIC_ACQUIRE);
  if ((uint32_t)(v >> 32) != ticket) return false;
  *out = (uint32_t)v;
  return true;
}

static std::atomic<unsigned long long> g_far_skips{0}, g_far_skip_misses{0}, g_async_frames{0};

// Adaptive near budget (per view).  A frame whose near chain leaves quads unfinished pays for a far chain -- its
// launches, the depth order of the far Gaussians, its instances -- and does not qualify for the far-chain speculation; a
// budget that is larger than necessary only costs the near chain's extra instances (2 M Gaussians / 1080p: 480 instead
// of 320 entries per tile +2.6 % of a step).  So a miss raises the view's budget by a quarter of the configured one, up
// to three times it -- ON PROBATION: the next frame of the view tells whether the raise helped.  If fewer than a quarter
// of the unfinished quads went away, those tiles do not finish for lack of budget -- sky, the border of the map, a sparse
// corner: nothing behind the near Gaussians covers them densely -- and a bigger near chain only costs (a rotating
// camera over the C3 scene: 7.8 M near instances at the cap instead of 2.6 M, +25 % per view); the raise is taken
// back and no other is tried for 256 frames of that view.  A long run of frames without a far chain (64) takes a
// sixteenth back, down to the configured budget.  (The same scene after a few dozen optimiser steps, or under the
// photometric loss, needs 360-480 entries per tile: 1.15 -> 0.91 and 1.04 -> 1.00 ms/step.)  Eight misses in a row of
// frames that splitting does not shorten by a quarter (near + far instances against all the frame's, or a far chain of
// four times the near chain) pause the splitting of that view for 256 frames.  Not while a test hook sets the budget.
static void budget_feedback(const ThreadCtx& c, ViewHist& h, uint32_t live, uint32_t R_near, uint32_t R_far,
                            uint32_t R_total = 0u) {
  if (c.near_entries_override >= 0) return;
  if (h.raise_cooldown > 0) h.raise_cooldown--;
  if (live == 0u) {
    h.near_miss_run = 0;
    h.probing = false;  // (a raise on probation has finished every quad: kept)
    if (++h.near_hit_run >= 64 && h.near_scale_q8 > 256u) {
      h.near_scale_q8 -= 16u;
      h.near_hit_run = 0;
    }
    return;
  }
  h.near_hit_run = 0;
  const bool sparse = (unsigned long long)R_far >= 4ull * (unsigned long long)R_near;
  const bool useless = sparse || (R_total != 0u && ((unsigned long long)R_near + R_far) * 4ull >= 3ull * R_total);
  bool reverted = false;
  if (h.probing) {  // the previous miss raised the budget: did that finish at least a quarter of the quads it left?
    h.probing = false;
    if ((unsigned long long)live * 4ull > (unsigned long long)h.probe_live * 3ull) {
      h.near_scale_q8 = h.probe_scale_q8;
      h.raise_cooldown = 256;
      reverted = true;
    }
  }
  if (!sparse && !reverted && h.raise_cooldown == 0 && h.near_scale_q8 < 768u) {
    h.probe_scale_q8 = h.near_scale_q8;
    h.probe_live = live;
    h.probing = true;
    h.near_scale_q8 = h.near_scale_q8 + 64u < 768u ? h.near_scale_q8 + 64u : 768u;
  }
  if (!useless) {
    h.near_miss_run = 0;
  } else if (++h.near_miss_run >= 8) {
    h.split_pause = 256;
    h.near_miss_run = 0;
  }
}

// The history the calling thread keeps for a view (see ViewHist).  A new view inherits the most recently used one's.
static ViewHist& view_hist(ThreadCtx& c, const void* key, int W, int H) {
  int hit = -1, lru = 0, mru = -1;
  for (int k = 0; k < VIEW_SLOTS; k++) {
    const ViewHist& v = c.views[k];
    if (v.used && v.key == key && v.W == W && v.H == H) { hit = k; continue; }
    if (v.used < c.views[lru].used || lru == hit) lru = k;
    if (v.used && (mru < 0 || v.used > c.views[mru].used)) mru = k;
  }
  const bool known = hit >= 0 && (c.views[hit].recent[0] | c.views[hit].recent[1] | c.views[hit].recent[2] |
                                  c.views[hit].recent[3]) != 0u;
  const int slot = hit >= 0 ? hit : lru;
  ViewHist& v = c.views[slot];
  if (!known) {  // a new view, or one whose history was forgotten: the thread's most recent history
    const unsigned gen = hit >= 0 ? v.generation : v.generation + 1u;
    v = mru >= 0 ? c.views[mru] : ViewHist();
    v.key = key; v.W = W; v.H = H;
    v.generation = gen;
  }
  v.used = ++c.view_clock;
  c.cur = slot;
  return v;
}

// What an asynchronous frame left open when gsr_forward returned -- did its far chain run, and over how many
// instances -- is read from the frame's mailbox slot the next time the thread asks (never waited for).  Frames are
// resolved oldest first; one whose words have not arrived stays pending (it is NOT dropped: a thread that runs ahead of
// the GPU, the steady state this mechanism exists for, would otherwise never count a miss nor adapt its budget).
static std::atomic<unsigned long long> g_async_outcomes_lost{0};
static void lazy_resolve(ThreadCtx& c) {
  while (c.pending_n > 0) {
    const ThreadCtx::Pending& p = c.pending[c.pending_head];
    uint32_t live = 0, far = 0;
    if (!peek_word(c, p.w_live, p.ticket, &live)) return;
    if (live != 0u && !peek_word(c, p.w_far, p.ticket, &far)) return;
    if (p.speculated) {  // (the speculation's own score: frames that enqueued their far chain outright are not counted)
      if (live == 0u) ++g_far_skips;
      else ++g_far_skip_misses;
    }
    ViewHist& h = c.views[p.view];
    if (h.generation == p.gen) {  // (unless the slot has gone to another view meanwhile)
      budget_feedback(c, h, live, p.near, far, p.total);
      h.recent_far[h.recent_far_pos] = far;
      h.recent_far_pos = (h.recent_far_pos + 1) & 3;
      h.have_far = true;
      h.far_idle_streak = live == 0u ? h.far_idle_streak + 1 : 0;
    }
    if (p.ticket == c.ticket) {  // the thread's most recent forward: what gsr_last_* report
      c.last_far = far;
      c.last_far_skipped = p.speculated && live == 0u;
      c.last_R = c.last_near + c.last_far;
    }
    c.pending_head = (c.pending_head + 1) & 7;
    c.pending_n--;
  }
}
static void lazy_push(ThreadCtx& c, uint32_t ticket, uint32_t near, uint32_t total, bool speculated) {
  if (c.pending_n == 8) {  // nine frames in flight: the oldest one's slot is about to be reused -- its outcome is lost,
    ++g_async_outcomes_lost;  // which is taken for a miss (the speculation has to earn its streak again)
    const ThreadCtx::Pending& p = c.pending[c.pending_head];
    if (c.views[p.view].generation == p.gen) c.views[p.view].far_idle_streak = 0;
    c.pending_head = (c.pending_head + 1) & 7;
    c.pending_n--;
  }
  c.pending[(c.pending_head + c.pending_n) & 7] =
      ThreadCtx::Pending{ticket, near, total, c.w_live, c.w_far, c.cur, c.views[c.cur].generation, speculated};
  c.pending_n++;
}

// Are several host threads rendering at the moment?  (Two hand-overs of "the thread that entered gsr_forward last" within
// the last 100 ms: a loop that renders its views from T > 1 threads switches several times per iteration, a process
// whose only rendering thread changes once does not count.)  Such a process gets no asynchronous frames: with other
// threads' work on the GPU the host-decided far chain's wait costs nothing (three views from three threads, 2 M Gaussians /
// 1080p: 0.698 ms per view against 0.744 with one thread's frames asynchronous), and it keeps the second stream and the
// stream-side waits -- a Beta interface -- out of a process whose five or more streams already share the device's few
// hardware queues.  (That combination is where the one GPU memory fault of round 3 appeared, DESIGN.md section 4: three
// rendering threads, reproducible, gone with GSR_ASYNC_FAR=0, with serialised kernels, and with a slower backward.)
static std::mutex g_switch_mu;
static const void* g_last_forward_thread = nullptr;
static std::chrono::steady_clock::time_point g_switch_time[2];  // the last two hand-overs, most recent first
static bool several_threads_render(const void* me) {
  const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> lk(g_switch_mu);
  if (g_last_forward_thread != me) {
    if (g_last_forward_thread != nullptr) {
      g_switch_time[1] = g_switch_time[0];
      g_switch_time[0] = now;
    }
    g_last_forward_thread = me;
  }
  return g_switch_time[1] != std::chrono::steady_clock::time_point() &&
         now - g_switch_time[1] < std::chrono::milliseconds(100);
}

// One binning chain: scan -> emit -> tile sort -> ranges, followed by the blend.  A whole frame is one chain over the
// blob (phase 0).  A near/far frame (gsr_forward) runs two chains over ONE blob carved for capA + capB instances:
// phase 1 bins the near Gaussians into slots / list positions [0, capA), phase 2 the far Gaussians that still matter
// into [capA, capA + capB); the sort scratch is shared (the chains are ordered on the stream).
struct Chain {
  int phase;             // 0 whole frame, 1 near, 2 far
  Count cnt;             // instances of this chain
  uint32_t near_budget;  // phase 1: the near phase ends with the Gaussian in whose slot run this falls
  uint32_t base;         // first slot / list position of this chain in the frame's slot space (phase 2: capA)
  uint32_t expect = 0;   // instances this chain is expected to hold (0 = its capacity): picks the form of the tile sort
};

struct DepthOrder {
  const uint32_t* near_order;  // what the near / whole-frame scan walks: g.order, or the sorted near candidates
  bool partial;                // partial depth sort: the full order does not exist yet when the far chain starts
  const uint32_t* ghist;       // k_preprocess's digit counts of all P keys (for that full sort), or null
};

static int enqueue_chain(const FrameParams& fp, GeomState& g, ImageState& im, BinningState& b, const Chain& ch,
                         ThreadCtx& c, const DepthOrder& dord, const float* background, float* out_color,
                         float* out_depth, float* out_acc, int debug, hipStream_t stream,
                         AsyncWords aw = AsyncWords()) {
  const Count cnt = ch.cnt;  // (carries the gate of an asynchronous frame's far chain)
  const int tiles = fp.gx * fp.gy;
  const int tile_bits = (int)gsr_higher_msb((uint32_t)tiles);  // the `bit` of rasterizer_impl.cu:295
  const bool start_in_A = (sort_passes(tile_bits) % 2) == 0;
  const bool key16 = tiles <= 65536;  // tile ids fit 16 bits for every image up to 4096 x 4096
  const bool far = ch.phase == 2;
  const uint4* sdesc = far ? g.sdescB : g.sdesc;
  uint32_t* chunk_first = far ? b.chunk_firstB : b.chunk_first;
  uint32_t* point_list = b.point_list + ch.base;
  uint8_t* inst_flag = b.inst_flag + ch.base;
  uint2* ranges = far ? im.rangesB : im.ranges;
  if (far) {
    if (dord.partial) {
      // the far chain walks the far Gaussians in depth order: the full sort a partial depth sort has left out.  It re-uses
      // the pass tickets and look-back status words the near sort has used (words 1024 .. scan_off of the scratch);
      // both launches carry this chain's gate
      Count all = cnt;
      all.dev = nullptr;
      all.cap = fp.P;
      STAGE(launch_clear_words(all, g.dsort.words + 1024, g.dsort.scan_off - 1024, stream));
      STAGE(launch_depth_sort(g.dkeysA, g.order, g.dkeysB, g.dvalsB, g.dsort, all, dord.ghist,
                              /*vals_are_positions=*/true, stream));
    }
    STAGE(launch_live_sat(fp, im, g.total + 9, cnt, stream));  // which tiles did the near chain leave unfinished
    STAGE(launch_scan_offsets_far(fp, g, cnt, ch.base, im.live_sat, chunk_first, b.tsort.counts,
                                  c.mailbox_dev + c.w_far, c.ticket, stream));
  }
  else
    STAGE(launch_scan_offsets(fp, g, cnt, chunk_first, im.ranges, im.rangesB, b.tsort.counts, ch.near_budget,
                              ch.phase == 1 ? c.mailbox_dev + 2 : nullptr, c.ticket, ch.phase == 1 ? c.top_hist : nullptr,
                              ch.phase == 1 ? dord.near_order : g.order, ch.phase == 1 && dord.partial, stream));
  // bucket form of the tile sort (gsr_internal.hpp): first pass on the top eight bits, then one launch per chain that
  // finishes every bucket and writes the ranges
  // (a far chain's capacity may be every instance behind the near budget while it usually holds a few per cent of that:
  // the form of the sort follows what is expected; either form is correct for any count)
  const bool buckets = tile_sort_buckets(tile_bits, key16, ch.expect ? (int)std::min<uint32_t>(ch.expect, (uint32_t)cnt.cap)
                                                                     : cnt.cap);
  const uint32_t shift0 = buckets ? (uint32_t)(tile_bits - 8) : 0u;
  const uint32_t mask0 = buckets ? 255u : (1u << sort_digit_bits(tile_bits)) - 1u;
  STAGE(launch_emit(fp, sdesc, cnt, chunk_first, start_in_A ? b.tkeysA : b.tkeysB, start_in_A ? point_list : b.ivalsB,
                    inst_flag, b.tsort.counts, shift0, mask0, key16,
                    /*store_pairs=*/!key16, stream));  // 16-bit keys: the pairs are generated inside the first sort pass
  const EmitFusion ef = {fp, sdesc, cnt, chunk_first};
  // 16-bit keys and at least two passes: the last pass counts the instances of every tile into the zeroed ranges
  // instead of writing the sorted keys, and a one-workgroup scan turns the counts into ranges; otherwise the range
  // kernel reads the sorted keys as the reference's identifyTileRanges does
  static const bool ranges_from_keys = getenv("GSR_RANGES_FROM_KEYS") != nullptr;  // diagnostics / fallback
  const bool count_ranges = key16 && sort_passes(tile_bits) >= 2 && !ranges_from_keys;
  const BucketPass bp = {ranges, tiles, ch.base};
  STAGE(launch_sort_pairs(b.tkeysA, point_list, b.tkeysB, b.ivalsB, b.tsort, cnt, tile_bits, start_in_A,
                          /*is_depth_sort=*/false, key16, /*first_hist_done=*/true, key16 ? &ef : nullptr,
                          count_ranges ? reinterpret_cast<uint32_t*>(ranges) : nullptr, stream, buckets ? &bp : nullptr));
  if (buckets) {
    // (the ranges are written by the bucket pass)
  } else if (count_ranges)
    STAGE(launch_ranges_from_counts(ranges, tiles, ch.base, cnt, stream));
  else
    STAGE(launch_tile_ranges(b.tkeysA, cnt, ranges, key16, ch.base, stream));
  if (debug && ch.phase == 0) {  // self-check of the binning chain: every list ordered by (depth bits, id)
    HIP_TRY(launch_verify_sorted_lists(im.ranges, tiles, b.point_list, g.splats, g.total + 4, stream));
    uint32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, g.total + 4, sizeof(bad), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (bad) return fail(GSR_ERR_HIP, "%u adjacent list entries out of (depth, id) order after the sort", bad);
  }
  // (the near blend counts the quads it leaves unfinished into the host's mailbox: done word 1, mailbox word 3)
  STAGE(launch_blend_forward(fp, g, b, im, background, out_color, out_depth, out_acc, ch.phase, c.done_counter + 1,
                             c.mailbox_dev + c.w_live, c.ticket, ch.phase == 1 ? aw : AsyncWords(), cnt, stream));
  return GSR_OK;
}

// near/far frames (gsr_set_near_far): 1 = allowed (default), 0 = never
static std::atomic<int>& near_far_flag() {
  static std::atomic<int> flag([] {
    const char* e = getenv("GSR_NEAR_FAR");
    return (e && e[0] == '0') ? 0 : 1;
  }());
  return flag;
}
static std::atomic<unsigned long long> g_near_far_forwards{0};

int gsr_forward(gsr_alloc_fn geometry_alloc, void* geometry_ctx, gsr_alloc_fn binning_alloc, void* binning_ctx,
                gsr_alloc_fn image_alloc, void* image_ctx, int P, int D, int M, const float* background, int width,
                int height, const float* means3D, const float* shs, const float* colors_precomp,
                const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered, float* out_color, float* out_depth, float* out_acc,
                int* radii, int debug, void* stream_) {
  (void)prefiltered;  // no effect in the reference forward either
  hipStream_t stream = (hipStream_t)stream_;
  g_err[0] = 0;
  if (P < 0 || width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P/width/height");
  if (!out_color || !out_depth || !out_acc) return fail(GSR_ERR_INVALID_ARGUMENT, "null output image");
  const size_t N = (size_t)width * height;
  if (P == 0) {  // the Torch glue never calls the rasterizer for P == 0 and returns zero images
                 // (rasterize_points.cu:78-93); same result here.
    HIP_TRY(hipMemsetAsync(out_color, 0, 3 * N * sizeof(float), stream));
    HIP_TRY(hipMemsetAsync(out_depth, 0, N * sizeof(float), stream));
    HIP_TRY(hipMemsetAsync(out_acc, 0, N * sizeof(float), stream));
    g_ctx.last_R = 0;
    return 0;
  }
  if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(GSR_ERR_INVALID_ARGUMENT, "null allocator");
  if (!means3D || !opacities || !background || !viewmatrix || !projmatrix || !cam_pos)
    return fail(GSR_ERR_INVALID_ARGUMENT, "null required input");
  if (!colors_precomp && !shs)  // reference: NUM_CHANNELS/colour check, rasterizer_impl.cu:229-231
    return fail(GSR_ERR_INVALID_ARGUMENT, "provide SHs or precomputed colours");
  if (!cov3D_precomp && (!scales || !rotations))
    return fail(GSR_ERR_INVALID_ARGUMENT, "provide scales+rotations or a precomputed 3D covariance");
  if (!colors_precomp && (D < 0 || D > 3 || M < (D + 1) * (D + 1) || M > 16))
    return fail(GSR_ERR_UNSUPPORTED, "SH degree %d with %d coefficients is not supported (D<=3, (D+1)^2<=M<=16)", D, M);
  if (width > 1023 * TILE || height > 1023 * TILE) return fail(GSR_ERR_UNSUPPORTED, "image larger than 16368 px");
  if (P >= (1 << 30)) return fail(GSR_ERR_UNSUPPORTED, "2^30 or more Gaussians");
  if (!(tan_fovx > 0.f) || !(tan_fovy > 0.f)) return fail(GSR_ERR_INVALID_ARGUMENT, "tan_fov must be > 0");

  const FrameParams fp = make_params(P, D, M, width, height, tan_fovx, tan_fovy, scale_modifier);
  char* gblob = geometry_alloc(geometry_ctx, gsr_geometry_bytes(P));
  if (!gblob) return fail(GSR_ERR_ALLOC, "geometry allocator returned NULL");
  char* iblob = image_alloc(image_ctx, gsr_image_bytes(width, height));
  if (!iblob) return fail(GSR_ERR_ALLOC, "image allocator returned NULL");
  GeomState g = GeomState::carve(gblob, (size_t)P);
  ImageState im = ImageState::carve(iblob, width, height);
  order_forget(iblob);

  // The instance count R sizes the binning blob, where the reference has its blocking cudaMemcpy
  // (rasterizer_impl.cu:277).  The last workgroup of k_preprocess stores (ticket, R) into a page-locked, host-mapped
  // word that the host polls (wait_num_rendered).
  //   * synchronous forward (a thread's first forward, debug forwards, GSR_SYNC_FORWARD=1): the host waits for the
  //     word right after enqueueing k_preprocess and the depth sort -- which does not depend on R and runs while the
  //     host sizes and allocates the blob -- and then enqueues the rest with the exact count;
  //   * speculative forward (the steady state): the blob is allocated for a capacity predicted from this thread's
  //     recent forwards, EVERYTHING is enqueued at once with the kernels reading R from device memory (Count), and the
  //     host reads the word only at the very end, when it has long been written: no host wait, and the GPU never
  //     waits for the host.  If R exceeds the capacity the kernels have clamped to it (in-bounds garbage); the
  //     host then allocates an exact blob and enqueues the binning chain again -- the only cost of a misprediction.
  ThreadCtx& c = g_ctx;
  // (GSR_ASYNC_FAR_MT=1, diagnostics: asynchronous frames although several threads render -- the combination that faulted)
  static const bool env_async_mt = getenv("GSR_ASYNC_FAR_MT") != nullptr;
  const bool multi_thread = several_threads_render(&c) && !env_async_mt;
  lazy_resolve(c);  // (what the thread's earlier asynchronous frames left open, as far as the mailbox has it by now)
  c.w_live = 3;
  c.w_far = 1;
  if (c.near_count_pending) {  // (likewise the previous partial depth sort's candidate count)
    uint32_t nn = 0;
    ViewHist& hv = c.views[c.near_count_view];
    if (peek_word(c, 4, c.near_count_ticket, &nn) && hv.generation == c.near_count_gen)
      hv.near_list_too_long = (unsigned long long)nn * 3ull > (unsigned long long)P;
    c.near_count_pending = false;
  }
  ViewHist& h = view_hist(c, viewmatrix, width, height);  // what this view's recent frames predict
  {
    const int rc = ctx_prepare(c, stream);
    if (rc != GSR_OK) return rc;
  }
  // Digit histograms of the depth sort, counted by k_preprocess: two library-owned [4][256] buffers behind the
  // counter, used alternately -- a forward counts into one (zero on entry) and clears the other for the next
  // forward of this thread, so a call that ends early never leaves a dirty buffer in the way.
  // GSR_DEPTH_HIST_PASS=1 restores the sort's own histogram pass (k_sort_hist_all).
  static const bool env_hist_pass = getenv("GSR_DEPTH_HIST_PASS") != nullptr;
  const bool own_hist_pass = env_hist_pass || !preprocess_counts_depth_digits(fp, shs, colors_precomp);
  uint32_t* const ghist2 = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(c.done_counter) + 64);
  if (!own_hist_pass) c.hist_flip ^= 1u;  // only a call that uses the pair advances it (the other buffer is clean)
  // (5 rows of 256: the four digit histograms + the tile counts summed by top byte, k_preprocess)
  uint32_t* const ghist_acc = own_hist_pass ? nullptr : ghist2 + 1280 * c.hist_flip;
  uint32_t* const ghist_clear = own_hist_pass ? nullptr : ghist2 + 1280 * (c.hist_flip ^ 1u);
  STAGE(launch_preprocess(fp, means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, viewmatrix,
                          projmatrix, cam_pos, g, radii, /*write_cov3D=*/debug != 0, c.done_counter, c.mailbox_dev, c.ticket,
                          ghist_acc, ghist_clear, stream));
  c.top_hist = ghist_acc ? ghist_acc + 3 * 256 : nullptr;  // rows 3 (counts by top byte) and 4 (tile sums) are adjacent
  if (debug) STAGE(launch_point_offsets(fp, g, stream));  // the reference's array, for the views only

  static const bool env_sync = getenv("GSR_SYNC_FORWARD") != nullptr;
  static const bool host_trace = getenv("GSR_HOST_TRACE") != nullptr;  // diagnostics: host-side waits
  uint32_t hint = 0, pred = 0;  // capacity to allocate for / instance count predicted (the decisions below use the latter)
  if (c.hint_override >= 0) {
    hint = pred = (uint32_t)c.hint_override;
    c.hint_override = -1;
  } else {
    for (int k = 0; k < 4; k++) hint = h.recent[k] > hint ? h.recent[k] : hint;
    if (hint) {
      hint = round_capacity((uint32_t)std::min<unsigned long long>(0x7fffffffull, (unsigned long long)hint * 5 / 4 + 65536));
      pred = hint;
      if (hint > h.cap_hw) {  // (ViewHist::cap_hw)
        h.cap_hw = round_capacity((uint32_t)std::min<unsigned long long>(0x7fffffffull, (unsigned long long)hint * 5 / 4));
        h.cap_low_run = 0;
      } else if (hint < h.cap_hw / 2u) {
        if (++h.cap_low_run >= 64) { h.cap_hw = hint; h.cap_low_run = 0; }
      } else {
        h.cap_low_run = 0;
      }
      hint = h.cap_hw;
    } else {
      h.cap_hw = 0;
    }
  }
  const bool speculate = !debug && !env_sync && hint > 0;
  const std::chrono::steady_clock::time_point t_enq = std::chrono::steady_clock::now();
  uint32_t R_host = 0;
  int key = 0;  // what the caller passes back to gsr_backward: the capacity the binning blob was carved for
  c.last_was_near_far = false;
  // Near/far frame (speculative forwards in the default binning mode, when the predicted instance count is at least
  // three times the near budget): the tiles' lists are depth-ordered and a pixel stops reading its list once its
  // transmittance falls below 1e-4 (forward.cu:380-383) -- at 2 M Gaussians / 1080p every tile is finished after
  // ~3 % of its list, and emitting, sorting and ranging the other 97 % is most of the forward.  So the frame is binned
  // in two chains over the Gaussians in depth order: the NEAR chain takes Gaussians until they fill a budget of
  // `near_entries` list entries per tile on average, is sorted and blended; then only the FAR Gaussians whose tile
  // rectangle still contains an unfinished tile are binned and blended on top (k_scan_offsets_far).  Instances that
  // are not emitted lie, in every tile they would have gone to, behind the point where every pixel has stopped: the
  // images, n_contrib, final_T and all gradients are bit-identical to the one-chain frame (tested), only the lists
  // are shorter.  gsr_set_reference_rects(1) frames are never split: their lists are the reference's, whole.
  static const long env_near_entries = getenv("GSR_NEAR_ENTRIES") ? atol(getenv("GSR_NEAR_ENTRIES")) : 320;
  const int tiles_n = fp.gx * fp.gy;
  const long long near_entries = c.near_entries_override >= 0
                                     ? c.near_entries_override
                                     : (env_near_entries * (long long)h.near_scale_q8 + 255) / 256;  // (budget_feedback)
  const unsigned long long budget64 = (unsigned long long)tiles_n * (unsigned long long)near_entries;
  bool split_paused = false;
  if (h.split_pause > 0 && c.near_entries_override < 0) {  // (budget_feedback: this view's splits kept missing)
    h.split_pause--;
    split_paused = true;
  }
  // (three times: BASELINE C2 -- 500 k Gaussians at 1280x720, 3.6 M instances, 3.1 budgets -- bins 1.15 M of them and its
  // forward goes 0.31 -> 0.245 ms; at 1.5 budgets, the 640x512 shape, a split saves nothing.  In quarters:)
  static const unsigned long long env_ratio_q2 =
      getenv("GSR_NEAR_FAR_MIN_RATIO_Q2") ? strtoull(getenv("GSR_NEAR_FAR_MIN_RATIO_Q2"), nullptr, 10) : 12ull;
  const bool near_far = speculate && (t_near_far >= 0 ? t_near_far != 0 : near_far_flag().load() != 0) && !fp.ref_rects && near_entries > 0 &&
                        budget64 < 0x20000000ull && !split_paused &&
                        (c.near_entries_override >= 0 || 4ull * (unsigned long long)pred >= env_ratio_q2 * budget64);  // (hook: always)
  // Far-chain speculation (see the near/far branch below): after two split frames in a row that left no quad
  // unfinished (or when the test hook asks) the thread's next split frame expects its far chain to stay idle.
  const bool speculate_far = near_far && (c.far_skip_override >= 0 ? c.far_skip_override == 1 : h.far_idle_streak >= 2);
  const bool speculation_forced = near_far && c.far_skip_override == 1;  // (test hook: also overrides the guard below)
  if (speculation_forced) c.far_skip_override = -1;
  // Depth order of the Gaussians.  Such a frame needs it for the NEAR candidates only (k_compact_near): they are
  // compacted into the sort's second buffer pair and sorted there (partial depth sort: at 2 M Gaussians / 1080p some
  // 50 000 pairs instead of 2 M, 0.11 -> 0.05 ms); the full sort moves into the far chain, for the frames that run it.
  static const bool env_full_sort = getenv("GSR_FULL_DEPTH_SORT") != nullptr;  // diagnostics / fallback
  // (not while the candidates are more than a third of the scene -- the line is drawn at top-byte granularity, a factor
  // of four in depth: the last partial sort's count says so; every 64th frame tries again)
  if (h.near_list_too_long && (c.ticket & 63u) == 0u) h.near_list_too_long = false;
  const bool partial_sort = speculate_far && c.top_hist != nullptr && !env_full_sort &&
                            (!h.near_list_too_long || speculation_forced);
  const uint32_t* near_order = g.order;
  if (partial_sort) {
    STAGE(launch_compact_near(fp, g, c.top_hist, (uint32_t)budget64, g.dkeysB, g.dvalsB, g.total + 15,
                              g.dsort.ghist_near((size_t)P), c.mailbox_dev + 4, c.ticket, stream));
    c.near_count_pending = true;
    c.near_count_ticket = c.ticket;
    c.near_count_view = c.cur;
    c.near_count_gen = h.generation;
    STAGE(launch_depth_sort(g.dkeysB, g.dvalsB, g.nkeys2, g.nvals2, g.dsort, Count{g.total + 15, P},
                            g.dsort.ghist_near((size_t)P), /*vals_are_positions=*/false, stream));
    near_order = g.dvalsB;
  } else {
    STAGE(launch_depth_sort(g.dkeysA, g.order, g.dkeysB, g.dvalsB, g.dsort, Count{nullptr, P}, ghist_acc,
                            /*vals_are_positions=*/true, stream));
  }
  // what a chain needs to know about the depth order (enqueue_chain)
  const DepthOrder dord = {near_order, partial_sort, ghist_acc};
  if (!speculate) {
    const int rc = wait_num_rendered(c, stream, &R_host);
    if (rc != GSR_OK) return rc;
    if (R_host > 0x7fffffffu) return fail(GSR_ERR_UNSUPPORTED, "more than 2^31 splat instances");
    const std::chrono::steady_clock::time_point ta = std::chrono::steady_clock::now();
    char* bblob = binning_alloc(binning_ctx, gsr_binning_bytes((int)R_host));
    if (host_trace)
      fprintf(stderr, "[gsr] synchronous forward: R=%u, waited %.1f us, binning alloc %.1f us\n", R_host,
              std::chrono::duration<double, std::micro>(ta - t_enq).count(),
              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ta).count());
    if (!bblob) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL");
    key = (int)R_host;
    BinningState b = BinningState::carve(bblob, (size_t)key);
    const int rc2 = enqueue_chain(fp, g, im, b, Chain{0, Count{nullptr, key}, 0xFFFFFFFFu, 0u}, c, dord, background, out_color,
                                  out_depth, out_acc, debug, stream);
    if (rc2 != GSR_OK) return rc2;
    c.last_near = R_host;
    c.last_far = 0;
  } else {
    bool redo = false;
    uint32_t R_near = 0, R_far = 0;
    if (near_far) {
      const uint32_t budget = (uint32_t)budget64;
      const uint32_t capA = budget + (uint32_t)tiles_n;  // the run the budget falls into ends at most one rectangle later
      uint32_t capB = 0;
      const bool capB_forced = c.far_hint_override >= 0;  // (test hook: may be too small on purpose)
      if (capB_forced) {
        capB = (uint32_t)c.far_hint_override;
        c.far_hint_override = -1;
      } else if (h.have_far) {
        for (int k = 0; k < 4; k++) capB = h.recent_far[k] > capB ? h.recent_far[k] : capB;
        // (half again as much as the view's recent far chains held: an overflow bins the whole frame a second time, ~1 ms
        // at 2 M Gaussians, the margin costs 53 bytes per instance -- a camera that comes round every eighth iteration of a
        // scene that is being optimised outgrew a quarter, twice in 72 frames)
        capB = (uint32_t)std::min<unsigned long long>(0x7fffffffull - capA, (unsigned long long)capB * 3 / 2 + 131072);
        capB = std::min(round_capacity(capB), 0x7fffffffu - capA);
      } else {
        capB = hint > budget ? hint - budget : 0u;  // no history: room for every instance behind the budget
      }
      // Far-chain speculation.  In a dense scene the near chain finishes every quad, frame after frame, and the far
      // chain's launches find nothing to do (~60 us at 1080p).  After two such frames in a row (or when the test hook
      // asks) the thread's next split frame is
      //   * ASYNCHRONOUS where the device has stream-side waits (async_far_ready): the far chain goes to the library's
      //     own stream behind a wait for the near blend's decision, every kernel of it gated on "quads were left
      //     unfinished"; the caller's stream waits for the frame's go word, which the near blend's last workgroup stores
      //     itself when nothing is left to do (otherwise the far chain's last kernel does).  Nobody waits for the
      //     decision on the host; with no host in the loop the far segment is sized for every instance behind the near
      //     budget, so it cannot overflow;
      //   * otherwise decided by the host: the near blend publishes the count of unfinished quads to the mailbox and the
      //     far chain is enqueued only if there are any (one host round trip instead of eleven idle launches).
      // The near blend parks the unfinished pixels' state in the same way in every variant: same result.
      const bool async_far = speculate_far && !multi_thread && async_far_ready(c);
      const bool skip_far = speculate_far && !async_far;
      // (Measured and not kept: frames that enqueue their far chain outright sizing the far segment like the asynchronous
      // ones and returning without waiting for the far count.  Eight views per iteration from one thread: 0.985 -> 0.994 ms
      // per view -- that loop is bound by the GPU, not by the host's wait -- at 1.9 GB of binning blob per view in flight.)
      const bool lazy_cap = async_far && !capB_forced;
      const uint32_t far_expect = std::max<uint32_t>(capB, 1u << 20);  // (from history, before the capacity is widened)
      if (lazy_cap) capB = std::max(capB, hint > budget ? hint - budget : 0u);
      if (capB < 4096u) capB = 4096u;
      if ((unsigned long long)capA + capB > 0x7fffffffull) capB = 0x7fffffffu - capA;
      key = (int)(capA + capB);
      char* bblob = binning_alloc(binning_ctx, gsr_binning_bytes(key));
      if (!bblob) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL");
      BinningState b = BinningState::carve(bblob, (size_t)key);
      AsyncWords aw;
      if (async_far) {
        if (c.async_seq >= 0x3FFFFFF0u) {  // (the decision word carries 2 seq + 1)
          HIP_TRY(hipStreamSynchronize(stream));
          HIP_TRY(hipStreamSynchronize(c.far_stream));
          HIP_TRY(hipMemset(c.sig_decide, 0, 8));
          HIP_TRY(hipMemset(c.sig_go, 0, 8));
          HIP_TRY(hipMemset(c.done_counter + 2, 0, 8));
          c.async_seq = 0;
        }
        aw.decide = c.sig_decide;
        aw.go = c.sig_go;
        aw.gate_dev = reinterpret_cast<uint32_t*>(c.done_counter + 2);
        aw.seq = ++c.async_seq;
      }
      if (lazy_cap) {
        c.w_live = 16 + 2 * (int)(++c.lazy_seq & 7u);  // this frame's own outcome slot (lazy_resolve)
        c.w_far = c.w_live + 1;
      }
      int rc = enqueue_chain(fp, g, im, b, Chain{1, Count{g.total + 6, (int)capA}, budget, 0u}, c, dord, background, out_color,
                             out_depth, out_acc, debug, stream, aw);
      if (rc != GSR_OK) return rc;
      Chain far_chain{2, Count{g.total + 8, (int)capB}, 0xFFFFFFFFu, capA, far_expect};
      far_chain.cnt.bounded = lazy_cap;  // (a capacity that is rarely used: grid-stride over a bounded grid)
      uint32_t live = 0;
      c.last_far_skipped = false;
      if (async_far) {
        far_chain.cnt.gate = aw.gate_dev;
        far_chain.cnt.gate_open = 2u * aw.seq + 1u;
        // (quads left unfinished: the far chain is opened from behind the near blend's kernel boundary, render.hip)
        HIP_TRY(launch_decide_far(g.total + 13, aw, stream));
        HIP_TRY(hipStreamWaitValue32(c.far_stream, c.sig_decide, 2u * aw.seq, hipStreamWaitValueGte));
        // (not timed by the event profiler: the chain waits on its stream for the decision and then, as a rule, only
        // launches and leaves; its kernels run beside the other stream's and would be booked twice)
        t_prof_suppress = true;
        rc = enqueue_chain(fp, g, im, b, far_chain, c, dord, background, out_color, out_depth, out_acc, debug, c.far_stream);
        t_prof_suppress = false;
        if (rc != GSR_OK) return rc;
        HIP_TRY(launch_release_go(far_chain.cnt, c.sig_go, aw.seq, c.far_stream));
        HIP_TRY(hipStreamWaitValue32(stream, c.sig_go, aw.seq, hipStreamWaitValueGte));
        HIP_TRY(launch_tile_order(fp, im, stream));  // (after the go word: the far chain may move quad_last)
        order_remember(iblob);
        ++g_async_frames;
      } else if (skip_far) {
        if (launch_tile_order(fp, im, stream) != hipSuccess) return fail(GSR_ERR_HIP, "k_tile_order launch failed");
      } else {
        rc = enqueue_chain(fp, g, im, b, far_chain, c, dord, background, out_color, out_depth, out_acc, debug, stream);
        if (rc != GSR_OK) return rc;
      }
      const std::chrono::steady_clock::time_point tw = std::chrono::steady_clock::now();
      if ((rc = wait_num_rendered(c, stream, &R_host)) != GSR_OK) return rc;
      if ((rc = wait_num_rendered(c, stream, &R_near, 2)) != GSR_OK) return rc;
      bool know_far = true;
      if (lazy_cap && (unsigned long long)R_host - R_near <= (unsigned long long)capB) {
        // the far segment holds whatever the far chain may emit: nothing left for the host to check or to wait for
        know_far = false;
        lazy_push(c, c.ticket, R_near, R_host, /*speculated=*/async_far);
        R_far = 0;
      } else {
        if ((rc = wait_num_rendered(c, stream, &live, c.w_live)) != GSR_OK) return rc;
        if (skip_far && live == 0u) {
          ++g_far_skips;
          c.last_far_skipped = true;
          order_remember(iblob);
          R_far = 0;
        } else if (async_far && live == 0u) {
          R_far = 0;
          c.last_far_skipped = true;
        } else {
          if (skip_far) {  // unfinished quads after all
            ++g_far_skip_misses;
            rc = enqueue_chain(fp, g, im, b, far_chain, c, dord, background, out_color, out_depth, out_acc, debug, stream);
            if (rc != GSR_OK) return rc;
          }
          if ((rc = wait_num_rendered(c, stream, &R_far, c.w_far)) != GSR_OK) return rc;
        }
        h.far_idle_streak = live == 0u ? h.far_idle_streak + 1 : 0;
        budget_feedback(c, h, live, R_near, R_far, R_host);
      }
      if (host_trace)
        fprintf(stderr, "[gsr] near/far forward: capacity %u + %u, near %u, far %u of %u instances, %u unfinished quads%s, "
                        "enqueue %.1f us, then waited %.1f us\n", capA, capB, R_near, R_far, R_host, live,
                async_far ? (know_far ? " (asynchronous far chain, checked by the host)" : " (asynchronous far chain)")
                : skip_far ? (live ? " (far chain enqueued late)" : " (far chain not enqueued)")
                : "",
                std::chrono::duration<double, std::micro>(tw - t_enq).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count());
      ++g_speculative_forwards;
      ++g_near_far_forwards;
      if (know_far) {
        h.recent_far[h.recent_far_pos] = R_far;
        h.recent_far_pos = (h.recent_far_pos + 1) & 3;
        h.have_far = true;
      }
      c.last_was_near_far = true;
      redo = R_far > capB;
      if (!redo) frame_note(iblob, NOTE_SPLIT);
    } else {
      key = (int)hint;
      char* bblob = binning_alloc(binning_ctx, gsr_binning_bytes(key));
      if (!bblob) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL");
      BinningState b = BinningState::carve(bblob, (size_t)key);
      int rc = enqueue_chain(fp, g, im, b, Chain{0, Count{g.total, key}, 0xFFFFFFFFu, 0u}, c, dord, background, out_color,
                             out_depth, out_acc, debug, stream);
      if (rc != GSR_OK) return rc;
      const std::chrono::steady_clock::time_point tw = std::chrono::steady_clock::now();
      rc = wait_num_rendered(c, stream, &R_host);
      if (rc != GSR_OK) return rc;
      if (host_trace)
        fprintf(stderr, "[gsr] speculative forward: capacity %d, R=%u, enqueue %.1f us, then waited %.1f us\n", key,
                R_host, std::chrono::duration<double, std::micro>(tw - t_enq).count(),
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tw).count());
      ++g_speculative_forwards;
      redo = R_host > (uint32_t)key;
      R_near = R_host;
    }
    if (R_host > 0x7fffffffu) return fail(GSR_ERR_UNSUPPORTED, "more than 2^31 splat instances");
    if (redo) {  // misprediction: the clamped results are discarded and the frame is binned again, in one exact chain
      ++g_speculation_overflows;
      order_forget(iblob);
      key = (int)R_host;
      // the scans run a second time: their tile tickets, look-back status words and the near-budget marker (cleared by
      // k_preprocess for the first run) must be zero again
      const size_t nscan = ((size_t)P + SCAN_TILE - 1) / SCAN_TILE;
      HIP_TRY(hipMemsetAsync(g.dsort.tickets() + 4, 0, 2 * sizeof(uint32_t), stream));
      HIP_TRY(hipMemsetAsync(g.dsort.scan_status(), 0, sizeof(unsigned long long) * 2 * nscan, stream));  // both scans'
      HIP_TRY(hipMemsetAsync(g.total + 11, 0, sizeof(uint32_t), stream));
      char* bblob = binning_alloc(binning_ctx, gsr_binning_bytes(key));
      if (!bblob) return fail(GSR_ERR_ALLOC, "binning allocator returned NULL");
      BinningState b = BinningState::carve(bblob, (size_t)key);
      const int rc = enqueue_chain(fp, g, im, b, Chain{0, Count{nullptr, key}, 0xFFFFFFFFu, 0u}, c, dord, background, out_color,
                                   out_depth, out_acc, debug, stream);
      if (rc != GSR_OK) return rc;
      c.last_was_near_far = false;
      R_near = R_host;
      R_far = 0;
    }
    c.last_near = R_near;
    c.last_far = R_far;
  }
  if (host_trace)
    fprintf(stderr, "[gsr] forward returns at %.1f us (process clock)\n",
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
  frame_note(iblob, NOTE_PRESENT);  // (every completed forward leaves a note: a backward that finds none was evicted)
  h.recent[h.recent_pos] = R_host;  // (all instances of the frame, emitted or not: what a one-chain frame needs)
  h.recent_pos = (h.recent_pos + 1) & 3;
  c.last_R = c.last_near + c.last_far;  // the instances this forward emitted, sorted and ranged
  return key;
}

int gsr_last_num_rendered(void) {
  lazy_resolve(g_ctx);
  return (int)g_ctx.last_R;
}
int gsr_set_near_far(int on) { return near_far_flag().exchange(on != 0 ? 1 : 0); }
int gsr_set_near_far_thread(int mode) {
  const int prev = t_near_far;
  t_near_far = mode < 0 ? -1 : (mode ? 1 : 0);
  return prev;
}
int gsr_near_far(void) { return t_near_far >= 0 ? t_near_far : near_far_flag().load(); }
int gsr_last_near_far(unsigned* near_instances, unsigned* far_instances) {
  lazy_resolve(g_ctx);
  if (near_instances) *near_instances = g_ctx.last_near;
  if (far_instances) *far_instances = g_ctx.last_far;
  return g_ctx.last_was_near_far ? 1 : 0;
}
void gsr_set_near_far_hints(long long near_entries_per_tile, long long far_capacity) {
  g_ctx.near_entries_override = near_entries_per_tile < 0 ? -1 : near_entries_per_tile;
  g_ctx.far_hint_override = far_capacity < 0 ? -1 : (far_capacity > 0x7fffffffll ? 0x7fffffffll : far_capacity);
  if (far_capacity < 0)  // forget the far history as well
    for (auto& v : g_ctx.views) v.have_far = false;
}
unsigned long long gsr_near_far_forwards(void) { return g_near_far_forwards.load(); }
int gsr_set_far_speculation(int mode) {
  const int prev = g_ctx.far_skip_override;
  g_ctx.far_skip_override = mode < 0 ? -1 : (mode ? 1 : 0);
  if (mode < 0)
    for (auto& v : g_ctx.views) v.far_idle_streak = 0;
  return prev;
}
int gsr_last_far_skipped(void) {
  lazy_resolve(g_ctx);
  return g_ctx.last_far_skipped ? 1 : 0;
}
unsigned long long gsr_async_far_frames(void) { return g_async_frames.load(); }
// (the three hooks below act on the view of the calling thread's last forward)
unsigned gsr_near_budget_scale(void) {
  lazy_resolve(g_ctx);
  return g_ctx.views[g_ctx.cur].near_scale_q8;
}
unsigned gsr_near_budget_feedback(unsigned unfinished_quads, unsigned near_instances, unsigned far_instances) {
  budget_feedback(g_ctx, g_ctx.views[g_ctx.cur], unfinished_quads, near_instances, far_instances);
  return g_ctx.views[g_ctx.cur].near_scale_q8;
}
int gsr_near_far_pause(int frames) {
  const int prev = g_ctx.views[g_ctx.cur].split_pause;
  if (frames >= 0) g_ctx.views[g_ctx.cur].split_pause = frames;
  return prev;
}
unsigned long long gsr_far_skips(void) { return g_far_skips.load(); }
unsigned long long gsr_far_skip_misses(void) { return g_far_skip_misses.load(); }
long long gsr_set_binning_capacity_hint(long long capacity) {
  const long long prev = g_ctx.hint_override;
  g_ctx.hint_override = capacity < 0 ? -1 : (capacity > 0x7fffffffll ? 0x7fffffffll : capacity);
  if (capacity == 0)  // (every view's: the thread's next forward is synchronous whatever it looks at)
    for (auto& v : g_ctx.views) v.recent[0] = v.recent[1] = v.recent[2] = v.recent[3] = 0;
  return prev;
}
unsigned long long gsr_speculative_forwards(void) { return g_speculative_forwards.load(); }
unsigned long long gsr_speculation_overflows(void) { return g_speculation_overflows.load(); }
int gsr_mailbox_slow_path_last(gsr_mailbox_event* out) {
  if (!out) return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  std::lock_guard<std::mutex> lk(g_slow_mu);
  *out = g_slow_last;
  return GSR_OK;
}

int gsr_set_reference_rects(int on) { return reference_rects_flag().exchange(on != 0 ? 1 : 0); }
int gsr_reference_rects(void) { return t_reference_rects >= 0 ? t_reference_rects : reference_rects_flag().load(); }
int gsr_set_reference_rects_thread(int mode) {
  const int prev = t_reference_rects;
  t_reference_rects = mode < 0 ? -1 : (mode ? 1 : 0);
  return prev;
}
unsigned long long gsr_frame_note_misses(void) { return g_note_misses.load(); }
unsigned long long gsr_async_outcomes_lost(void) { return g_async_outcomes_lost.load(); }
int gsr_async_outcomes_pending(void) {
  lazy_resolve(g_ctx);
  return g_ctx.pending_n;
}

unsigned long long gsr_mailbox_slow_path_hits(void) { return g_mailbox_slow_hits.load(); }

int gsr_backward(int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier,
                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                 const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                 const float* dL_dacc, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot, int debug,
                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  g_err[0] = 0;
  if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P/R/width/height");
  if (P == 0) return GSR_OK;
  if (!geom_buffer || !binning_buffer || !image_buffer) return fail(GSR_ERR_INVALID_ARGUMENT, "null state blob");
  if (!means3D || !background || !viewmatrix || !projmatrix || !campos || !dL_dpix || !dL_dacc)
    return fail(GSR_ERR_INVALID_ARGUMENT, "null required input");
  // (dL_dcov3D may be NULL when the covariance is computed from scales and rotations: it is then an intermediate nobody
  // reads, and 24 of the kernel's 218 bytes per Gaussian are not written)
  if (!dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_dmean3D || (!dL_dcov3D && cov3D_precomp) ||
      !dL_dscale || !dL_drot || (M > 0 && !dL_dsh))
    return fail(GSR_ERR_INVALID_ARGUMENT, "null gradient output");
  static const bool host_trace_b = getenv("GSR_HOST_TRACE") != nullptr;
  if (host_trace_b)
    fprintf(stderr, "[gsr] backward entered at %.1f us (process clock)\n",
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count());
  const FrameParams fp = make_params(P, D, M, width, height, tan_fovx, tan_fovy, scale_modifier);
  GeomState g = GeomState::carve(geom_buffer, (size_t)P);
  BinningState b = BinningState::carve(binning_buffer, (size_t)R);
  ImageState im = ImageState::carve(image_buffer, width, height);
  if (!radii) radii = g.radii;  // rasterizer_impl.cu:386-388

  if (R > 0) {
    // inst_flag, touched and total[2] are zero here: the forward initialises them and the gather kernels
    // below clear what the blend backward sets, so the same blobs can be differentiated again.
    const unsigned notes = frame_notes(image_buffer);
    STAGE(launch_blend_backward(fp, g, b, im, background, dL_dpix, dL_dacc, (notes & NOTE_ORDER) != 0u, stream));
    STAGE(launch_gather_records(fp, g, b, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, (notes & NOTE_SPLIT) != 0u,
                                stream));
  }
  STAGE(launch_gaussian_backward(fp, g, b, radii, means3D, scales, rotations, colors_precomp ? nullptr : shs,
                                 cov3D_precomp, viewmatrix, projmatrix, campos, colors_precomp != nullptr, dL_dmean2D,
                                 dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot,
                                 stream));
  return GSR_OK;
}

int gsr_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* stream_) {
  (void)projmatrix;  // unused by the reference as well (auxiliary.h:130-135: the x/y test is commented out)
  g_err[0] = 0;
  if (P < 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P");
  if (P == 0) return GSR_OK;
  if (!means3D || !viewmatrix || !present) return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  HIP_TRY(launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_));
  return GSR_OK;
}

int gsr_activate(int P, int M, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                 const float* features_dc, const float* features_rest, float* scales, float* rotations,
                 float* opacities, float* shs, void* stream_) {
  g_err[0] = 0;
  if (P < 0 || M < 1 || M > 16) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P/M");
  if (P == 0) return GSR_OK;
  if (!scaling_raw || !rotation_raw || !opacity_raw || !features_dc || (M > 1 && !features_rest) || !scales ||
      !rotations || !opacities || !shs)
    return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  HIP_TRY(launch_activate(P, M, scaling_raw, rotation_raw, opacity_raw, features_dc, features_rest, scales, rotations,
                          opacities, shs, (hipStream_t)stream_));
  return GSR_OK;
}

int gsr_activate_backward(int P, int M, const float* rotation_raw, const float* scales, const float* opacities,
                          const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                          const float* dL_dshs, float* dL_dscaling_raw, float* dL_drotation_raw,
                          float* dL_dopacity_raw, float* dL_dfeatures_dc, float* dL_dfeatures_rest, void* stream_) {
  g_err[0] = 0;
  if (P < 0 || M < 1 || M > 16) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P/M");
  if (P == 0) return GSR_OK;
  if (!rotation_raw || !scales || !opacities || !dL_dscales || !dL_drotations || !dL_dopacities || !dL_dshs ||
      !dL_dscaling_raw || !dL_drotation_raw || !dL_dopacity_raw || !dL_dfeatures_dc || (M > 1 && !dL_dfeatures_rest))
    return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  HIP_TRY(launch_activate_backward(P, M, rotation_raw, scales, opacities, dL_dscales, dL_drotations, dL_dopacities,
                                   dL_dshs, dL_dscaling_raw, dL_drotation_raw, dL_dopacity_raw, dL_dfeatures_dc,
                                   dL_dfeatures_rest, (hipStream_t)stream_));
  return GSR_OK;
}

int gsr_adam_step(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const size_t* numel, const float* lr, double beta1, double beta2, double eps,
                  int step, int zero_grads, void* stream_) {
  g_err[0] = 0;
  if (n_tensors < 0 || n_tensors > 8) return fail(GSR_ERR_INVALID_ARGUMENT, "1..8 tensors per call");
  if (step < 1) return fail(GSR_ERR_INVALID_ARGUMENT, "step counts from 1");
  if (n_tensors == 0) return GSR_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr) return fail(GSR_ERR_INVALID_ARGUMENT, "null array");
  for (int k = 0; k < n_tensors; k++)
    if (numel[k] && (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k]))
      return fail(GSR_ERR_INVALID_ARGUMENT, "null tensor pointer");
  HIP_TRY(launch_adam(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, step, zero_grads,
                      (hipStream_t)stream_));
  return GSR_OK;
}

size_t gsr_photometric_loss_workspace(int channels, int height, int width) {
  if (channels <= 0 || height <= 0 || width <= 0) return 0;
  return loss_workspace_bytes(channels, height, width);
}

int gsr_photometric_loss(int channels, int height, int width, const float* img, const float* gt,
                         const float* window11_host, float lambda_dssim, float* loss_out3, float* dL_dimg,
                         char* workspace, size_t workspace_bytes, void* stream_) {
  g_err[0] = 0;
  if (channels <= 0 || height <= 0 || width <= 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad image shape");
  if ((unsigned long long)height * (unsigned long long)width >= 0x7fffffffull)  // (32-bit offsets inside a plane)
    return fail(GSR_ERR_INVALID_ARGUMENT, "image plane too large");
  if (!img || !gt || !window11_host || !loss_out3 || !workspace) return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  if (workspace_bytes < loss_workspace_bytes(channels, height, width))
    return fail(GSR_ERR_INVALID_ARGUMENT, "workspace too small: need %zu bytes",
                loss_workspace_bytes(channels, height, width));
  HIP_TRY(launch_photometric_loss(channels, height, width, img, gt, window11_host, lambda_dssim, loss_out3, dL_dimg,
                                  workspace, (hipStream_t)stream_));
  return GSR_OK;
}

int gsr_model_step(int P, int M, float* const* params6, float* const* exp_avg6, float* const* exp_avg_sq6,
                   const float* dL_dxyz, const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                   const float* dL_dshs, float* scales_out, float* rotations_out, float* opacities_out, float* shs_out,
                   const float* lr6, double beta1, double beta2, double eps, int step, void* stream_) {
  g_err[0] = 0;
  if (P < 0 || M < 1 || M > 16) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P/M");
  if (step < 1) return fail(GSR_ERR_INVALID_ARGUMENT, "step counts from 1");
  if (P == 0) return GSR_OK;
  if (!params6 || !exp_avg6 || !exp_avg_sq6 || !lr6 || !dL_dxyz || !dL_dscales || !dL_drotations || !dL_dopacities ||
      !dL_dshs)
    return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  for (int k = 0; k < 6; k++)
    if ((k != 2 || M > 1) && (!params6[k] || !exp_avg6[k] || !exp_avg_sq6[k]))
      return fail(GSR_ERR_INVALID_ARGUMENT, "null tensor pointer (group %d)", k);
  HIP_TRY(launch_model_step(P, M, params6, exp_avg6, exp_avg_sq6, dL_dxyz, dL_dscales, dL_drotations, dL_dopacities,
                            dL_dshs, scales_out, rotations_out, opacities_out, shs_out, lr6, beta1, beta2, eps, step,
                            (hipStream_t)stream_));
  return GSR_OK;
}

int gsr_init_gaussians(int n, int M, const float* xyz, const float* covs, const float* rgbs, float scale_factor,
                       float* xyz_out, float* features_dc_out, float* features_rest_out, float* scaling_out,
                       float* rotation_out, float* opacity_out, void* stream_) {
  g_err[0] = 0;
  if (n < 0 || M < 1 || M > 16) return fail(GSR_ERR_INVALID_ARGUMENT, "bad n/M");
  if (n == 0) return GSR_OK;
  if (!xyz || !covs || !rgbs || !xyz_out || !features_dc_out || (M > 1 && !features_rest_out) || !scaling_out ||
      !rotation_out || !opacity_out)
    return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  HIP_TRY(launch_init_gaussians(n, M, xyz, covs, rgbs, scale_factor, xyz_out, features_dc_out, features_rest_out,
                                scaling_out, rotation_out, opacity_out, (hipStream_t)stream_));
  return GSR_OK;
}

size_t gsr_ply_row_floats(int M) { return M >= 1 ? (size_t)(14 + 3 * M) : 0; }

int gsr_pack_ply_rows(int P, int M, const float* xyz, const float* features_dc, const float* features_rest,
                      const float* opacity, const float* scaling, const float* rotation, float* rows, void* stream_) {
  g_err[0] = 0;
  if (P < 0 || M < 1 || M > 16) return fail(GSR_ERR_INVALID_ARGUMENT, "bad P/M");
  if (P == 0) return GSR_OK;
  if (!xyz || !features_dc || (M > 1 && !features_rest) || !opacity || !scaling || !rotation || !rows)
    return fail(GSR_ERR_INVALID_ARGUMENT, "null pointer");
  HIP_TRY(launch_pack_ply_rows(P, M, xyz, features_dc, features_rest, opacity, scaling, rotation, rows,
                               (hipStream_t)stream_));
  return GSR_OK;
}

int gsr_geometry_view_of(char* geom_buffer, int P, gsr_geometry_view* out) {
  if (!geom_buffer || !out || P < 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad argument");
  GeomState g = GeomState::carve(geom_buffer, (size_t)P);
  out->depths = nullptr;  // not kept: the view-space depth is splats[i][9]
  out->radii = g.radii;
  out->splats = reinterpret_cast<const float*>(g.splats);
  out->cov3D = g.cov3D;
  out->tiles_touched = nullptr;  // not kept separately: gpack[i][0]
  out->gpack = reinterpret_cast<const uint32_t*>(g.gpack);
  out->point_offsets = g.point_offsets;
  out->clamped = g.clamped;
  out->depth_order = g.order;
  out->num_rendered = g.total;
  return GSR_OK;
}

int gsr_binning_view_of(char* binning_buffer, int R, gsr_binning_view* out) {
  if (!binning_buffer || !out || R < 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad argument");
  BinningState b = BinningState::carve(binning_buffer, (size_t)R);
  out->point_list = b.point_list;
  return GSR_OK;
}

int gsr_image_view_of(char* image_buffer, int width, int height, gsr_image_view* out) {
  if (!image_buffer || !out || width <= 0 || height <= 0) return fail(GSR_ERR_INVALID_ARGUMENT, "bad argument");
  ImageState im = ImageState::carve(image_buffer, width, height);
  out->ranges = reinterpret_cast<const uint32_t*>(im.ranges);
  out->final_T = im.final_T;
  out->n_contrib = im.n_contrib;
  out->quad_last = im.quad_last;
  out->ranges_far = reinterpret_cast<const uint32_t*>(im.rangesB);
  return GSR_OK;
}

}  // extern "C"
