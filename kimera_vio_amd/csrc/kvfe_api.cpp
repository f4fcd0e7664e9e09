// libkvfe C ABI (include/kvfe.h): context management, buffer layout in HBM, and the host-side
// orchestration that mirrors StereoVisionImuFrontend::processFirstStereoFrame / processStereoFrame
// (src/frontend/StereoVisionImuFrontend.cpp:245-481) by enqueuing the HIP kernels of k_*.hip on one
// stream.  No step of a frame needs the host: keyframe decisions, list compaction and landmark-id
// assignment all happen on the device, so `batch` streams advance in lock-step per launch.
// There is NO CPU fallback: without a gfx950 device kvfe_create fails with KVFE_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/kvfe.h"
#include "host_calib.hpp"
#include "kvfe_dev.hpp"

using namespace kvfe;

namespace kvfe {
int warm_dma_engines(int device_ordinal, void* dev_buf, void* host_buf, size_t bytes);   // host_dma_warm.cpp
}
namespace {

constexpr int ACAP = 8192;  // accepted-corner capacity (LDS sort capacity of the select kernel)
constexpr int PROF_FLAG_SAMPLES = 512;  // sampled steps whose stream flags are kept between two profile reads
constexpr int MIN_GROUP_STREAMS = 8;  // automatic stream groups hold at least this many streams

enum Stage {
  ST_PYRAMID = 0,
  ST_TRACK,
  ST_TRACK_FINALIZE,
  ST_MINEIG,
  ST_SELECT,
  ST_SUBPIX,
  ST_RECTIFY,
  ST_STEREO,
  ST_STEREO_NEW,
  ST_FINALIZE,
  ST_RANSAC_MONO,
  ST_RANSAC_STEREO,
  ST_OUT_PACK,       // round 6: the step's output records gathered on the tail's stream ...
  ST_OUT_TRANSFER,   // ... and their transfer to the pinned ring slot on the output stream (many streams)
  ST_COUNT
};
static_assert(ST_COUNT <= KVFE_N_STAGES, "kvfe_stage_times holds KVFE_N_STAGES stages");
const char* kStageNames[ST_COUNT] = {"pyramid",  "lk_track", "track_finalize", "mineig_localmax",
                                     "gftt_select", "subpix_append", "rectify", "stereo_match",
                                     "stereo_match_new", "step_finalize", "ransac_mono",
                                     "ransac_stereo", "out_pack", "out_transfer"};

struct Buffers {  // everything that scales with the number of streams
  unsigned char* lvl0[2] = {nullptr, nullptr};  // [B][H][W] own copy of the left image per pyramid slot (device-pointer steps)
  int B = 0;
  unsigned char* rect[2] = {nullptr, nullptr};
  unsigned char* pyr[2] = {nullptr, nullptr};
  // ctx-owned image slots (host-input / staged / equalised paths): a left image is read by two
  // consecutive steps (LK re-reads frame k-1), a right image by one -> three / two slots let the
  // upload of frame k+1 overlap the kernels of frame k
  unsigned char* raw_left[3] = {nullptr, nullptr, nullptr};
  unsigned char* raw_right2[2] = {nullptr, nullptr};
  unsigned char*& raw_right = raw_right2[0];
  unsigned char* eq_in[2] = {nullptr, nullptr};  // staged upload target when equalizeImage is on
  int* eq_hist = nullptr;                        // [2][B][256]
  unsigned char* user_mask = nullptr;
  unsigned char* depth_slot[2] = {nullptr, nullptr};  // RGBD: ctx-owned depth images of the host-input path
  unsigned char* depth_mask = nullptr;                 // RGBD: DepthFrame::getDetectionMask, [B][H][W]
  FrameTab ft[3];
  StereoTab st;
  StereoTab lst;  // stereo tables of the last keyframe (geometric outlier rejection reads them)
  RansacScratch rs;
  StreamState ss = {};
  DetectScratch ds;
  LkScratch lk;
  double* kf_R_cur = nullptr;
  long long* in_ts = nullptr;
  int* in_force = nullptr;
};

}  // namespace

struct kvfe_ctx {
  kvfe_config cfg;
  KParams P;
  Tables T;
  kvfe_rectification rect;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<void*> allocs;
  std::vector<void*> host_allocs;
  Buffers fe;    // front-end (batch streams)
  Buffers comp;  // component API (one stream), allocated lazily
  KParams Pc;    // KParams with B = 1 for the component API
  bool comp_ready = false;
  int role_k = 0, role_km1 = 1, role_lkf = 2;
  int pyr_cur = 0;
  bool own_level0 = false;   // this step's left image is copied into fe.lvl0 for the next step's LK
  const unsigned char* prev_left = nullptr;
  size_t prev_row_stride = 0, prev_img_stride = 0;
  int raw_slot = 0;
  long long img_step = 0;  // steps that went through the ctx-owned image slots
  // staged input (kvfe_frontend_step_staged)
  unsigned char* stage_host[KVFE_STAGING_SLOTS] = {};
  hipEvent_t stage_copied[KVFE_STAGING_SLOTS] = {};
  bool stage_pending[KVFE_STAGING_SLOTS] = {};
  hipStream_t copy_stream = nullptr;
  hipEvent_t step_done[4] = {};
  bool step_done_valid[4] = {};
  bool last_step_staged = false;
  int pts_bound = 0;
  // pinned input staging ring
  static constexpr int RING = 64;
  // staged steps: the per-stream inputs travel to a device ring slot by a copy behind the frames' upload (do_step)
  unsigned char* in_dev = nullptr;                   // [RING][ring_bytes]; set when ALL staging resources exist
  unsigned char* in_dev_staging = nullptr;           // (its allocation)
  bool staging_ready = false;                        // ensure_staging completed: copy stream, step_done / ev_chain events, input ring
  hipEvent_t in_ev[RING] = {};
  bool inputs_by_copy_call = false;                  // this do_step call: set by kvfe_frontend_step_staged
  unsigned char* ring_host[RING] = {};
  hipEvent_t ring_ev[RING] = {};
  bool ring_used[RING] = {};
  int ring_pos = 0;
  size_t ring_bytes = 0, ring_bytes_dev = 0;
  // device-side tables
  UndistortDev und[2][4];  // [cam][useR + 2*useP]
  std::vector<float> h_map[2][2];  // host copies of the maps [cam][x|y]
  // profiling
  bool prof_on = false;      // this step records stage events
  int prof_stride = 0;       // 0 = profiling off; N = every N-th step records
  long long prof_step = 0;
  std::vector<hipEvent_t> prof_ev;  // events of the recorded steps; a recorded step owns 2 * ST_COUNT INDICES into it
  std::vector<int> prof_idx;        // [sample][2 * ST_COUNT] begin / end event of a stage (-1: stage not run)
  int prof_last_end = -1;           // the event that ended the previous stage of this step ...
  hipStream_t prof_last_stream = nullptr;   // ... and the stream it was recorded on
  // corner refinement runs on a side stream, concurrently with rectification and the stereo
  // matching of the tracked keypoints (its result is only needed by the newly detected ones)
  hipStream_t side = nullptr;
  hipEvent_t ev_chain[4] = {};                       // "the side stream's rectify / match / reject chain of step i is done", i mod 4
  long long chain_seq = 0;                           // steps whose chain went to the side stream

  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mono = nullptr;
  hipEvent_t ev_main = nullptr, ev_tail = nullptr;   // main-stream part of the fork done / tail of the step (side stream) done
  hipEvent_t ev_commit = nullptr;                    // this step's new corners are in the frame table (side stream)
  bool commit_pending = false;                       // the next step's tracking has not been ordered after ev_commit yet
  bool fork_swap = false;                            // few streams: the corner refinement stays on the main stream (do_step)
  // Round 6, a few streams through kvfe_frontend_step_host (the synchronous single-robot use, shim::spinOnce): the host
  // reads the step's flags behind the keyframe decision and does not launch the keyframe-only kernels of a step in which
  // no stream is a keyframe -- at the reference cadence four steps of five, each ~95 us of a dozen kernels that return at
  // their first flag test
  int* quiet_flags_host = nullptr;                   // [B] pinned + mapped: track_finalize publishes flags | seq << 8 (StreamState)
  int* quiet_flags_dev = nullptr;                    // (its device address)
  int quiet_seq = 0;
  bool quiet_check_call = false;
  // kvfe_frontend_step_host, a few streams: the RIGHT image comes up on its own stream -- the pyramid and the tracking
  // launch only read the left one, and nothing reads the right one before the rectification of a keyframe
  hipStream_t up2_stream = nullptr;
  hipEvent_t ev_up2 = nullptr;
  bool right_pending = false;                        // this do_step call: the right image's upload ends with ev_up2
  bool serial_call = false;                          // this do_step call keeps every kernel on the main stream (see do_step)
  bool frames_persist_call = false;                  // this do_step call reads caller frames that stay valid for one more step
  bool chain_pending = false;                        // fork_swap: the side stream's outlier rejection has not been joined yet
  bool tail_pending = false;                          // the last step's tail has not been joined into the main stream yet
  std::vector<int> prof_pending;
  double prof_ms[ST_COUNT] = {};
  int prof_samples = 0;
  // per-stage activity of the sampled steps: the keyframe-only stages are launched every step but work only for the
  // streams whose flags say so (read back per sampled step from the device)
  double prof_ms_active[ST_COUNT] = {};
  double prof_active_streams[ST_COUNT] = {};
  int prof_active_launches[ST_COUNT] = {};
  int* prof_flags_host = nullptr;     // pinned [PROF_FLAG_SAMPLES][B]
  std::vector<int> prof_flag_slot;    // per pending sample: its slot in prof_flags_host (-1: none)
  int prof_flag_next = 0;
  std::string last_error;
  // output side (kvfe_dev.hpp "output side"): OUT_RING pinned host slots of B packed records; with many streams a
  // device staging buffer per slot + one DMA transfer on `out_stream`, with a few (out_direct) the pack kernel writes
  // the mapped slot itself
  unsigned char* out_host[OUT_RING] = {};
  unsigned char* out_host_dev[OUT_RING] = {};   // device address of the mapped slot
  unsigned char* out_stage[OUT_RING] = {};
  hipEvent_t ev_packed[OUT_RING] = {}, ev_out[OUT_RING] = {};
  hipStream_t out_stream = nullptr;
  size_t out_tab_bytes = 0, out_slot_size = 0;   // offset table in front of a slot's records / bytes of a slot
  size_t out_copied[OUT_RING] = {};              // bytes of the slot the step's transfer covers
  size_t out_guess = 0;                          // bytes the next transfer covers: the last known need + a margin
  hipStream_t topup_stream = nullptr;            // locate_output: the rest of a record set that outgrew its transfer
  size_t out_guess_forced = 0;                   // KVFE_OUT_TRANSFER_BYTES (debugging aid: exercises the top-up path)
  int out_cap = 0;           // entries a record has room for: what a frame table can hold (pts_bound), not what it is allocated for
  bool out_direct = false;
  long long out_steps = 0;   // steps whose record has been enqueued; slot of step i = i % OUT_RING
  long long out_topups = 0;  // accesses that had to fetch the rest of a step's records (locate_output)
  // dense stereo (allocated on first use, re-allocated when the volume geometry changes)
  DenseBuffers dense;
  std::vector<void*> dense_allocs;
  float* dense_dispf = nullptr;   // reprojectImageTo3D input / output staging
  float* dense_xyz = nullptr;
  unsigned* dense_minkey = nullptr;
  hipEvent_t dense_ev[2] = {};
  double dense_ms = 0;            // kernel time of kvfe_dense_stereo_reconstruction calls (HIP events)
  long long dense_pairs = 0;
  long long dense_fallbacks = 0;  // chunks repeated on the direction sweeps after a two-pass hand-over wait ran out
  // stream groups: a context with batch >= 2*MIN_GROUP_STREAMS splits its streams into `groups`
  // child contexts (own HIP stream, own buffers, shared constant tables).  The children free-run;
  // the only coupling is the token below that staggers their phases so that the latency-bound
  // kernels of one group (select, cornerSubPix tails) overlap the throughput-bound kernels (LK,
  // min-eig, SSD) of the others.
  std::vector<kvfe_ctx*> children;
  kvfe_ctx* parent = nullptr;
  int s0 = 0;                       // first stream of this child in the parent's batch
  hipEvent_t ev_tracked = nullptr;  // recorded after this group's track_finalize launch
  bool ev_tracked_valid = false;
};

namespace {

#define HIPCHK(ctx, expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      return KVFE_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

// Every public entry point runs on the context's device whatever the calling thread's current device is
// (distinct contexts may live on different GPUs / threads, kvfe.h), and leaves the caller's device as it was.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const kvfe_ctx* c) {
    if (c) enter(c->cfg.device);
  }
  explicit DeviceGuard(int device) { enter(device); }
  void enter(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) hipSetDevice(prev);
  }
};

// The last step's tail runs on the side stream (do_step); every entry point that reads or resets the front-end state
// through the main stream joins it first, so that "main stream complete" means "step complete" for the caller.
inline void join_tail(kvfe_ctx* c) {
  if (c && c->tail_pending && c->stream) {
    hipStreamWaitEvent(c->stream, c->ev_tail, 0);
    c->tail_pending = false;
    c->commit_pending = false;   // (the tail follows the commit on the side stream)
    c->chain_pending = false;    // (... and the side stream's part of the fork)
  }
  if (c)
    for (kvfe_ctx* ch : c->children) join_tail(ch);
}

// ---------------------------------------------------------------------------------------------
// KVFE_GUARD_ALLOC = 1 | 2 (debugging aid, round 6; GPU AddressSanitizer is not available on every pool): every device
// buffer of the library becomes its own mapping between two UNMAPPED guard ranges (HIP virtual memory management:
// hipMemAddressReserve + hipMemCreate + hipMemMap), placed so that the buffer ENDS where the mapping ends (1: a read or
// write 16 bytes or more past the end of any buffer faults) or BEGINS where it begins (2: any access in front of a
// buffer faults).  An out-of-bounds access that lands in a neighbouring allocation is silent with hipMalloc -- round 6's
// fuzz finding took 66 configurations of allocation history to become a fault -- and deterministic here:
// `KVFE_GUARD_ALLOC=1 python -m pytest tests -m gpu` and the same with 2 (tools/r6/gpu_guard.sh).  Results do not
// change; allocation is slower and every buffer costs at least one granule (2 MB on MI355X).
// ---------------------------------------------------------------------------------------------
struct GuardRec { void* base; size_t reserve; void* mapped; size_t mapped_bytes; hipMemGenericAllocationHandle_t handle; };
std::mutex g_guard_mu;
std::unordered_map<void*, GuardRec> g_guard;
int guard_mode() {
  static const int m = [] { const char* e = std::getenv("KVFE_GUARD_ALLOC"); return e ? std::atoi(e) : 0; }();
  return m;
}
hipError_t dev_malloc(void** out, size_t bytes) {
  const int mode = guard_mode();
  if (mode == 3) {   // plain allocations filled with a poison byte (KVFE_POISON, default 0xff): a read of memory no kernel of the
    hipError_t e3 = hipMalloc(out, bytes);   // call has written shows up as a result that depends on it
    if (e3 != hipSuccess) return e3;
    static const int poison = [] { const char* e = std::getenv("KVFE_POISON"); return e ? (int)std::strtol(e, nullptr, 0) : 0xff; }();
    if ((e3 = hipMemset(*out, poison, bytes)) != hipSuccess) return e3;
    return hipDeviceSynchronize();   // (the fill runs on the null stream; the library's streams are non-blocking and do not wait for it)
  }
  if (mode != 1 && mode != 2) return hipMalloc(out, bytes);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t g = 0;
  if ((e = hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  if (g == 0) g = 2u << 20;
  const size_t need = std::max<size_t>((bytes + 15) & ~(size_t)15, 16);
  const size_t mapped = (need + g - 1) / g * g;
  GuardRec r = {};
  r.reserve = mapped + 2 * g;
  if ((e = hipMemAddressReserve(&r.base, r.reserve, g, nullptr, 0)) != hipSuccess) return e;
  if ((e = hipMemCreate(&r.handle, mapped, &prop, 0)) != hipSuccess) {
    (void)hipMemAddressFree(r.base, r.reserve);
    return e;
  }
  r.mapped = static_cast<char*>(r.base) + g;
  r.mapped_bytes = mapped;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemMap(r.mapped, mapped, 0, r.handle, 0)) != hipSuccess ||
      (e = hipMemSetAccess(r.mapped, mapped, &acc, 1)) != hipSuccess) {
    (void)hipMemRelease(r.handle);
    (void)hipMemAddressFree(r.base, r.reserve);
    return e;
  }
  void* user = mode == 1 ? static_cast<char*>(r.mapped) + (mapped - need) : r.mapped;
  static const bool log_allocs = std::getenv("KVFE_GUARD_ALLOC_LOG") != nullptr;   // (which buffer a fault address belongs to)
  if (log_allocs)
    std::fprintf(stderr, "KVFE_GUARD_ALLOC: %zu bytes at %p .. %p, mapping %p .. %p\n", bytes, user,
                 static_cast<void*>(static_cast<char*>(user) + bytes), r.mapped, static_cast<void*>(static_cast<char*>(r.mapped) + mapped));
  {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[user] = r;
  }
  *out = user;
  return hipSuccess;
}
void dev_free(void* p) {
  if (!p) return;
  GuardRec r = {};
  bool guarded = false;
  {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    auto it = g_guard.find(p);
    if (it != g_guard.end()) {
      r = it->second;
      g_guard.erase(it);
      guarded = true;
    }
  }
  if (!guarded) {
    (void)hipFree(p);
    return;
  }
  (void)hipDeviceSynchronize();   // (hipFree's implicit synchronisation)
  if (std::getenv("KVFE_GUARD_ALLOC_LOG")) std::fprintf(stderr, "KVFE_GUARD_ALLOC: free %p (mapping %p)\n", p, r.mapped);
  (void)hipMemUnmap(r.mapped, r.mapped_bytes);
  (void)hipMemRelease(r.handle);
  // The address range stays reserved (KVFE_GUARD_VA_REUSE=1 frees it): a later reservation that landed on a range this
  // process had unmapped before saw rows of a freshly written buffer hold another buffer's values (tools/r6/
  // guard_dense_probe.py: the labels of cv::filterSpeckles after seven calls that re-allocate the dense buffers) -- a stale
  // translation, not a kernel of the library: plain allocations filled with 0xff / 0xa5 / 0 (KVFE_GUARD_ALLOC=3) run the
  // same sequence to the same results.  A used-once address also keeps a dangling pointer of the library faulting.
  static const bool va_reuse = [] { const char* e = std::getenv("KVFE_GUARD_VA_REUSE"); return e && std::atoi(e) != 0; }();
  if (va_reuse) (void)hipMemAddressFree(r.base, r.reserve);
}

template <typename T>
kvfe_status dalloc(kvfe_ctx* c, T** p, size_t n, bool zero = true) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  HIPCHK(c, dev_malloc(&q, bytes));
  c->allocs.push_back(q);
  if (zero) HIPCHK(c, hipMemsetAsync(q, 0, bytes, c->stream));
  *p = reinterpret_cast<T*>(q);
  return KVFE_OK;
}

#define TRY(expr)                      \
  do {                                 \
    kvfe_status _s = (expr);           \
    if (_s != KVFE_OK) return _s;      \
  } while (0)

// TrackerStatusSummary(): both statuses INVALID, identity poses, zero information
kvfe_status reset_tracker_status(kvfe_ctx* c, Buffers& b) {
  const size_t B = b.B;
  std::vector<int> st(B * 2, TRK_INVALID);
  std::vector<double> pose(B * 24, 0.0);
  for (size_t i = 0; i < B * 2; i++) pose[i * 12] = pose[i * 12 + 5] = pose[i * 12 + 10] = 1.0;
  HIPCHK(c, hipMemcpyAsync(b.ss.trk_status, st.data(), sizeof(int) * st.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b.ss.trk_pose, pose.data(), sizeof(double) * pose.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(b.ss.trk_info, 0, sizeof(double) * 9 * B, c->stream));
  HIPCHK(c, hipMemsetAsync(b.ss.trk_counts, 0, sizeof(int) * 6 * B, c->stream));
  HIPCHK(c, hipMemcpyAsync(b.ss.pnp_status, st.data(), sizeof(int) * B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b.ss.pnp_pose, pose.data(), sizeof(double) * 12 * B, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(b.ss.pnp_counts, 0, sizeof(int) * 3 * B, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  return KVFE_OK;
}

kvfe_status alloc_buffers(kvfe_ctx* c, Buffers& b, const KParams& P) {
  b.B = P.B;
  const size_t N = (size_t)P.W * P.H, B = P.B, K = (size_t)P.kcap * B;
  for (int i = 0; i < 2; i++) {
    TRY(dalloc(c, &b.rect[i], N * B));
    TRY(dalloc(c, &b.pyr[i], (size_t)P.pyr_stride * B));
    TRY(dalloc(c, &b.raw_left[i], N * B));
    TRY(dalloc(c, &b.raw_right2[i], N * B));
  }
  TRY(dalloc(c, &b.raw_left[2], N * B));
  TRY(dalloc(c, &b.eq_hist, 2 * 256 * B));
  if (P.rgbd) {
    const size_t bpp = P.depth_f32 ? 4 : 2;
    for (int i = 0; i < 2; i++) TRY(dalloc(c, &b.depth_slot[i], N * B * bpp));
    TRY(dalloc(c, &b.depth_mask, N * B));
  }
  for (int i = 0; i < 3; i++) {
    TRY(dalloc(c, &b.ft[i].kp, K));
    TRY(dalloc(c, &b.ft[i].lmk, K));
    TRY(dalloc(c, &b.ft[i].age, K));
    TRY(dalloc(c, &b.ft[i].versor, K * 3));
    TRY(dalloc(c, &b.ft[i].count, B));
    TRY(dalloc(c, &b.ft[i].timestamp, B));
  }
  TRY(dalloc(c, &b.st.left_rect, K));
  TRY(dalloc(c, &b.st.left_status, K));
  TRY(dalloc(c, &b.st.right_rect, K));
  TRY(dalloc(c, &b.st.right_status, K));
  TRY(dalloc(c, &b.st.depth, K));
  TRY(dalloc(c, &b.st.right_kp, K));
  TRY(dalloc(c, &b.st.kp3d, K * 3));
  std::memset(&b.lst, 0, sizeof(b.lst));
  TRY(dalloc(c, &b.lst.left_rect, K));
  TRY(dalloc(c, &b.lst.right_rect, K));
  TRY(dalloc(c, &b.lst.right_status, K));
  TRY(dalloc(c, &b.lst.kp3d, K * 3));
  TRY(dalloc(c, &b.rs.matches, K));
  TRY(dalloc(c, &b.rs.f_ref, K * 3));
  TRY(dalloc(c, &b.rs.f_cur, K * 3));
  TRY(dalloc(c, &b.rs.relc, K * 9));
  TRY(dalloc(c, &b.rs.votef, K * 12));
  TRY(dalloc(c, &b.rs.acc, K * 12));
  TRY(dalloc(c, &b.rs.inliers, K));
  TRY(dalloc(c, &b.rs.n_inliers, B));
  TRY(dalloc(c, &b.rs.cnt, K));
  TRY(dalloc(c, &b.rs.n_matches, B));
  TRY(dalloc(c, &b.ss.trk_status, B * 2));
  TRY(dalloc(c, &b.ss.trk_pose, B * 24));
  TRY(dalloc(c, &b.ss.trk_info, B * 9));
  TRY(dalloc(c, &b.ss.trk_counts, B * 6));
  TRY(dalloc(c, &b.ss.pnp_status, B));
  TRY(dalloc(c, &b.ss.pnp_counts, B * 3));
  TRY(dalloc(c, &b.ss.pnp_pose, B * 12));
  TRY(dalloc(c, &b.ss.map_ids, B * (size_t)P.map_cap));
  TRY(dalloc(c, &b.ss.map_xyz, B * (size_t)P.map_cap * 3));
  TRY(dalloc(c, &b.ss.map_n, B));
  TRY(dalloc(c, &b.ss.flags, B));
  TRY(dalloc(c, &b.ss.n_tracked, B));
  TRY(dalloc(c, &b.ss.n_detected, B));
  TRY(dalloc(c, &b.ss.n_meas, B));
  TRY(dalloc(c, &b.ss.lmk_counter, B));
  TRY(dalloc(c, &b.ss.frame_count, B));
  TRY(dalloc(c, &b.ss.kf_R_ref, B * 9));
  {  // per-step inputs live in one block so that a step needs a single H2D copy:
     // [B][9] f64 keyframe_R_cur_frame | [B] i64 timestamp | [B] i32 force_keyframe
    unsigned char* blk;
    TRY(dalloc(c, &blk, (sizeof(double) * 9 + sizeof(long long) + sizeof(int)) * B));
    b.kf_R_cur = reinterpret_cast<double*>(blk);
    b.in_ts = reinterpret_cast<long long*>(blk + sizeof(double) * 9 * B);
    b.in_force = reinterpret_cast<int*>(blk + (sizeof(double) * 9 + sizeof(long long)) * B);
  }
  b.ss.kf_R_cur = b.kf_R_cur;
  b.ss.in_timestamp = b.in_ts;
  b.ss.in_force_kf = b.in_force;
  TRY(dalloc(c, &b.ss.meas_lmk, K));
  TRY(dalloc(c, &b.ss.meas_uLuRv, K * 3));
  int sort_cap = 1;
  while (sort_cap < P.ccap) sort_cap <<= 1;
  b.ds.sort_cap = sort_cap;
  TRY(dalloc(c, &b.ds.cand, (size_t)P.ccap * B));
  TRY(dalloc(c, &b.ds.cand_count, B));
  TRY(dalloc(c, &b.ds.maxkey, B));
  TRY(dalloc(c, &b.ds.corners, (size_t)P.acap * B));
  TRY(dalloc(c, &b.ds.n_corners, B));
  TRY(dalloc(c, &b.ds.newc, (size_t)P.acap * B));
  TRY(dalloc(c, &b.ds.n_new, B));
  TRY(dalloc(c, &b.ds.sp_next, B));
  TRY(dalloc(c, &b.ds.need, B));
  TRY(dalloc(c, &b.ds.cell_items, (size_t)P.ccap * B));
  TRY(dalloc(c, &b.ds.state, (size_t)P.ccap * B));
  TRY(dalloc(c, &b.ds.sortbuf, (size_t)sort_cap * B));
  if (P.detector == 0)   // FeatureDetectorType::FAST: the detection mask as a bitmap
    TRY(dalloc(c, &b.ds.me_maskbits, (size_t)P.H * me_mask_words(P.W) * B));
  else
    b.ds.me_maskbits = nullptr;
  TRY(dalloc(c, &b.lk.prev_pts, K));
  TRY(dalloc(c, &b.lk.next_pts, K));
  TRY(dalloc(c, &b.lk.status, K));
  TRY(dalloc(c, &b.lk.err, K));
  TRY(dalloc(c, &b.lk.npts, B));
  b.lk.skip_age = nullptr;   // (set per launch by the front-end step)
  TRY(dalloc(c, &b.lk.src_idx, K));
  TRY(reset_tracker_status(c, b));
  // keyframe_R_ref_frame_ = identity
  std::vector<double> eye(B * 9, 0.0);
  for (size_t s = 0; s < B; s++) eye[s * 9] = eye[s * 9 + 4] = eye[s * 9 + 8] = 1.0;
  HIPCHK(c, hipMemcpyAsync(b.ss.kf_R_ref, eye.data(), sizeof(double) * B * 9,
                           hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KVFE_OK;
}

UndistortDev to_dev(const UndistortCtx& u) {
  UndistortDev d;
  d.fx = u.fx;
  d.fy = u.fy;
  d.cx = u.cx;
  d.cy = u.cy;
  d.ifx = u.ifx;
  d.ify = u.ify;
  for (int i = 0; i < 8; i++) d.k[i] = u.k[i];
  for (int i = 0; i < 9; i++) d.RR[i] = u.RR.m[i];
  d.has_dist = u.has_dist;
  d.pad = 0;
  return d;
}

void subpix_mask_table(int win, int zero_zone, std::vector<float>& m) {
  // cv::cornerSubPix weight mask (float exp from the host libm, as the reference computes it)
  const int ww = 2 * win + 1;
  m.resize((size_t)ww * ww);
  for (int i = 0; i < ww; i++) {
    float y = (float)(i - win) / win;
    float vy = std::exp(-y * y);
    for (int j = 0; j < ww; j++) {
      float x = (float)(j - win) / win;
      m[i * ww + j] = (float)(vy * std::exp(-x * x));
    }
  }
  if (zero_zone >= 0 && zero_zone * 2 + 1 < ww) {
    for (int i = win - zero_zone; i <= win + zero_zone; i++)
      for (int j = win - zero_zone; j <= win + zero_zone; j++) m[i * ww + j] = 0;
  }
}

// std::mt19937 (ISO C++ [rand.eng.mers]) + libstdc++'s uniform_int_distribution<int>(0, INT_MAX):
// KVFE_RNG_LIBSTDCXX_PRE11 redraws while the output is >= 2^31, KVFE_RNG_LIBSTDCXX_11 shifts it right
void ransac_rnd_stream(int policy, int n, int* out) {
  uint32_t mt[624];
  mt[0] = 12345u;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
  int idx = 624;
  auto next = [&]() {
    if (idx >= 624) {
      for (int k = 0; k < 624; k++) {
        const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  };
  for (int i = 0; i < n; i++) {
    uint32_t r = next();
    if (policy == KVFE_RNG_LIBSTDCXX_11) {
      out[i] = (int)(r >> 1);
    } else {
      while (r >= 0x80000000u) r = next();
      out[i] = (int)r;
    }
  }
}

void matx33f_inv(const float* a, float* b) {  // cv::Matx33f::inv() (direct formula, float)
  auto A = [&](int r, int c) { return a[r * 3 + c]; };
  float d = A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) -
            A(0, 1) * (A(1, 0) * A(2, 2) - A(2, 0) * A(1, 2)) +
            A(0, 2) * (A(1, 0) * A(2, 1) - A(2, 0) * A(1, 1));
  d = 1 / d;
  b[0] = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) * d;
  b[1] = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * d;
  b[2] = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * d;
  b[3] = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) * d;
  b[4] = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * d;
  b[5] = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * d;
  b[6] = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) * d;
  b[7] = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * d;
  b[8] = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * d;
}

kvfe_status validate(const kvfe_config* cfg, std::string* why) {
  const kvfe_frontend_params& p = cfg->params;
  auto fail = [&](const char* m, kvfe_status s) {
    *why = m;
    return s;
  };
  if (cfg->batch < 1) return fail("batch must be >= 1", KVFE_ERR_INVALID_ARG);
  if (cfg->frontend_type != KVFE_FRONTEND_STEREO && cfg->frontend_type != KVFE_FRONTEND_MONO &&
      cfg->frontend_type != KVFE_FRONTEND_RGBD)
    return fail("unknown frontend_type", KVFE_ERR_INVALID_ARG);
  const bool rgbd = cfg->frontend_type == KVFE_FRONTEND_RGBD;
  const bool mono = cfg->frontend_type == KVFE_FRONTEND_MONO || rgbd;   // one camera, R = I, P = K
  if (rgbd) {
    const kvfe_depth_params& d = cfg->depth;
    if (d.depth_type != KVFE_DEPTH_U16 && d.depth_type != KVFE_DEPTH_F32)
      return fail("bad depth_type", KVFE_ERR_INVALID_ARG);
    if (!(d.virtual_baseline > 0.f)) return fail("virtual_baseline must be positive", KVFE_ERR_INVALID_ARG);  // CameraParams.cpp:344
    if (!(d.depth_to_meters > 0.f)) return fail("depth_to_meters must be positive", KVFE_ERR_INVALID_ARG);
    if (!d.is_registered)
      return fail("unregistered depth images need cv::rgbd::registerDepth, which is not implemented",
                  KVFE_ERR_UNSUPPORTED);
    if (p.stereo.equalize_image) return fail("equalize_image with the RGBD front-end", KVFE_ERR_UNSUPPORTED);
    if (cfg->stream_groups > 1) return fail("stream_groups > 1 with the RGBD front-end", KVFE_ERR_UNSUPPORTED);
  }
  if (!mono && (cfg->left.width != cfg->right.width || cfg->left.height != cfg->right.height))
    return fail("left/right image sizes differ", KVFE_ERR_INVALID_ARG);
  if (cfg->left.width < 16 || cfg->left.height < 16) return fail("image too small", KVFE_ERR_INVALID_ARG);
  if (p.stereo.ssd_tie_policy != KVFE_SSD_TIE_EXACT && p.stereo.ssd_tie_policy != KVFE_SSD_TIE_F32)
    return fail("bad ssd_tie_policy", KVFE_ERR_INVALID_ARG);
  for (int v : {cfg->device_frames_persist, cfg->single_hip_stream, cfg->copy_inputs, cfg->ssd_impl, cfg->lk_impl})
    if (v != 0 && v != 1) return fail("execution options of kvfe_config are 0 or 1", KVFE_ERR_INVALID_ARG);
  if (p.use_ransac) {
    const kvfe_tracker_params& tr = p.tracker;
    if (tr.ransac_randomize)
      return fail("ransac_randomize=1 (time-seeded sampling) is not reproducible: set 0", KVFE_ERR_UNSUPPORTED);
    if (!tr.ransac_use_2point_mono && tr.pose_2d2d_algorithm != 1)
      return fail("2d2d_algorithm: only NISTER (1) is implemented for ransac_use_2point_mono=0",
                  KVFE_ERR_UNSUPPORTED);
    if (p.use_pnp_tracking && (!mono || rgbd)) {   // (MonoVisionImuFrontend has no PnP stage)
      if (p.pnp.pnp_algorithm != 3 && p.pnp.pnp_algorithm != 1)
        return fail("use_pnp_tracking: pnp_algorithm must be 3 (EPNP) or 1 (KneipP3P)", KVFE_ERR_UNSUPPORTED);
      if (p.pnp.optimize_2d3d_pose_from_inliers)
        return fail("optimize_2d3d_pose_from_inliers is not implemented", KVFE_ERR_UNSUPPORTED);
      if (!(p.pnp.ransac_threshold_pnp > 0.0)) return fail("ransac_threshold_pnp must be > 0", KVFE_ERR_INVALID_ARG);
    }
    if (tr.ransac_max_iterations < 1 || tr.ransac_max_iterations > 1000)
      return fail("ransac_max_iterations out of range [1,1000]", KVFE_ERR_INVALID_ARG);
    if (!(tr.ransac_probability > 0.0 && tr.ransac_probability < 1.0))
      return fail("ransac_probability must be in (0,1)", KVFE_ERR_INVALID_ARG);
    if (tr.ransac_rng_policy != KVFE_RNG_LIBSTDCXX_PRE11 && tr.ransac_rng_policy != KVFE_RNG_LIBSTDCXX_11)
      return fail("bad ransac_rng_policy", KVFE_ERR_INVALID_ARG);
  }
  const kvfe_detector_params& d = p.detector;
  if (d.feature_detector_type != KVFE_DET_GFTT && d.feature_detector_type != KVFE_DET_FAST)
    return fail("feature_detector_type: GFTT and FAST are implemented (ORB is not; AGAST is LOG(FATAL) upstream, "
                "FeatureDetector.cpp:67-70)", KVFE_ERR_UNSUPPORTED);
  if (d.feature_detector_type == KVFE_DET_FAST && (cfg->left.width < 7 || cfg->left.height < 7))
    return fail("FAST needs an image of at least 7 x 7 pixels", KVFE_ERR_INVALID_ARG);
  if (d.use_harris_detector && (long long)cfg->left.width * cfg->left.height < 4)
    return fail("use_harris_detector needs an image of at least 4 pixels", KVFE_ERR_INVALID_ARG);
  if (d.block_size != 3) return fail("block_size must be 3", KVFE_ERR_UNSUPPORTED);
  if (d.enable_non_max_suppression &&
      (d.non_max_suppression_type < KVFE_ANMS_TOPN || d.non_max_suppression_type > KVFE_ANMS_BINNING))
    return fail("unknown non_max_suppression_type", KVFE_ERR_INVALID_ARG);
  if (d.min_distance < 0 || d.min_distance > MAX_RADIUS)
    return fail("min_distance out of range [0,127]", KVFE_ERR_INVALID_ARG);
  if (d.max_nr_keypoints_before_anms < 0)
    return fail("max_nr_keypoints_before_anms < 0", KVFE_ERR_INVALID_ARG);
  if (d.max_nr_keypoints_before_anms > ACAP)
    return fail("max_nr_keypoints_before_anms > 8192", KVFE_ERR_UNSUPPORTED);
  if (d.nr_horizontal_bins * d.nr_vertical_bins > KVFE_MAX_BINS || d.nr_horizontal_bins < 1 ||
      d.nr_vertical_bins < 1)
    return fail("bad bin grid", KVFE_ERR_INVALID_ARG);
  if (d.max_features_per_frame < 1) return fail("maxFeaturesPerFrame < 1", KVFE_ERR_INVALID_ARG);
  if (d.enable_subpixel_corner_refinement && (d.subpix_window_size < 1 || d.subpix_window_size > 15))
    return fail("subpixel window_size out of range [1,15]", KVFE_ERR_INVALID_ARG);
  const kvfe_tracker_params& t = p.tracker;
  if (t.klt_win_size < 3 || t.klt_win_size > 40) return fail("klt_win_size out of range", KVFE_ERR_INVALID_ARG);
  if (t.klt_max_iter < 1 || t.klt_max_level < 0 || !(t.klt_eps > 0))
    return fail("bad KLT termination parameters", KVFE_ERR_INVALID_ARG);
  const kvfe_stereo_params& s = p.stereo;
  if (s.templ_cols % 2 != 1 || s.templ_rows % 2 != 1 || s.templ_cols < 3 || s.templ_rows < 1)
    return fail("template size must be odd", KVFE_ERR_INVALID_ARG);
  if (s.stripe_extra_rows % 2 != 0 || s.stripe_extra_rows < 0)
    return fail("stripe_extra_rows must be even", KVFE_ERR_INVALID_ARG);
  if (!(s.min_point_dist > 0)) return fail("minPointDist must be > 0", KVFE_ERR_INVALID_ARG);
  if (s.templ_cols > cfg->left.width || s.templ_rows + s.stripe_extra_rows > cfg->left.height)
    return fail("template larger than image", KVFE_ERR_INVALID_ARG);
  return KVFE_OK;
}

kvfe_status fill_params(kvfe_ctx* c) {
  const kvfe_config& cfg = c->cfg;
  const kvfe_frontend_params& p = cfg.params;
  KParams& P = c->P;
  std::memset(&P, 0, sizeof(P));
  P.W = cfg.left.width;
  P.H = cfg.left.height;
  P.B = cfg.batch;
  const kvfe_detector_params& d = p.detector;
  P.max_features = d.max_features_per_frame;
  P.enable_anms = d.enable_non_max_suppression;
  P.anms_type = d.non_max_suppression_type;
  P.min_distance = d.min_distance;
  P.max_corners = d.max_nr_keypoints_before_anms;
  P.hbins = d.nr_horizontal_bins;
  P.vbins = d.nr_vertical_bins;
  P.block_size = d.block_size;
  P.detector = d.feature_detector_type;
  P.fast_thresh = d.fast_thresh;
  P.use_harris = d.use_harris_detector ? 1 : 0;
  P.harris_k = d.k;
  P.subpix_enable = d.enable_subpixel_corner_refinement;
  P.subpix_win = d.subpix_window_size;
  P.subpix_zero = d.subpix_zero_zone;
  P.subpix_iters = std::min(std::max(d.subpix_max_iters, 1), 100);
  P.sortidx_policy = d.sortidx_policy;
  P.quality = d.quality_level;
  {
    double e = std::max(d.subpix_epsilon, 0.);
    P.subpix_eps2 = e * e;
  }
  const kvfe_tracker_params& t = p.tracker;
  P.klt_win = t.klt_win_size;
  P.klt_iters = std::min(std::max(t.klt_max_iter, 0), 100);
  P.max_age = t.max_feature_track_age;
  P.predictor = t.optical_flow_predictor_type;
  {
    double e = std::min(std::max(t.klt_eps, 0.), 10.);
    P.klt_eps2 = e * e;
  }
  P.disparity_thr = t.disparity_threshold;
  const kvfe_stereo_params& s = p.stereo;
  P.templ_cols = s.templ_cols;
  P.templ_rows = s.templ_rows;
  P.stereo_subpix = s.subpixel_refinement;
  P.use_stereo_tracking = p.use_stereo_tracking;
  P.min_point_dist = s.min_point_dist;
  P.max_point_dist = s.max_point_dist;
  P.tol_template = s.tolerance_template_matching;
  P.fx_rect = c->rect.P1[0];
  P.baseline = c->rect.baseline;
  // StereoMatcher::getRightKeypointsRectified (StereoMatcher.cpp:214-231)
  P.stripe_rows = s.templ_rows + s.stripe_extra_rows;
  int stripe_cols = (int)std::round(P.fx_rect * P.baseline / s.min_point_dist) + s.templ_cols + 4;
  if (stripe_cols % 2 != 1) stripe_cols += 1;
  if (stripe_cols > P.W) stripe_cols = P.W;
  P.stripe_cols = stripe_cols;
  P.min_kf_ns = p.min_intra_keyframe_time_ns;
  P.max_kf_ns = p.max_intra_keyframe_time_ns;
  P.max_disp_lkf = p.max_disparity_since_lkf;
  P.min_features = p.min_number_features;
  P.rgbd = cfg.frontend_type == KVFE_FRONTEND_RGBD ? 1 : 0;
  P.mono = (cfg.frontend_type == KVFE_FRONTEND_MONO || P.rgbd) ? 1 : 0;
  if (P.mono && !P.rgbd) P.use_stereo_tracking = 0;  // mono measurements carry uR = NaN
  P.meas_right = P.rgbd ? 1 : P.use_stereo_tracking;  // RgbdVisionImuFrontend::fillSmartStereoMeasurements keeps uR
  if (P.rgbd) {
    const kvfe_depth_params& d = cfg.depth;
    P.depth_f32 = d.depth_type == KVFE_DEPTH_F32 ? 1 : 0;
    P.depth_to_m = d.depth_to_meters;
    P.depth_min = d.min_depth;
    P.mask_lo_f = d.min_depth * 1.0f / d.depth_to_meters;   // DepthFrame.cpp:77-78
    P.mask_hi_f = d.max_depth * 1.0f / d.depth_to_meters;
    P.mask_lo_u = (int)static_cast<uint16_t>(P.mask_lo_f);
    P.mask_hi_u = (int)static_cast<uint16_t>(P.mask_hi_f);
    P.depth_fx_b = cfg.left.intrinsics[0] * d.virtual_baseline;   // RgbdFrame.cpp:65
    P.baseline = (double)d.virtual_baseline;                      // RgbdCamera::getFakeStereoCalib
  }
  P.use_ransac = p.use_ransac ? 1 : 0;
  P.ransac_2pt_mono = t.ransac_use_2point_mono ? 1 : 0;
  P.ransac_1pt_stereo = t.ransac_use_1point_stereo ? 1 : 0;
  P.ransac_max_iters = t.ransac_max_iterations;
  P.min_mono_inliers = t.min_nr_mono_inliers;
  P.min_stereo_inliers = t.min_nr_stereo_inliers;
  P.ransac_thr_mono = t.ransac_threshold_mono;
  P.ransac_probability = t.ransac_probability;
  P.ransac_thr_stereo = (float)t.ransac_threshold_stereo;
  P.ransac_thr_stereo_d = t.ransac_threshold_stereo;
  // gtsam::Cal3_S2Stereo(P1) of StereoCamera.cpp:75-83
  P.fy_rect = c->rect.P1[5];
  P.cx_rect = c->rect.P1[2];
  P.cy_rect = c->rect.P1[6];
  // pyramid geometry (cv::buildOpticalFlowPyramid: stop when a level is <= the window)
  int w = P.W, h = P.H, off = 0, nl = 0;
  for (int l = 0; l <= t.klt_max_level && l < MAX_LEVELS; l++) {
    P.lw[l] = w;
    P.lh[l] = h;
    P.loff[l] = l == 0 ? 0 : off;
    if (l > 0) off += ((w * h + 63) / 64) * 64;
    nl = l + 1;
    w = (w + 1) / 2;
    h = (h + 1) / 2;
    if (w <= P.klt_win || h <= P.klt_win) break;
  }
  P.nlevels = nl;
  {   // compute units of THIS context's device (a process may hold contexts on several devices)
    int ncu = 0;
    P.n_cu = (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg.device) == hipSuccess && ncu > 0) ? ncu : 256;
  }
  P.ssd_dot4 = cfg.ssd_impl == 1 ? 1 : 0;
  P.lk_one = cfg.lk_impl == 1 ? 1 : 0;
  P.ssd_f32 = p.stereo.ssd_tie_policy == KVFE_SSD_TIE_F32 ? 1 : 0;
  P.klt_maxlevel = nl - 1;
  P.pyr_stride = std::max(off, 64);
  P.acap = ACAP;
  const size_t N = (size_t)P.W * P.H;
  P.ccap = cfg.candidate_capacity > 0 ? cfg.candidate_capacity : (int)std::max<size_t>(N / 4, 4096);
  P.use_pnp = p.use_pnp_tracking ? 1 : 0;
  P.pnp_alg = p.pnp.pnp_algorithm;
  P.pnp_min_inliers = p.pnp.min_pnp_inliers;
  P.pnp_threshold = 1.0 - std::cos(std::atan(std::sqrt(2.0) * p.pnp.ransac_threshold_pnp /
                                             (0.5 * (cfg.left.intrinsics[0] + cfg.left.intrinsics[1]))));
  P.map_cap = cfg.landmark_map_capacity > 0 ? cfg.landmark_map_capacity : 8192;
  P.kcap = P.max_features + P.max_corners + 64;
  {
    // the per-stream bookkeeping and outlier-rejection kernels keep one frame's keypoint ids / indices in LDS: refuse a
    // capacity they cannot hold (and raise their dynamic-LDS opt-in limit on this device) instead of failing at launch
    int limit = track_max_kcap();
    if (c->cfg.params.use_ransac || true)   // (the component-level RANSAC calls use the same capacity)
      limit = std::min(limit, ransac_max_kcap(c->cfg.params.tracker.ransac_use_2point_mono == 0));
    if (P.kcap > limit) {
      c->last_error = "max_features_per_frame + max_nr_keypoints_before_anms + 64 = " + std::to_string(P.kcap) +
                      " keypoints per frame exceed the " + std::to_string(limit) +
                      " the tracking / outlier-rejection kernels hold in LDS";
      return KVFE_ERR_UNSUPPORTED;
    }
  }
  c->pts_bound = P.kcap;
  // (TopN / Binning never return more than need (+ one per bin); the radius-search variants can
  // return any number of corners when their binary search fails, so only kcap bounds them)
  if (P.enable_anms && (P.anms_type == KVFE_ANMS_TOPN || P.anms_type == KVFE_ANMS_BINNING))
    c->pts_bound = std::min(P.kcap, P.max_features + P.hbins * P.vbins + 8);
  return KVFE_OK;
}

kvfe_status build_tables(kvfe_ctx* c) {
  const kvfe_config& cfg = c->cfg;
  const KParams& P = c->P;
  Tables& T = c->T;
  std::memset(&T, 0, sizeof(T));
  const size_t N = (size_t)P.W * P.H;
  // maps
  for (int cam = 0; cam < 2; cam++) {
    const kvfe_camera_params& cp = cam == 0 ? cfg.left : cfg.right;
    c->h_map[cam][0].resize(N);
    c->h_map[cam][1].resize(N);
    TRY(init_undistort_rectify_map(cp, cam == 0 ? c->rect.R1 : c->rect.R2,
                                   cam == 0 ? c->rect.P1 : c->rect.P2, c->h_map[cam][0].data(),
                                   c->h_map[cam][1].data()));
    std::vector<float2> inter(N);
    for (size_t i = 0; i < N; i++) inter[i] = make_float2(c->h_map[cam][0][i], c->h_map[cam][1][i]);
    float2* dm;
    TRY(dalloc(c, &dm, N, false));
    HIPCHK(c, hipMemcpy(dm, inter.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
    T.map[cam] = dm;
    int4* dbox;
    TRY(dalloc(c, &dbox, rectify_box_count(P.W, P.H), false));
    launch_rectify_boxes(dm, P.W, P.H, dbox, nullptr);
    unsigned* dtap;
    TRY(dalloc(c, &dtap, N, false));
    launch_rectify_taps(dm, P.W, P.H, dbox, dtap, nullptr);
    HIPCHK(c, hipDeviceSynchronize());
    T.rect_box[cam] = dbox;
    T.rect_tap[cam] = dtap;
    for (int m = 0; m < 4; m++) {
      const bool useR = m & 1, useP = m & 2;
      c->und[cam][m] = to_dev(make_undistort_ctx(cp, useR ? (cam == 0 ? c->rect.R1 : c->rect.R2) : nullptr,
                                                 useP ? (cam == 0 ? c->rect.P1 : c->rect.P2) : nullptr));
    }
  }
  T.und_left_R = c->und[0][1];
  T.und_left_RP = c->und[0][3];
  // cornerSubPix masks
  {
    std::vector<float> m;
    subpix_mask_table(std::max(P.subpix_win, 1), P.subpix_zero, m);
    float* dmask;
    TRY(dalloc(c, &dmask, m.size(), false));
    HIPCHK(c, hipMemcpy(dmask, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice));
    T.subpix_mask = dmask;
    subpix_mask_table(10, -1, m);
    TRY(dalloc(c, &dmask, m.size(), false));
    HIPCHK(c, hipMemcpy(dmask, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice));
    T.subpix_mask10 = dmask;
  }
  // cv::circle spans
  {
    std::vector<int> hw = circle_half_widths(P.min_distance);
    int* d;
    TRY(dalloc(c, &d, hw.size(), false));
    HIPCHK(c, hipMemcpy(d, hw.data(), sizeof(int) * hw.size(), hipMemcpyHostToDevice));
    T.circle_hw = d;
  }
  {
    unsigned char* d;
    TRY(dalloc(c, &d, KVFE_MAX_BINS, false));
    HIPCHK(c, hipMemcpy(d, cfg.params.detector.binning_mask, KVFE_MAX_BINS, hipMemcpyHostToDevice));
    T.binning_mask = d;
  }
  // cv::sortIdx permutations for every possible list length
  {
    // (every type but TopN / BrownANMS receives the cv::sortIdx-permuted keypoints)
    const int M = (P.enable_anms && P.anms_type >= KVFE_ANMS_SDC) ? std::min(P.max_corners, P.acap) : 0;
    std::vector<unsigned int> off(P.acap + 2, 0);
    size_t total = 0;
    for (int n = 0; n <= M; n++) {
      off[n] = (unsigned int)total;
      total += n;
    }
    for (int n = M + 1; n <= P.acap + 1; n++) off[n] = (unsigned int)total;
    std::vector<uint16_t> tab(std::max<size_t>(total, 1));
    for (int n = 1; n <= M; n++) sortidx_permutation(n, P.sortidx_policy, tab.data() + off[n]);
    unsigned short* dt;
    unsigned int* doff;
    TRY(dalloc(c, &dt, tab.size(), false));
    TRY(dalloc(c, &doff, off.size(), false));
    HIPCHK(c, hipMemcpy(dt, tab.data(), sizeof(uint16_t) * tab.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(doff, off.data(), sizeof(unsigned int) * off.size(), hipMemcpyHostToDevice));
    T.sortidx = dt;
    T.sortidx_off = doff;
  }
  // opengv::sac::SampleConsensusProblem::rnd(): std::mt19937 seeded with 12345 (randomSeed = false)
  // drawn through std::uniform_int_distribution<int>(0, INT_MAX) -- a fixed stream
  {
    const int n = 16384;   // 8 draws per 5-point hypothesis: 2048 hypotheses
    std::vector<int> r(n);
    ransac_rnd_stream(cfg.params.tracker.ransac_rng_policy, n, r.data());
    int* d;
    TRY(dalloc(c, &d, n, false));
    HIPCHK(c, hipMemcpy(d, r.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    T.ransac_rnd = d;
    T.n_ransac_rnd = n;
  }
  // predictor constants: K_ (Matx33f) and K_.inv()
  {
    const M3 K = camera_matrix(cfg.left);
    for (int i = 0; i < 9; i++) T.Kf[i] = (float)K.m[i];
    matx33f_inv(T.Kf, T.Kinvf);
  }
  return KVFE_OK;
}

kvfe_status ensure_comp(kvfe_ctx* c) {
  if (c->comp_ready) return KVFE_OK;
  c->Pc = c->P;
  c->Pc.B = 1;
  TRY(alloc_buffers(c, c->comp, c->Pc));
  TRY(dalloc(c, &c->comp.user_mask, (size_t)c->P.W * c->P.H));
  c->comp_ready = true;
  return KVFE_OK;
}

kvfe_status upload_image(kvfe_ctx* c, unsigned char* dst, const uint8_t* src, size_t stride) {
  HIPCHK(c, hipMemcpy2DAsync(dst, c->P.W, src, stride, c->P.W, c->P.H, hipMemcpyHostToDevice,
                             c->stream));
  return KVFE_OK;
}

// A stage that starts on the stream the previous stage just ended on shares that event (back-to-back kernels: the end of
// one IS the beginning of the next), which nearly halves the timing packets a recorded step puts on the streams.
void prof_begin(kvfe_ctx* c, int stage, hipStream_t st) {
  if (!c->prof_on) return;
  int* idx = c->prof_idx.data() + c->prof_idx.size() - 2 * ST_COUNT;
  if (c->prof_last_end >= 0 && c->prof_last_stream == st) {
    idx[2 * stage] = c->prof_last_end;
    return;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  c->prof_ev.push_back(e);
  idx[2 * stage] = (int)c->prof_ev.size() - 1;
  hipEventRecord(e, st);
}
void prof_end(kvfe_ctx* c, int stage, hipStream_t st) {
  if (!c->prof_on) return;
  int* idx = c->prof_idx.data() + c->prof_idx.size() - 2 * ST_COUNT;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  c->prof_ev.push_back(e);
  idx[2 * stage + 1] = (int)c->prof_ev.size() - 1;
  hipEventRecord(e, st);
  c->prof_last_end = idx[2 * stage + 1];
  c->prof_last_stream = st;
}
// work that is not part of any stage follows on `st`: the next stage there starts with its own event
void prof_break(kvfe_ctx* c) { c->prof_last_end = -1; }

// which stream flag gates a stage's work (0: every stream, every step)
int stage_flag(int s) {
  switch (s) {
    case ST_MINEIG: case ST_SELECT: case ST_SUBPIX: return FLAG_DETECT;
    case ST_RECTIFY: case ST_STEREO: case ST_STEREO_NEW: case ST_RANSAC_STEREO: return FLAG_STEREO;
    case ST_RANSAC_MONO: return FLAG_KEYFRAME;
    default: return 0;
  }
}

void prof_collect(kvfe_ctx* c) {
  if (c->prof_pending.empty()) return;
  // KVFE_PROF_TIMELINE=N (debugging aid): begin / end of every stage of the first N sampled steps of a read, in
  // microseconds from the FIRST sampled step's first event -- the un-profiled counterpart of a rocprofv3 kernel timeline
  // (under rocprofv3 the output transfer runs as a blit kernel and delays the kernels that end beside it)
  static const int timeline = [] { const char* e = std::getenv("KVFE_PROF_TIMELINE"); return e ? std::atoi(e) : 0; }();
  for (size_t k = 0; k < c->prof_pending.size(); k++) {
    const int* idx = c->prof_idx.data() + (size_t)c->prof_pending[k] * 2 * ST_COUNT;
    const int slot = k < c->prof_flag_slot.size() ? c->prof_flag_slot[k] : -1;
    if ((int)k < timeline) {
      const int* idx0 = c->prof_idx.data() + (size_t)c->prof_pending[0] * 2 * ST_COUNT;
      int base = -1;
      for (int s = 0; s < ST_COUNT && base < 0; s++) base = idx0[2 * s];
      std::fprintf(stderr, "KVFE_PROF_TIMELINE sample %zu:", k);
      for (int s = 0; s < ST_COUNT && base >= 0; s++) {
        float b0 = 0.f, b1 = 0.f;
        if (idx[2 * s] < 0 || idx[2 * s + 1] < 0) continue;
        if (hipEventElapsedTime(&b0, c->prof_ev[base], c->prof_ev[idx[2 * s]]) != hipSuccess ||
            hipEventElapsedTime(&b1, c->prof_ev[base], c->prof_ev[idx[2 * s + 1]]) != hipSuccess) continue;
        std::fprintf(stderr, " %s %.0f-%.0f", kStageNames[s], 1e3 * b0, 1e3 * b1);
      }
      std::fprintf(stderr, "\n");
    }
    for (int s = 0; s < ST_COUNT; s++) {
      float ms = 0.f;
      if (idx[2 * s] < 0 || idx[2 * s + 1] < 0) continue;
      if (hipEventElapsedTime(&ms, c->prof_ev[idx[2 * s]], c->prof_ev[idx[2 * s + 1]]) != hipSuccess) continue;
      c->prof_ms[s] += ms;
      int active = c->P.B;
      const int f = stage_flag(s);
      if (f) {
        // a keyframe-only stage: its activity comes from the step's flag word; a sample without one (more than
        // PROF_FLAG_SAMPLES sampled steps between two reads) stays out of the activity statistics instead of
        // counting as "all streams active"
        if (slot < 0 || !c->prof_flags_host) continue;
        active = 0;
        for (int i = 0; i < c->P.B; i++) active += (c->prof_flags_host[(size_t)slot * c->P.B + i] & f) ? 1 : 0;
      }
      if (active > 0) {
        c->prof_ms_active[s] += ms;
        c->prof_active_streams[s] += active;
        c->prof_active_launches[s]++;
      }
    }
    c->prof_samples++;
  }
  for (hipEvent_t e : c->prof_ev) hipEventDestroy(e);
  c->prof_ev.clear();
  c->prof_idx.clear();
  c->prof_pending.clear();
  c->prof_flag_slot.clear();
  c->prof_flag_next = 0;
}

// The step's output records (kvfe_dev.hpp "output side"), enqueued behind step_finalize on the stream `sd` of the tail
// and BEFORE the tail's event: the next step's track_finalize (which clears the landmarks of lost tracks in this frame's
// table) waits for that event, so the records hold the frame as the reference's StereoFrontendOutput would.  Many streams:
// a device-to-device gather (microseconds) on `sd`, then the PCIe transfer by the DMA engine on the output stream, beside
// the next step's tracking launch.  A few streams: the gather writes the mapped pinned slot directly.
// KVFE_HOST_PROF=1 (debugging aid): host time spent inside the calls below, per context, printed by kvfe_destroy
struct HostProf {
  double ms[8] = {};
  long long n[8] = {};
};
static const bool kHostProf = std::getenv("KVFE_HOST_PROF") != nullptr;
static HostProf g_hostprof;
static std::vector<float> g_hostprof_calls;
struct HostTimer {
  int i;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimer(int idx) : i(idx) { if (kHostProf) t0 = std::chrono::steady_clock::now(); }
  ~HostTimer() {
    if (!kHostProf) return;
    g_hostprof.ms[i] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g_hostprof.n[i]++;
  }
};
static void hostprof_print_and_reset(const char* what) {
  if (!kHostProf || !g_hostprof.n[0]) return;
  static const char* names[8] = {"do_step", "enqueue_outputs", "out memcpyAsync", "out_pack launch", "flag poll", "ring wait", "host upload calls", "output event wait"};
  std::fprintf(stderr, "KVFE_HOST_PROF %s:", what);
  for (int i = 0; i < 8; i++)
    if (g_hostprof.n[i]) std::fprintf(stderr, " %s %.4f ms x %lld;", names[i], g_hostprof.ms[i] / g_hostprof.n[i], g_hostprof.n[i]);
  std::fprintf(stderr, "\nKVFE_HOST_PROF out memcpyAsync per call (us):");
  for (float v : g_hostprof_calls) std::fprintf(stderr, " %.0f", 1e3 * v);
  std::fprintf(stderr, "\n");
  g_hostprof_calls.clear();
  g_hostprof = HostProf();
}

kvfe_status enqueue_outputs(kvfe_ctx* c, const FrameTab& K, hipStream_t sd) {
  HostTimer _t(1);
  const int slot = (int)(c->out_steps % OUT_RING);
  Buffers& b = c->fe;
  if (c->out_direct) {
    prof_begin(c, ST_OUT_PACK, sd);
    launch_out_pack(c->P, K, b.st, b.ss, c->out_host_dev[slot], c->out_tab_bytes, c->out_cap, sd);
    prof_end(c, ST_OUT_PACK, sd);
    HIPCHK(c, hipEventRecord(c->ev_out[slot], sd));
    c->out_copied[slot] = c->out_slot_size;
  } else {
    // what the newest COMPLETED transfer shows the records need (entry B of its table), plus a quarter and 64 KB
    for (int back = 1; back < OUT_RING && back <= c->out_steps; back++) {
      const int ps = (int)((c->out_steps - back) % OUT_RING);
      if (hipEventQuery(c->ev_out[ps]) != hipSuccess) continue;
      const size_t need = (size_t)reinterpret_cast<const unsigned long long*>(c->out_host[ps])[c->P.B];
      if (need >= c->out_tab_bytes && need <= c->out_slot_size)
        c->out_guess = std::min(c->out_slot_size, (need + need / 4 + 65536 + 255) & ~(size_t)255);
      break;
    }
    const size_t bytes = c->out_guess_forced ? std::min(c->out_slot_size, std::max(c->out_tab_bytes, c->out_guess_forced))
                                             : c->out_guess;
    if (c->out_steps >= OUT_RING) HIPCHK(c, hipStreamWaitEvent(sd, c->ev_out[slot], 0));   // the slot's last transfer read it
    prof_begin(c, ST_OUT_PACK, sd);
    {
      HostTimer _t3(3);
      launch_out_pack(c->P, K, b.st, b.ss, c->out_stage[slot], c->out_tab_bytes, c->out_cap, sd);
    }
    prof_end(c, ST_OUT_PACK, sd);
    HIPCHK(c, hipEventRecord(c->ev_packed[slot], sd));
    HIPCHK(c, hipStreamWaitEvent(c->out_stream, c->ev_packed[slot], 0));
    prof_begin(c, ST_OUT_TRANSFER, c->out_stream);
    {
      HostTimer _t2(2);
      const auto t0_ = std::chrono::steady_clock::now();
      HIPCHK(c, hipMemcpyAsync(c->out_host[slot], c->out_stage[slot], bytes, hipMemcpyDeviceToHost, c->out_stream));
      if (kHostProf) g_hostprof_calls.push_back((float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count());
    }
    prof_end(c, ST_OUT_TRANSFER, c->out_stream);
    prof_break(c);
    HIPCHK(c, hipEventRecord(c->ev_out[slot], c->out_stream));
    c->out_copied[slot] = bytes;
  }
  c->out_steps++;
  return KVFE_OK;
}

kvfe_status do_step(kvfe_ctx* c, const unsigned char* left, const unsigned char* right,
                    size_t row_stride, size_t img_stride, const kvfe_frame_input* inputs) {
  HostTimer _t0(0);
  const KParams& P = c->P;
  Buffers& b = c->fe;
  hipStream_t st = c->stream;
  // ---- per-stream inputs through the pinned ring ------------------------------------------------
  const int slot = c->ring_pos;
  c->ring_pos = (c->ring_pos + 1) % kvfe_ctx::RING;
  if (c->ring_used[slot]) {
    HostTimer _t5(5);
    HIPCHK(c, hipEventSynchronize(c->ring_ev[slot]));
  }
  unsigned char* hb = c->ring_host[slot];
  double* hR = reinterpret_cast<double*>(hb);
  long long* hts = reinterpret_cast<long long*>(hb + sizeof(double) * 9 * P.B);
  int* hf = reinterpret_cast<int*>(hb + (sizeof(double) * 9 + sizeof(long long)) * P.B);
  for (int s = 0; s < P.B; s++) {
    std::memcpy(hR + 9 * s, inputs[s].keyframe_R_cur_frame, sizeof(double) * 9);
    hts[s] = inputs[s].timestamp_ns;
    hf[s] = inputs[s].force_keyframe;
  }
  // The kernels read this step's inputs straight from the pinned, device-mapped ring slot (a few
  // hundred bytes over PCIe, once per per-stream block): no H2D copy and none of the two launch
  // gaps around it.  The slot is released by the event recorded after the step's last kernel.
  const bool copy_inputs = c->cfg.copy_inputs != 0;
  if (copy_inputs) {
    join_tail(c);   // the previous step's finalisation (side stream) reads the single device copy of the inputs
    HIPCHK(c, hipMemcpyAsync(b.kf_R_cur, hb, (sizeof(double) * 9 + sizeof(long long) + sizeof(int)) * P.B,
                             hipMemcpyHostToDevice, st));
    b.ss.kf_R_cur = b.kf_R_cur;
    b.ss.in_timestamp = b.in_ts;
    b.ss.in_force_kf = b.in_force;
  } else if (c->inputs_by_copy_call && c->in_dev) {
    // STAGED STEPS of many streams: no kernel reads host memory.  A kernel that reads the mapped pinned slot sends its
    // read request up the PCIe link and waits for the answer to come down it -- behind the 46 MB of frames the NEXT step's
    // upload may be pulling down at that moment; the inputs therefore travel by a small copy behind the step's own frames
    // on the copy stream.  Measured (profiles/r5_analysis.md section 4): on one box of the pool 2.57 -> 1.52 ms per staged
    // step of the headline workload (24.9 k -> 42.0 k pairs/s) with the copy against the mapped read in the same call; on
    // two other boxes both forms read 2.3 - 2.5 ms (26 - 28 k) -- there the upload and the step do not overlap whichever
    // way the inputs travel.  Steps fed from device memory keep the mapped slot (the copy and its event cost them 5 %).
    unsigned char* d = c->in_dev + (size_t)slot * c->ring_bytes_dev;
    HIPCHK(c, hipMemcpyAsync(d, hb, (sizeof(double) * 9 + sizeof(long long) + sizeof(int)) * P.B, hipMemcpyHostToDevice,
                             c->copy_stream));
    HIPCHK(c, hipEventRecord(c->in_ev[slot], c->copy_stream));
    HIPCHK(c, hipStreamWaitEvent(st, c->in_ev[slot], 0));
    b.ss.kf_R_cur = reinterpret_cast<const double*>(d);
    b.ss.in_timestamp = reinterpret_cast<const long long*>(d + sizeof(double) * 9 * P.B);
    b.ss.in_force_kf = reinterpret_cast<const int*>(d + (sizeof(double) * 9 + sizeof(long long)) * P.B);
  } else {
    void* dp = nullptr;
    HIPCHK(c, hipHostGetDevicePointer(&dp, hb, 0));
    unsigned char* d = reinterpret_cast<unsigned char*>(dp);
    b.ss.kf_R_cur = reinterpret_cast<const double*>(d);
    b.ss.in_timestamp = reinterpret_cast<const long long*>(d + sizeof(double) * 9 * P.B);
    b.ss.in_force_kf = reinterpret_cast<const int*>(d + (sizeof(double) * 9 + sizeof(long long)) * P.B);
  }
  struct SlotRelease {  // records the slot's event when do_step returns (all kernels enqueued), after the step's last
    kvfe_ctx* c;        // kernels on EITHER stream: once the step has forked, the side stream (which waits for the main
    int slot;           // stream's part before its tail) -- also on an error return between the fork and the tail event
    hipStream_t st, side;
    bool side_used = false, tail_recorded = false;
    ~SlotRelease() {
      if (side_used && side) {
        if (!tail_recorded) {   // error path: the side stream has not been made to wait for the main stream yet
          hipEventRecord(c->ev_main, st);
          hipStreamWaitEvent(side, c->ev_main, 0);
        }
        hipEventRecord(c->ring_ev[slot], side);
      } else {
        hipEventRecord(c->ring_ev[slot], st);
      }
      c->ring_used[slot] = true;
    }
  } slot_release{c, slot, st, c->serial_call ? nullptr : c->side};
  // (serial_call: frames that arrive over PCIe while the step runs -- kvfe_frontend_step_staged -- keep every kernel on the
  // main stream: with a transfer in flight each cross-stream hand-over of the forked step completes late, see there)
  hipStream_t const side = c->serial_call ? nullptr : c->side;

  c->prof_on = c->prof_stride > 0 && (c->prof_step++ % c->prof_stride) == 0;
  if (c->prof_on) {
    c->prof_pending.push_back((int)(c->prof_idx.size() / (2 * ST_COUNT)));
    c->prof_flag_slot.push_back(-1);   // (filled in behind track_finalize: an error return in between keeps the lists aligned)
    c->prof_idx.insert(c->prof_idx.end(), 2 * ST_COUNT, -1);
    c->prof_last_end = -1;
  }
  const FrameTab& K = b.ft[c->role_k];
  const FrameTab& KM1 = b.ft[c->role_km1];
  const FrameTab& LKF = b.ft[c->role_lkf];
  const int pc = c->pyr_cur, pp = pc ^ 1;
  hipStream_t sd = side ? side : st;

  prof_begin(c, ST_PYRAMID, st);
  launch_pyramid(P, left, row_stride, img_stride, b.pyr[pc], st, c->own_level0 ? b.lvl0[pc] : nullptr);
  prof_end(c, ST_PYRAMID, st);
  // The previous step's side-stream work is joined in two places.  The predictor and the tracking launch need its
  // refined new corners and the per-stream state detect_commit wrote (ev_commit) -- joined HERE, behind the pyramid,
  // which does not depend on it and hides the cross-stream hand-over.  Its tail (stereo matching of the new corners,
  // measurements, lkf <- k) is only needed by track_finalize: it runs next to this step's tracking launch.
  if (c->chain_pending) {   // the previous step's rectify / match / reject chain ran on the side stream (fork swapped)
    // It reads nothing this step's tracking writes and writes nothing it reads; what has to wait for it is the release
    // of the previous frame's image buffers.  Frames the caller keeps valid for one more step (device_frames_persist)
    // need no wait here: the chain is followed by the tail on the same stream, and track_finalize joins the tail.
    if (!c->frames_persist_call) {
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_main, 0));
      prof_break(c);
    }
    c->chain_pending = false;
  }
  if (c->commit_pending) {
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_commit, 0));
    prof_break(c);   // (the wait is not part of the tracking stage)
  }
  c->commit_pending = false;
  prof_begin(c, ST_TRACK, st);
  launch_track_prepare(P, c->T, KM1, b.ss, b.lk, st);
  if (c->prev_left) {
    LkScratch lkq = b.lk;
    lkq.skip_age = KM1.age;   // (points past maxFeatureAge are dropped whatever their result: not tracked at all)
    launch_lk(P, c->prev_left, c->prev_row_stride, c->prev_img_stride, b.pyr[pp], left, row_stride,
              img_stride, b.pyr[pc], lkq, c->pts_bound, st, false);
  }
  prof_end(c, ST_TRACK, st);
  if (c->tail_pending) {   // the keyframe decision reads lkf <- k of the previous step's tail and rewrites the stream flags
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_tail, 0));
    c->tail_pending = false;
    prof_break(c);
  }
  static const bool quiet_env = [] { const char* e = std::getenv("KVFE_QUIET_STEPS"); return !e || std::atoi(e) != 0; }();   // (A/B switch)
  const bool quiet_check = quiet_env && c->quiet_check_call && P.B <= 4 && c->out_direct && side && !P.mono && !P.rgbd && !P.use_pnp && c->own_stream;
  if (quiet_check) {
    if (!c->quiet_flags_host) {
      void* h = nullptr;
      void* d = nullptr;
      HIPCHK(c, hipHostMalloc(&h, sizeof(int) * (size_t)P.B, hipHostMallocMapped | hipHostMallocCoherent));
      c->host_allocs.push_back(h);
      std::memset(h, 0, sizeof(int) * (size_t)P.B);
      HIPCHK(c, hipHostGetDevicePointer(&d, h, 0));
      c->quiet_flags_host = reinterpret_cast<int*>(h);
      c->quiet_flags_dev = reinterpret_cast<int*>(d);
    }
    c->quiet_seq = c->quiet_seq % 0x7fffff + 1;
    b.ss.host_flags = c->quiet_flags_dev;
    b.ss.host_seq = c->quiet_seq;
  } else {
    b.ss.host_flags = nullptr;
  }
  prof_begin(c, ST_TRACK_FINALIZE, st);
  launch_track_finalize(P, c->T, KM1, LKF, K, b.ss, b.lk, st);
  prof_end(c, ST_TRACK_FINALIZE, st);
  b.ss.host_flags = nullptr;   // (only track_finalize publishes)
  if (c->prof_on) {   // this step's keyframe / detect / stereo flags, for the per-stage activity of the profile
    int slot = -1;
    if (c->prof_flags_host && c->prof_flag_next < PROF_FLAG_SAMPLES) {
      slot = c->prof_flag_next++;
      HIPCHK(c, hipMemcpyAsync(c->prof_flags_host + (size_t)slot * P.B, b.ss.flags, sizeof(int) * P.B,
                               hipMemcpyDeviceToHost, st));
    }
    c->prof_flag_slot.back() = slot;
  }
  if (c->ev_tracked) {
    HIPCHK(c, hipEventRecord(c->ev_tracked, st));
    c->ev_tracked_valid = true;
  }
  // QUIET STEP (see quiet_check_call).  What a step in which no stream is a keyframe still has to do behind the keyframe
  // decision is detect_commit (per-step state), step_finalize and the output record; every other kernel of the chain
  // returns at its first test of a flag that is off.  Those three are launched SPECULATIVELY right behind track_finalize,
  // gated on the device (KParams::quiet_gate: they return unless no stream of the batch is a keyframe), while the host
  // polls the flags track_finalize publishes: a quiet step is then complete without a host round trip in the middle of
  // it, and a keyframe step goes on as always (three kernels that returned at once: ~20 us on one step of five).
  if (quiet_check) {
    KParams Pq = P;
    Pq.quiet_gate = 1;
    const int oslot = (int)(c->out_steps % OUT_RING);
    prof_begin(c, ST_SUBPIX, st);
    launch_subpix_append(Pq, c->T, left, row_stride, img_stride, K, b.ss, b.ds, 1 | 512, st);   // (detect_commit only)
    prof_end(c, ST_SUBPIX, st);
    prof_begin(c, ST_FINALIZE, st);
    launch_step_finalize(Pq, K, LKF, b.st, b.lst, b.ss, st);
    prof_end(c, ST_FINALIZE, st);
    if (c->right_pending) {   // the record of a step is not complete before its frames have left the caller's memory
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_up2, 0));
      prof_break(c);
    }
    prof_begin(c, ST_OUT_PACK, st);
    launch_out_pack(Pq, K, b.st, b.ss, c->out_host_dev[oslot], c->out_tab_bytes, c->out_cap, st);   // (out_direct: B <= 4)
    prof_end(c, ST_OUT_PACK, st);
    HIPCHK(c, hipEventRecord(c->ev_out[oslot], st));   // (a keyframe step records it again behind its own record)
    prof_break(c);
    // the flag words carry this step's tag once track_finalize has written them; the poll is bounded (2 s), after that
    // the stream is awaited and the device copy of the flags read
    bool quiet = true;
    {
      HostTimer _t4(4);
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(2);
      for (int s = 0; s < P.B && quiet; s++) {
        volatile int* w = c->quiet_flags_host + s;
        int v = *w;
        for (unsigned spins = 0; (v >> 8) != c->quiet_seq; v = *w) {
          if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() > t_end) {
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, hipMemcpy(&v, b.ss.flags + s, sizeof(int), hipMemcpyDeviceToHost));
            v |= c->quiet_seq << 8;
            break;
          }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
        if (v & (FLAG_KEYFRAME | FLAG_DETECT | FLAG_STEREO | FLAG_FIRST)) quiet = false;
      }
    }
    if (quiet) {   // the step is enqueued: what enqueue_outputs and the end of do_step keep track of
      c->out_copied[oslot] = c->out_slot_size;
      c->out_steps++;
      HIPCHK(c, hipGetLastError());
      std::swap(c->role_k, c->role_km1);
      c->pyr_cur ^= 1;
      c->prev_left = c->own_level0 ? b.lvl0[pc] : left;
      c->prev_row_stride = c->own_level0 ? (size_t)P.W : row_stride;
      c->prev_img_stride = c->own_level0 ? (size_t)P.W * P.H : img_stride;
      return KVFE_OK;
    }
  }
  // Rectification right behind the keyframe decision, on the side stream, when the caller FORCES a keyframe on every
  // stream (then the host knows that the step rectifies; otherwise the decision is the device's) and the chain is on the
  // side stream anyway: beside the mono rejection, the min-eigenvalue launch and the selection it runs at the speed it has
  // alone -- 0.046 ms instead of 0.167 ms beside cornerSubPix, whose blocks leave LDS room for one rectification block
  // per CU -- and cornerSubPix loses a neighbour (0.423 -> 0.401 ms): step 1.076 -> 1.070 ms, real frames +1.6 %
  // (tools/r4/gpu_aa.sh).  Without forced keyframes three of four steps would pay the extra hand-over for nothing (-2 %
  // at the reference cadence), so those keep the rectification at the head of the chain.
  bool all_forced = side && !P.mono && (c->fork_swap || (c->frames_persist_call && c->own_stream));
  for (int s = 0; s < P.B && all_forced; s++) all_forced = inputs[s].force_keyframe != 0;
  const bool rect_early = all_forced;
  if (rect_early) {
    HIPCHK(c, hipEventRecord(c->ev_fork, st));
    HIPCHK(c, hipStreamWaitEvent(sd, c->ev_fork, 0));
    if (c->right_pending) HIPCHK(c, hipStreamWaitEvent(sd, c->ev_up2, 0));
    prof_begin(c, ST_RECTIFY, sd);
    const unsigned char* srcs[2] = {left, right};
    launch_rectify(P, c->T, srcs, row_stride, img_stride, b.rect, b.ss.flags, FLAG_STEREO, sd);
    prof_end(c, ST_RECTIFY, sd);
    slot_release.side_used = true;
  }
  // keyframes: mono geometric outlier rejection before detection (it frees landmarks, so more
  // corners are extracted, StereoVisionImuFrontend.cpp:349-363,413-417)
  prof_begin(c, ST_RANSAC_MONO, st);
  launch_mono_ransac(P, c->T, K, LKF, b.ss, b.rs, st);
  prof_end(c, ST_RANSAC_MONO, st);
  prof_begin(c, ST_MINEIG, st);
  if (P.rgbd)  // left_frame->detection_mask_ = depth_img_.getDetectionMask(...) (RgbdVisionImuFrontend.cpp:200,357)
    launch_depth_mask(P, right, row_stride, img_stride, b.ss, FLAG_DETECT, b.depth_mask, st);
  launch_mineig(P, c->T, left, row_stride, img_stride, P.rgbd ? b.depth_mask : nullptr, K, b.ss, b.ds, 1, st);
  prof_end(c, ST_MINEIG, st);
  prof_begin(c, ST_SELECT, st);
  launch_select(P, c->T, K, b.ss, b.ds, -1, st);
  prof_end(c, ST_SELECT, st);
  if (P.mono) {
    // MonoVisionImuFrontend::processFrame (:288-318): refine + append the new corners, undistort all
    // keypoints (Camera::undistortKeypoints), measurements
    prof_begin(c, ST_SUBPIX, st);
    launch_subpix_append(P, c->T, left, row_stride, img_stride, K, b.ss, b.ds, 1, st);
    prof_end(c, ST_SUBPIX, st);
    prof_begin(c, ST_STEREO, st);
    // (tracked entries, incl. those RANSAC just invalidated, + new corners: at most twice the bound of either)
    launch_undistort_left(P, c->T, K, b.st, b.ss, FLAG_STEREO, std::min(P.kcap, 2 * c->pts_bound), st);
    if (P.rgbd)  // RgbdFrame::fillStereoFrame: hallucinated right keypoints, depths and 3-D points from the depth image
      launch_rgbd_fill(P, c->T, right, row_stride, img_stride, K, b.st, b.ss, FLAG_STEREO,
                       std::min(P.kcap, 2 * c->pts_bound), st);
    prof_end(c, ST_STEREO, st);
    if (P.rgbd && P.use_ransac) {
      // outlierRejectionStereo with the fake stereo camera (RgbdVisionImuFrontend.cpp:312-327).  The reference runs
      // it before detection on the tracked keypoints; their table entries are the same after the second
      // fillStereoFrame, and it does not touch landmarks, so running it here gives the same result
      prof_begin(c, ST_RANSAC_STEREO, st);
      launch_stereo_ransac(P, c->T, K, LKF, b.st, b.lst, b.ss, b.rs, c->pts_bound, st);
      prof_end(c, ST_RANSAC_STEREO, st);
    }
    if (P.rgbd && (P.use_pnp || !P.use_ransac))   // RgbdVisionImuFrontend.cpp:328-346
      launch_pnp_frontend(P, c->T, K, b.st, b.ss, b.rs, st);
    prof_begin(c, ST_FINALIZE, st);
    launch_step_finalize(P, K, LKF, b.st, b.lst, b.ss, st);
    prof_end(c, ST_FINALIZE, st);
    TRY(enqueue_outputs(c, K, st));
    HIPCHK(c, hipGetLastError());
    std::swap(c->role_k, c->role_km1);
    c->pyr_cur ^= 1;
    c->prev_left = c->own_level0 ? b.lvl0[pc] : left;
    c->prev_row_stride = c->own_level0 ? (size_t)P.W : row_stride;
    c->prev_img_stride = c->own_level0 ? (size_t)P.W * P.H : img_stride;
    return KVFE_OK;
  }
  // fork: cornerSubPix + append of the new corners (side stream: few waves, latency bound) ||
  // rectify + stereo matching + stereo outlier rejection of the tracked keypoints (main stream:
  // throughput bound); join before the new keypoints are matched.  (Measured alternative: moving the
  // rectify / stereo chain to the side stream right after track_finalize is 10 % slower -- its
  // workgroups delay the one-block-per-stream kernels of the detection chain; stream priorities
  // do not change that.)
  // fa: the stream of the corner refinement, fb: the stream of the rectify / match / reject chain.  Many streams: the
  // chain is the throughput-bound side and stays on the main stream, the refinement forks off.  A few streams
  // (fork_swap): every kernel is a latency, the refinement is on the loop that bounds the step and keeps the main
  // stream -- no cross-stream hand-over on that loop -- and the chain forks off.
  // Many streams take the same arrangement when the caller's frames persist (kvfe_config.device_frames_persist, device
  // frames): the chain then does not have to be joined in front of the next tracking launch, so neither cross-stream
  // hand-over (fork, join: 9 + 21 us in the kernel trace) is on the loop -- +1.9 % on the 64-stream headline, +2.0 % on
  // real frames, +1.1 % at the reference cadence (profiles/r4_analysis.md).  Without that guarantee the chain has to be
  // awaited before the caller may reuse its buffers, and it stays on the main stream.  (Measured and not kept: starting
  // the chain only when the refinement is done -- the refinement gains 0.035 ms, the chain then runs beside the next
  // tracking launch, starved, and the tail gates the keyframe decision: step 1.145 -> 1.263 ms.)
  const bool swap = side && (c->fork_swap || (c->frames_persist_call && c->own_stream && !P.mono));
  hipStream_t fa = swap ? st : sd, fb = swap ? sd : st;
  if (side) {
    HIPCHK(c, hipEventRecord(c->ev_fork, st));
    HIPCHK(c, hipStreamWaitEvent(sd, c->ev_fork, 0));
    slot_release.side_used = true;
  }
  prof_begin(c, ST_SUBPIX, fa);
  launch_subpix_append(P, c->T, left, row_stride, img_stride, K, b.ss, b.ds, 1, fa);
  prof_end(c, ST_SUBPIX, fa);
  if (side && (c->own_stream || swap)) {
    HIPCHK(c, hipEventRecord(c->ev_commit, fa));
    c->commit_pending = !swap;   // (swap: the next step's tracking follows the commit in stream order)
  }
  if (!rect_early) {
  if (c->right_pending) {
    HIPCHK(c, hipStreamWaitEvent(fb, c->ev_up2, 0));
    prof_break(c);
  }
  prof_begin(c, ST_RECTIFY, fb);
  {
    const unsigned char* srcs[2] = {left, right};
    launch_rectify(P, c->T, srcs, row_stride, img_stride, b.rect, b.ss.flags, FLAG_STEREO, fb);
  }
  prof_end(c, ST_RECTIFY, fb);
  }
  prof_begin(c, ST_STEREO, fb);
  launch_stereo(P, c->T, b.rect[0], b.rect[1], K, b.st, b.ss, FLAG_STEREO, c->pts_bound, 1, fb);
  prof_end(c, ST_STEREO, fb);
  // stereo geometric outlier rejection on the matches of the tracked keypoints (:364-387).  Its results (right-keypoint
  // statuses of the outliers, the stereo pose and status; the PnP pose) are read by the tail only -- the landmark removal
  // that the next step's tracking reads is the mono rejection's -- so it runs at the head of the tail on the side stream,
  // off the main stream's critical path (round 3: -1.2 % step time on four A/B pairs, +7 % on configs[4]).
  // NOTE for whoever edits launch_stereo_ransac / launch_pnp_frontend: they run concurrently with the NEXT step's
  // track_prepare and tracking launch and must therefore not write frame-table fields those read (K.kp, K.lmk, K.count).
  const bool ransac_tail = side && !swap;
  auto stereo_rejection = [&](hipStream_t rs) -> kvfe_status {
  prof_begin(c, ST_RANSAC_STEREO, rs);
  if (P.use_ransac) {
    // (the device's own predicate, rot_is_identity of kvfe_dev.hpp: a stream without a usable gyro rotation takes the
    // 3-point problem; the launch of that kernel is skipped when no stream of the batch does)
    bool need_arun = !P.ransac_1pt_stereo;
    for (int s = 0; s < P.B && !need_arun; s++) need_arun = rot_is_identity(inputs[s].keyframe_R_cur_frame);
    launch_stereo_ransac(P, c->T, K, LKF, b.st, b.lst, b.ss, b.rs, c->pts_bound, rs, need_arun);
  }
  // outlierRejectionPnP(*stereoFrame_k_) (:389-399): after the stereo rejection, before detection
  if (P.use_pnp && P.use_ransac) launch_pnp_frontend(P, c->T, K, b.st, b.ss, b.rs, rs);
  prof_end(c, ST_RANSAC_STEREO, rs);
  return KVFE_OK;
  };
  if (!ransac_tail) TRY(stereo_rejection(fb));
  // the tail -- stereo matching of the new corners, finalisation -- runs on the side stream behind both parts of the
  // fork (the part on the main stream is awaited there); it is joined by the next step before its keyframe decision
  // (or by kvfe_synchronize / kvfe_frontend_get_output)
  if (side) {
    if (swap) {
      HIPCHK(c, hipEventRecord(c->ev_main, sd));            // the chain (last reader of this frame's image slots) is done
      if (c->staging_ready) HIPCHK(c, hipEventRecord(c->ev_chain[c->chain_seq % 4], sd));
      c->chain_seq++;
      c->chain_pending = true;
      HIPCHK(c, hipStreamWaitEvent(sd, c->ev_commit, 0));   // the refined new corners (main stream)
    } else {
      HIPCHK(c, hipEventRecord(c->ev_main, st));
      HIPCHK(c, hipStreamWaitEvent(sd, c->ev_main, 0));
    }
  }
  if (ransac_tail) TRY(stereo_rejection(sd));
  prof_begin(c, ST_STEREO_NEW, sd);
  launch_stereo(P, c->T, b.rect[0], b.rect[1], K, b.st, b.ss, FLAG_STEREO, c->pts_bound, 2, sd);
  prof_end(c, ST_STEREO_NEW, sd);
  prof_begin(c, ST_FINALIZE, sd);
  launch_step_finalize(P, K, LKF, b.st, b.lst, b.ss, sd);
  prof_end(c, ST_FINALIZE, sd);
  if (c->right_pending) HIPCHK(c, hipStreamWaitEvent(sd, c->ev_up2, 0));   // (complete = the frames have left the caller's memory)
  TRY(enqueue_outputs(c, K, sd));
  if (side) {
    HIPCHK(c, hipEventRecord(c->ev_tail, sd));
    c->tail_pending = true;
    slot_release.tail_recorded = true;
    if (!c->own_stream) {
      // caller-owned stream (kvfe_config.hip_stream): work the caller enqueues on it after this call must be ordered
      // after the WHOLE step, so the tail is joined here instead of being deferred to the next step
      HIPCHK(c, hipStreamWaitEvent(st, c->ev_tail, 0));
      c->tail_pending = false;
    }
  }
  HIPCHK(c, hipGetLastError());

  // stereoFrame_km1_ = stereoFrame_k_
  std::swap(c->role_k, c->role_km1);
  c->pyr_cur ^= 1;
  c->prev_left = c->own_level0 ? b.lvl0[pc] : left;
  c->prev_row_stride = c->own_level0 ? (size_t)P.W : row_stride;
  c->prev_img_stride = c->own_level0 ? (size_t)P.W * P.H : img_stride;
  return KVFE_OK;
}

// all groups of a parent context: group g starts its step once group g-1 (cyclically: the last
// group's previous step) has issued its tracking stage
kvfe_status step_groups(kvfe_ctx* c, const unsigned char* left, const unsigned char* right,
                        size_t row_stride, size_t img_stride, const kvfe_frame_input* inputs,
                        bool host_input) {
  const int G = (int)c->children.size();
  for (int g = 0; g < G; g++) {
    kvfe_ctx* ch = c->children[g];
    kvfe_ctx* pred = c->children[(g + G - 1) % G];
    if (G > 1 && pred->ev_tracked_valid) HIPCHK(c, hipStreamWaitEvent(ch->stream, pred->ev_tracked, 0));
    const unsigned char* l = left + (size_t)ch->s0 * img_stride;
    const unsigned char* r = right + (size_t)ch->s0 * img_stride;
    // through the child's own entry point, so that whatever the entry point does before do_step (uploads,
    // cv::equalizeHist into the ctx-owned slots when equalizeImage is on) also happens per group
    kvfe_status s = host_input
                        ? kvfe_frontend_step_host(ch, l, r, row_stride, img_stride, inputs + ch->s0)
                        : kvfe_frontend_step_device(ch, l, r, row_stride, img_stride, inputs + ch->s0);
    if (s != KVFE_OK) {
      c->last_error = ch->last_error;
      return s;
    }
  }
  return KVFE_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* kvfe_version(void) { return "libkvfe 0.1 (gfx950, HIP)"; }

const char* kvfe_status_string(kvfe_status s) {
  switch (s) {
    case KVFE_OK: return "ok";
    case KVFE_ERR_INVALID_ARG: return "invalid argument";
    case KVFE_ERR_UNSUPPORTED: return "unsupported configuration";
    case KVFE_ERR_NO_DEVICE: return "no usable gfx950 device";
    case KVFE_ERR_HIP: return "HIP runtime error";
    case KVFE_ERR_CAPACITY: return "device list capacity exceeded";
    case KVFE_ERR_NOT_READY: return "not ready";
    default: return "unknown";
  }
}

const char* kvfe_last_error(const kvfe_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

void kvfe_default_frontend_params(kvfe_frontend_params* p) {
  std::memset(p, 0, sizeof(*p));
  kvfe_detector_params& d = p->detector;
  d.feature_detector_type = KVFE_DET_GFTT;
  d.max_features_per_frame = 400;
  d.enable_subpixel_corner_refinement = 1;
  d.subpix_window_size = 10;
  d.subpix_zero_zone = -1;
  d.subpix_max_iters = 10;
  d.subpix_epsilon = 0.01;
  d.enable_non_max_suppression = 1;
  d.non_max_suppression_type = KVFE_ANMS_RANGETREE;
  d.min_distance = 10;
  d.max_nr_keypoints_before_anms = 2000;
  d.nr_horizontal_bins = 5;
  d.nr_vertical_bins = 5;
  for (int i = 0; i < 25; i++) d.binning_mask[i] = 1;
  d.quality_level = 0.001;
  d.block_size = 3;
  d.k = 0.04;
  d.sortidx_policy = KVFE_SORTIDX_LIBSTDCXX;
  d.fast_thresh = 10;                       // FeatureDetectorParams.h:105
  kvfe_tracker_params& t = p->tracker;
  t.klt_win_size = 24;
  t.klt_max_iter = 30;
  t.klt_max_level = 3;
  t.max_feature_track_age = 25;
  t.klt_eps = 0.01;
  t.optical_flow_predictor_type = KVFE_FLOW_NO_PREDICTION;
  t.disparity_threshold = 0.5;
  t.min_nr_mono_inliers = 10;
  t.min_nr_stereo_inliers = 5;
  t.ransac_threshold_mono = 1.0e-6;
  t.ransac_threshold_stereo = 1.0;
  t.ransac_max_iterations = 100;
  t.ransac_randomize = 1;  // class default (VisionImuTrackerParams.h:64); every shipped YAML sets 0
  t.ransac_probability = 0.995;
  t.ransac_use_1point_stereo = 1;
  t.ransac_use_2point_mono = 1;
  t.ransac_rng_policy = KVFE_RNG_LIBSTDCXX_PRE11;
  t.pose_2d2d_algorithm = 1;  // Pose2d2dAlgorithm::NISTER (VisionImuTrackerParams.h:68)
  kvfe_stereo_params& s = p->stereo;
  s.tolerance_template_matching = 0.15;
  s.templ_cols = 101;
  s.templ_rows = 11;
  s.min_point_dist = 0.1;
  s.max_point_dist = 15.0;
  s.equalize_image = 0;
  p->min_intra_keyframe_time_ns = 0.2 * 10e6;
  p->max_intra_keyframe_time_ns = 10.0 * 10e6;
  p->max_disparity_since_lkf = 200.0;
  p->use_stereo_tracking = 1;
  p->use_ransac = 1;  // VisionImuFrontendParams.h:56
  p->use_pnp_tracking = 1;                 // VisionImuFrontendParams.h:59 (every shipped YAML but KinectAzure sets 0)
  p->pnp.pnp_algorithm = 3;                // Pose3d2dAlgorithm::EPNP (VisionImuTrackerParams.h:75)
  p->pnp.min_pnp_inliers = 10;             // VisionImuTrackerParams.h:73
  p->pnp.ransac_threshold_pnp = 1.0;       // :74
  p->pnp.optimize_2d3d_pose_from_inliers = 0;
}

kvfe_status kvfe_compute_rectification(const kvfe_camera_params* left,
                                       const kvfe_camera_params* right, kvfe_rectification* out) {
  if (!left || !right || !out) return KVFE_ERR_INVALID_ARG;
  return stereo_rectify(*left, *right, out);
}

kvfe_status kvfe_compute_undistort_rectify_maps(const kvfe_camera_params* cam, const double R[9],
                                                const double P[12], float* map_x, float* map_y) {
  if (!cam || !R || !P || !map_x || !map_y) return KVFE_ERR_INVALID_ARG;
  return init_undistort_rectify_map(*cam, R, P, map_x, map_y);
}

static kvfe_status create_one(const kvfe_config* cfg, kvfe_ctx* parent, int s0, int batch,
                              bool alloc_frontend, kvfe_ctx** out) {
  kvfe_ctx* c = new kvfe_ctx();
  c->cfg = *cfg;
  c->cfg.batch = batch;
  c->parent = parent;
  c->s0 = s0;
  kvfe_status s = KVFE_OK;
  if (parent) {
    c->rect = parent->rect;
  } else if (cfg->frontend_type == KVFE_FRONTEND_MONO || cfg->frontend_type == KVFE_FRONTEND_RGBD) {
    // Camera::Camera (src/frontend/Camera.cpp:29-49): no rectification, R = I and P = K
    std::memset(&c->rect, 0, sizeof(c->rect));
    const M3 K = camera_matrix(cfg->left);
    for (int i = 0; i < 3; i++) {
      c->rect.R1[i * 4] = c->rect.R2[i * 4] = 1.0;
      for (int j = 0; j < 3; j++) c->rect.P1[i * 4 + j] = c->rect.P2[i * 4 + j] = K.m[i * 3 + j];
    }
    c->cfg.right = cfg->left;
    if (cfg->frontend_type == KVFE_FRONTEND_RGBD) c->rect.baseline = (double)cfg->depth.virtual_baseline;
  } else {
    s = stereo_rectify(cfg->left, cfg->right, &c->rect);
  }
  if (s != KVFE_OK) {
    delete c;
    return s;
  }
  if (cfg->hip_stream && !parent) {
    c->stream = reinterpret_cast<hipStream_t>(cfg->hip_stream);
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return KVFE_ERR_HIP;
    }
    c->own_stream = true;
  }
  s = fill_params(c);
  if (s == KVFE_OK) {
    if (parent) {  // constant tables are shared with (and owned by) the parent
      c->T = parent->T;
      std::memcpy(c->und, parent->und, sizeof(c->und));
    } else {
      s = build_tables(c);
    }
  }
  if (s == KVFE_OK && alloc_frontend) s = alloc_buffers(c, c->fe, c->P);
  if (s == KVFE_OK && alloc_frontend) {
    c->ring_bytes = (sizeof(double) * 9 + sizeof(long long) + sizeof(int)) * (size_t)c->P.B + 64;
    for (int i = 0; i < kvfe_ctx::RING && s == KVFE_OK; i++) {
      void* h = nullptr;
      if (hipHostMalloc(&h, c->ring_bytes, hipHostMallocDefault) != hipSuccess) s = KVFE_ERR_HIP;
      c->ring_host[i] = reinterpret_cast<unsigned char*>(h);
      if (s == KVFE_OK) c->host_allocs.push_back(h);
      if (s == KVFE_OK && hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming) != hipSuccess)
        s = KVFE_ERR_HIP;
    }
  }
  if (s == KVFE_OK && alloc_frontend) {
    // (a frame table holds at most the tracked entries plus the new corners, each bounded by pts_bound -- the bound the
    // per-keypoint launches of the step use -- although it is allocated for kcap)
    c->out_cap = std::min(c->P.kcap, 2 * c->pts_bound);
    c->out_tab_bytes = out_table_bytes(c->P.B);
    c->out_slot_size = out_slot_bytes(c->P.B, c->out_cap);
    c->out_guess = c->out_slot_size;   // (until a transfer has completed)
    if (const char* e = getenv("KVFE_OUT_TRANSFER_BYTES")) c->out_guess_forced = (size_t)std::max(0ll, atoll(e));
    c->out_direct = c->P.B <= 4 || cfg->single_hip_stream != 0;
    const size_t bytes = c->out_slot_size;
    for (int i = 0; i < OUT_RING && s == KVFE_OK; i++) {
      void* h = nullptr;
      void* d = nullptr;
      if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) s = KVFE_ERR_HIP;
      if (s == KVFE_OK) {
        c->host_allocs.push_back(h);
        c->out_host[i] = reinterpret_cast<unsigned char*>(h);
        std::memset(h, 0, std::min<size_t>(bytes, OUT_HDR_BYTES));
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) s = KVFE_ERR_HIP;
        c->out_host_dev[i] = reinterpret_cast<unsigned char*>(d);
      }
      if (s == KVFE_OK && !c->out_direct) s = dalloc(c, &c->out_stage[i], bytes);
      if (s == KVFE_OK && (hipEventCreateWithFlags(&c->ev_packed[i], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming) != hipSuccess))
        s = KVFE_ERR_HIP;
    }
    if (s == KVFE_OK && !c->out_direct && hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking) != hipSuccess)
      s = KVFE_ERR_HIP;
    // every SDMA engine's queue is created here instead of inside a hipMemcpyAsync of the step loop (host_dma_warm.cpp)
    // (also for a few streams, whose records need no transfer: their frames come up the link through the same engines)
    if (s == KVFE_OK && hipStreamSynchronize(c->stream) == hipSuccess)   // (the buffers' zero fill)
      (void)warm_dma_engines(cfg->device, c->fe.raw_left[0], c->out_host[0], std::min<size_t>(bytes, (size_t)65536));
  }
  if (s == KVFE_OK && parent &&
      hipEventCreateWithFlags(&c->ev_tracked, hipEventDisableTiming) != hipSuccess)
    s = KVFE_ERR_HIP;
  const bool no_side = cfg->single_hip_stream != 0;
  if (s == KVFE_OK && alloc_frontend && !no_side) {
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_mono, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_commit, hipEventDisableTiming) != hipSuccess)
      s = KVFE_ERR_HIP;
    // a few streams are latency bound: the loop refinement -> tracking -> keyframe decision -> detection -> refinement
    // is what a step costs, so it stays on ONE stream and the rectify / match / reject chain takes the side stream
    // (many streams: the same arrangement for the calls whose frames persist, do_step)
    c->fork_swap = c->P.B <= 4 && c->own_stream && !c->P.mono;
  }
  if (s != KVFE_OK) {
    std::fprintf(stderr, "kvfe_create failed: %s\n", c->last_error.c_str());
    kvfe_destroy(c);
    return s;
  }
  *out = c;
  return KVFE_OK;
}

kvfe_status kvfe_create(const kvfe_config* cfg, kvfe_ctx** out) {
  if (!cfg || !out) return KVFE_ERR_INVALID_ARG;
  *out = nullptr;
  std::string why;
  kvfe_status vs = validate(cfg, &why);
  if (vs != KVFE_OK) {
    std::fprintf(stderr, "kvfe_create: %s\n", why.c_str());
    return vs;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev) {
    std::fprintf(stderr, "kvfe_create: no HIP device (libkvfe has no CPU fallback)\n");
    return KVFE_ERR_NO_DEVICE;
  }
  DeviceGuard _dev(cfg->device);   // the caller's current device is restored on return
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) return KVFE_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return KVFE_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "kvfe_create: device is %s, libkvfe is built for gfx950 only\n",
                 prop.gcnArchName);
    return KVFE_ERR_NO_DEVICE;
  }
  // number of stream groups: explicit, or by batch size (a caller-owned stream keeps one group so
  // that all work stays ordered on that stream)
  int groups = cfg->stream_groups;
  if (groups < 0) return KVFE_ERR_INVALID_ARG;
  if (cfg->hip_stream) groups = 1;
  // measured on MI355X (profiles/r1_groups_sweep.md): with the current kernels one group is fastest at
  // every batch size (the latency-bound kernels do not shrink with the group), so 0 means 1
  if (groups == 0) groups = 1;
  groups = std::max(1, std::min(groups, cfg->batch));
  kvfe_ctx* c = nullptr;
  kvfe_status s = create_one(cfg, nullptr, 0, cfg->batch, groups == 1, &c);
  if (s != KVFE_OK) return s;
  if (groups > 1) {
    if (hipStreamSynchronize(c->stream) != hipSuccess) {  // shared tables are complete
      kvfe_destroy(c);
      return KVFE_ERR_HIP;
    }
    const int base = cfg->batch / groups, rem = cfg->batch % groups;
    int s0 = 0;
    for (int g = 0; g < groups && s == KVFE_OK; g++) {
      const int bg = base + (g < rem ? 1 : 0);
      kvfe_ctx* ch = nullptr;
      s = create_one(cfg, c, s0, bg, true, &ch);
      if (s == KVFE_OK) c->children.push_back(ch);
      s0 += bg;
    }
    if (s != KVFE_OK) {
      kvfe_destroy(c);
      return s;
    }
  }
  *out = c;
  return KVFE_OK;
}

void kvfe_destroy(kvfe_ctx* c) {
  if (c && c->children.empty()) hostprof_print_and_reset("context destroyed");
  DeviceGuard _dev(c);
  if (!c) return;
  for (kvfe_ctx* ch : c->children) kvfe_destroy(ch);
  c->children.clear();
  if (c->side) hipStreamSynchronize(c->side);   // (the last step's tail lives there)
  if (c->stream) hipStreamSynchronize(c->stream);
  for (hipEvent_t e : c->prof_ev) hipEventDestroy(e);
  for (int i = 0; i < kvfe_ctx::RING; i++) {
    if (c->ring_ev[i]) hipEventDestroy(c->ring_ev[i]);
    if (c->in_ev[i]) hipEventDestroy(c->in_ev[i]);
  }

  if (c->ev_tracked) hipEventDestroy(c->ev_tracked);
  if (c->side) {
    hipStreamSynchronize(c->side);
    hipStreamDestroy(c->side);
  }
  if (c->copy_stream) {
    hipStreamSynchronize(c->copy_stream);
    hipStreamDestroy(c->copy_stream);
  }
  if (c->up2_stream) {
    hipStreamSynchronize(c->up2_stream);
    hipStreamDestroy(c->up2_stream);
  }
  if (c->ev_up2) hipEventDestroy(c->ev_up2);
  if (c->out_stream) {
    hipStreamSynchronize(c->out_stream);
    hipStreamDestroy(c->out_stream);
  }
  if (c->topup_stream) hipStreamDestroy(c->topup_stream);
  for (int i = 0; i < OUT_RING; i++) {
    if (c->ev_packed[i]) hipEventDestroy(c->ev_packed[i]);
    if (c->ev_out[i]) hipEventDestroy(c->ev_out[i]);
  }
  for (int i = 0; i < KVFE_STAGING_SLOTS; i++)
    if (c->stage_copied[i]) hipEventDestroy(c->stage_copied[i]);
  for (int i = 0; i < 4; i++) {
    if (c->step_done[i]) hipEventDestroy(c->step_done[i]);
    if (c->ev_chain[i]) hipEventDestroy(c->ev_chain[i]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  if (c->ev_join) hipEventDestroy(c->ev_join);
  if (c->ev_mono) hipEventDestroy(c->ev_mono);
  if (c->ev_main) hipEventDestroy(c->ev_main);
  if (c->ev_tail) hipEventDestroy(c->ev_tail);
  if (c->ev_commit) hipEventDestroy(c->ev_commit);

  for (void* p : c->allocs) dev_free(p);
  for (void* p : c->dense_allocs) dev_free(p);
  for (int i = 0; i < 2; i++)
    if (c->dense_ev[i]) hipEventDestroy(c->dense_ev[i]);
  for (void* p : c->host_allocs) hipHostFree(p);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

kvfe_status kvfe_get_rectification(const kvfe_ctx* c, kvfe_rectification* out) {
  DeviceGuard _dev(c);
  if (!c || !out) return KVFE_ERR_INVALID_ARG;
  *out = c->rect;
  return KVFE_OK;
}

kvfe_status kvfe_synchronize(kvfe_ctx* c) {
  DeviceGuard _dev(c);
  join_tail(c);
  if (!c) return KVFE_ERR_INVALID_ARG;
  for (kvfe_ctx* ch : c->children) TRY(kvfe_synchronize(ch));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  return KVFE_OK;
}

// ---- component level ---------------------------------------------------------------------------
kvfe_status kvfe_undistort_rectify_image(kvfe_ctx* c, int32_t cam, const uint8_t* src,
                                         size_t src_stride, uint8_t* dst, size_t dst_stride) {
  DeviceGuard _dev(c);
  if (!c || !src || !dst || cam < 0 || cam > 1) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  TRY(upload_image(c, b.raw_left[0], src, src_stride));
  const unsigned char* srcs[2] = {b.raw_left[0], b.raw_left[0]};
  unsigned char* dsts[2] = {b.rect[0], b.rect[1]};
  launch_rectify(c->Pc, c->T, srcs, c->P.W, (size_t)c->P.W * c->P.H, dsts, nullptr, 0, c->stream);
  HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, b.rect[cam], c->P.W, c->P.W, c->P.H,
                             hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KVFE_OK;
}

kvfe_status kvfe_equalize_hist(kvfe_ctx* c, const uint8_t* src, size_t src_stride, uint8_t* dst,
                               size_t dst_stride) {
  DeviceGuard _dev(c);
  if (!c || !src || !dst) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  TRY(upload_image(c, b.raw_left[0], src, src_stride));
  launch_equalize_hist(c->P.W, c->P.H, 1, b.raw_left[0], c->P.W, (size_t)c->P.W * c->P.H, b.raw_left[1],
                       b.eq_hist, c->stream);
  HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, b.raw_left[1], c->P.W, c->P.W, c->P.H, hipMemcpyDeviceToHost,
                             c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KVFE_OK;
}

static kvfe_status undistort_common(kvfe_ctx* c, int cam, const float* xy, int n, int useR, int useP,
                                    float* out_xy, double* out_versors) {
  DeviceGuard _dev(c);
  if (!c || !xy || n < 0 || cam < 0 || cam > 1) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  if (n > c->Pc.kcap) return KVFE_ERR_CAPACITY;
  Buffers& b = c->comp;
  HIPCHK(c, hipMemcpyAsync(b.lk.prev_pts, xy, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
  const UndistortDev& U = c->und[cam][(useR ? 1 : 0) + (useP ? 2 : 0)];
  launch_undistort_points(U, b.lk.prev_pts, n, out_xy ? b.lk.next_pts : nullptr,
                          out_versors ? b.ft[0].versor : nullptr, c->stream);
  if (out_xy)
    HIPCHK(c, hipMemcpyAsync(out_xy, b.lk.next_pts, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
  if (out_versors)
    HIPCHK(c, hipMemcpyAsync(out_versors, b.ft[0].versor, sizeof(double) * 3 * n,
                             hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KVFE_OK;
}

kvfe_status kvfe_undistort_rectify_keypoints(kvfe_ctx* c, int32_t cam, const float* xy, int32_t n,
                                             int32_t use_R, int32_t use_P, float* out_xy) {
  DeviceGuard _dev(c);
  if (!out_xy) return KVFE_ERR_INVALID_ARG;
  return undistort_common(c, cam, xy, n, use_R, use_P, out_xy, nullptr);
}

kvfe_status kvfe_get_bearing_vectors(kvfe_ctx* c, int32_t cam, const float* xy, int32_t n,
                                     double* out_versors) {
  DeviceGuard _dev(c);
  if (!out_versors) return KVFE_ERR_INVALID_ARG;
  return undistort_common(c, cam, xy, n, 1, 0, nullptr, out_versors);
}

static kvfe_status detect_common(kvfe_ctx* c, const uint8_t* img, size_t stride,
                                 const uint8_t* mask, size_t mask_stride, const float* tracked_xy,
                                 int n_tracked, int need, bool raw, float* out_xy, int capacity,
                                 int* out_n) {
  DeviceGuard _dev(c);
  if (!c || !img || !out_xy || !out_n || n_tracked < 0) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n_tracked > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  TRY(upload_image(c, b.raw_left[0], img, stride));
  if (mask)
    HIPCHK(c, hipMemcpy2DAsync(b.user_mask, P.W, mask, mask_stride, P.W, P.H, hipMemcpyHostToDevice, st));
  const FrameTab& K = b.ft[0];
  if (n_tracked > 0)
    HIPCHK(c, hipMemcpyAsync(K.kp, tracked_xy, sizeof(float2) * n_tracked, hipMemcpyHostToDevice, st));
  const int flags = FLAG_DETECT | FLAG_INIT;
  HIPCHK(c, hipMemcpyAsync(K.count, &n_tracked, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.ss.flags, &flags, sizeof(int), hipMemcpyHostToDevice, st));
  launch_mineig(P, c->T, b.raw_left[0], P.W, (size_t)P.W * P.H, mask ? b.user_mask : nullptr, K,
                b.ss, b.ds, raw ? 0 : 1, st);
  launch_select(P, c->T, K, b.ss, b.ds, std::max(need, 0), st);
  int n = 0, fl = 0;
  if (raw) {
    HIPCHK(c, hipMemcpyAsync(&n, b.ds.n_corners, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&fl, b.ss.flags, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int m = std::min(n, capacity);
    if (m > 0) HIPCHK(c, hipMemcpy(out_xy, b.ds.corners, sizeof(float2) * m, hipMemcpyDeviceToHost));
  } else {
    launch_subpix_append(P, c->T, b.raw_left[0], P.W, (size_t)P.W * P.H, K, b.ss, b.ds, 0, st);
    HIPCHK(c, hipMemcpyAsync(&n, b.ds.n_new, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&fl, b.ss.flags, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const int m = std::min(n, capacity);
    if (m > 0) HIPCHK(c, hipMemcpy(out_xy, b.ds.newc, sizeof(float2) * m, hipMemcpyDeviceToHost));
  }
  *out_n = n;
  if (fl & FLAG_OVERFLOW) {
    c->last_error = "device candidate / corner list capacity exceeded";
    return KVFE_ERR_CAPACITY;
  }
  return KVFE_OK;
}

kvfe_status kvfe_raw_feature_detection(kvfe_ctx* c, const uint8_t* img, size_t stride,
                                       const uint8_t* mask, size_t mask_stride, float* out_xy,
                                       int32_t capacity, int32_t* out_n) {
  DeviceGuard _dev(c);
  return detect_common(c, img, stride, mask, mask_stride, nullptr, 0, 0, true, out_xy, capacity, out_n);
}

kvfe_status kvfe_feature_detection(kvfe_ctx* c, const uint8_t* img, size_t stride,
                                   const float* tracked_xy, int32_t n_tracked,
                                   int32_t need_n_corners, float* out_xy, int32_t capacity,
                                   int32_t* out_n) {
  DeviceGuard _dev(c);
  if (n_tracked > 0 && !tracked_xy) return KVFE_ERR_INVALID_ARG;
  return detect_common(c, img, stride, nullptr, 0, tracked_xy, n_tracked, need_n_corners, false,
                       out_xy, capacity, out_n);
}

kvfe_status kvfe_corner_subpix(kvfe_ctx* c, const uint8_t* img, size_t stride, float* xy, int32_t n,
                               int32_t half_win, int32_t zero_zone, int32_t max_iters, double eps) {
  DeviceGuard _dev(c);
  if (!c || !img || !xy || n < 0 || half_win < 1 || half_win > 15) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  if (n > c->Pc.kcap) return KVFE_ERR_CAPACITY;
  TRY(upload_image(c, b.raw_left[0], img, stride));
  std::vector<float> m;
  subpix_mask_table(half_win, zero_zone, m);
  float* dmask = reinterpret_cast<float*>(b.ds.sortbuf);  // scratch
  HIPCHK(c, hipMemcpyAsync(dmask, m.data(), sizeof(float) * m.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(b.lk.prev_pts, xy, sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
  const int it = std::min(std::max(max_iters, 1), 100);
  const double e = std::max(eps, 0.);
  launch_subpix_points(c->Pc, dmask, b.raw_left[0], c->P.W, c->P.W, c->P.H, b.lk.prev_pts, n,
                       half_win, it, e * e, c->stream);
  HIPCHK(c, hipMemcpyAsync(xy, b.lk.prev_pts, sizeof(float2) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KVFE_OK;
}

kvfe_status kvfe_calc_optical_flow_pyr_lk(kvfe_ctx* c, const uint8_t* prev_img,
                                          const uint8_t* cur_img, size_t stride,
                                          const float* prev_xy, float* cur_xy, int32_t n,
                                          uint8_t* status, float* err) {
  DeviceGuard _dev(c);
  if (!c || !prev_img || !cur_img || !prev_xy || !cur_xy || !status || n < 0) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  TRY(upload_image(c, b.raw_left[0], prev_img, stride));
  TRY(upload_image(c, b.raw_left[1], cur_img, stride));
  const size_t N = (size_t)P.W * P.H;
  launch_pyramid(P, b.raw_left[0], P.W, N, b.pyr[0], st);
  launch_pyramid(P, b.raw_left[1], P.W, N, b.pyr[1], st);
  HIPCHK(c, hipMemcpyAsync(b.lk.prev_pts, prev_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.lk.next_pts, cur_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.lk.npts, &n, sizeof(int), hipMemcpyHostToDevice, st));
  launch_lk(P, b.raw_left[0], P.W, N, b.pyr[0], b.raw_left[1], P.W, N, b.pyr[1], b.lk, n, st);
  HIPCHK(c, hipMemcpyAsync(cur_xy, b.lk.next_pts, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(status, b.lk.status, n, hipMemcpyDeviceToHost, st));
  if (err) HIPCHK(c, hipMemcpyAsync(err, b.lk.err, sizeof(float) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_build_optical_flow_pyramid(kvfe_ctx* c, const uint8_t* imgs, size_t row_stride, size_t image_stride,
                                            int32_t n_images, uint8_t* levels_out, size_t levels_capacity,
                                            int32_t* level_sizes_out, int32_t* n_levels_out, uint8_t* level0_copy_out) {
  DeviceGuard _dev(c);
  if (!c || !imgs || !levels_out || n_images < 1) return KVFE_ERR_INVALID_ARG;
  KParams P = c->P;
  P.B = n_images;
  if (row_stride < (size_t)P.W || (n_images > 1 && image_stride < row_stride * (size_t)(P.H - 1) + P.W))
    return KVFE_ERR_INVALID_ARG;
  size_t per_image = 0;
  for (int l = 1; l < P.nlevels; l++) per_image += (size_t)P.lw[l] * P.lh[l];
  if (levels_capacity < per_image * n_images) return KVFE_ERR_CAPACITY;
  if (n_levels_out) *n_levels_out = P.nlevels;
  if (level_sizes_out)
    for (int l = 0; l < P.nlevels; l++) {
      level_sizes_out[2 * l] = P.lw[l];
      level_sizes_out[2 * l + 1] = P.lh[l];
    }
  hipStream_t st = c->stream;
  // component call (tests, inspection): its own scratch, nothing of the front-end state is touched
  const size_t src_bytes = image_stride * (size_t)(n_images - 1) + row_stride * (size_t)(P.H - 1) + P.W;
  const size_t N = (size_t)P.W * P.H;
  unsigned char *dsrc = nullptr, *dpyr = nullptr, *dcopy = nullptr;
  auto release = [&]() {
    if (dsrc) dev_free(dsrc);
    if (dpyr) dev_free(dpyr);
    if (dcopy) dev_free(dcopy);
  };
  kvfe_status rc = KVFE_OK;
  do {
    if (dev_malloc((void**)&dsrc, src_bytes) != hipSuccess || dev_malloc((void**)&dpyr, (size_t)P.pyr_stride * n_images) != hipSuccess ||
        (level0_copy_out && dev_malloc((void**)&dcopy, N * n_images) != hipSuccess)) {
      c->last_error = "kvfe_build_optical_flow_pyramid: device allocation failed";
      rc = KVFE_ERR_HIP;
      break;
    }
    if (hipMemcpyAsync(dsrc, imgs, src_bytes, hipMemcpyHostToDevice, st) != hipSuccess) { rc = KVFE_ERR_HIP; break; }
    launch_pyramid(P, dsrc, row_stride, image_stride, dpyr, st, dcopy);
    if (hipGetLastError() != hipSuccess) { rc = KVFE_ERR_HIP; break; }
    for (int s = 0; s < n_images && rc == KVFE_OK; s++) {
      size_t off = 0;
      for (int l = 1; l < P.nlevels; l++) {
        const size_t bytes = (size_t)P.lw[l] * P.lh[l];
        if (hipMemcpyAsync(levels_out + per_image * s + off, dpyr + (size_t)P.pyr_stride * s + P.loff[l], bytes,
                           hipMemcpyDeviceToHost, st) != hipSuccess) { rc = KVFE_ERR_HIP; break; }
        off += bytes;
      }
    }
    if (rc == KVFE_OK && level0_copy_out &&
        hipMemcpyAsync(level0_copy_out, dcopy, N * n_images, hipMemcpyDeviceToHost, st) != hipSuccess)
      rc = KVFE_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) rc = KVFE_ERR_HIP;
  } while (0);
  release();
  if (rc == KVFE_ERR_HIP && c->last_error.empty()) c->last_error = "kvfe_build_optical_flow_pyramid: HIP error";
  return rc;
}

kvfe_status kvfe_frontend_debug_pyramid(kvfe_ctx* c, int32_t steps_back, uint8_t* level0_copy_out, uint8_t* levels_out,
                                        size_t levels_capacity) {
  DeviceGuard _dev(c);
  if (!c || steps_back < 0 || steps_back > 1 || !c->children.empty() || !c->fe.pyr[0]) return KVFE_ERR_INVALID_ARG;
  TRY(kvfe_synchronize(c));
  const KParams& P = c->P;
  Buffers& b = c->fe;
  const int slot = c->pyr_cur ^ 1 ^ steps_back;   // (do_step flips pyr_cur when it returns)
  if (levels_out) {   // packed as kvfe_build_optical_flow_pyramid packs them
    size_t per_image = 0;
    for (int l = 1; l < P.nlevels; l++) per_image += (size_t)P.lw[l] * P.lh[l];
    if (levels_capacity < per_image * P.B) return KVFE_ERR_CAPACITY;
    for (int s = 0; s < P.B; s++) {
      size_t off = 0;
      for (int l = 1; l < P.nlevels; l++) {
        const size_t bytes = (size_t)P.lw[l] * P.lh[l];
        HIPCHK(c, hipMemcpy(levels_out + per_image * s + off, b.pyr[slot] + (size_t)P.pyr_stride * s + P.loff[l], bytes,
                            hipMemcpyDeviceToHost));
        off += bytes;
      }
    }
  }
  if (level0_copy_out) {
    if (!b.lvl0[slot]) return KVFE_ERR_UNSUPPORTED;
    HIPCHK(c, hipMemcpy(level0_copy_out, b.lvl0[slot], (size_t)P.W * P.H * P.B, hipMemcpyDeviceToHost));
  }
  return KVFE_OK;
}

kvfe_status kvfe_predict_sparse_flow(kvfe_ctx* c, const float* prev_xy, int32_t n,
                                     const double ref_R_cur[9], float* out_xy) {
  DeviceGuard _dev(c);
  if (!c || !prev_xy || !ref_R_cur || !out_xy || n < 0) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  if (n > c->Pc.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.lk.prev_pts, prev_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.kf_R_cur, ref_R_cur, sizeof(double) * 9, hipMemcpyHostToDevice, st));
  launch_predict_flow(c->Pc, c->T, b.kf_R_cur, b.lk.prev_pts, n, b.lk.next_pts, st);
  HIPCHK(c, hipMemcpyAsync(out_xy, b.lk.next_pts, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_get_right_keypoints_rectified(kvfe_ctx* c, const uint8_t* left_rect,
                                               const uint8_t* right_rect, size_t stride,
                                               const float* left_rect_xy,
                                               const uint8_t* left_status, int32_t n,
                                               float* right_rect_xy, uint8_t* right_status,
                                               double* score) {
  DeviceGuard _dev(c);
  if (!c || !left_rect || !right_rect || !left_rect_xy || !left_status || !right_rect_xy ||
      !right_status || n < 0)
    return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  TRY(upload_image(c, b.rect[0], left_rect, stride));
  TRY(upload_image(c, b.rect[1], right_rect, stride));
  HIPCHK(c, hipMemcpyAsync(b.st.left_rect, left_rect_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.st.left_status, left_status, n, hipMemcpyHostToDevice, st));
  launch_stereo_match_only(P, c->T, b.rect[0], b.rect[1], b.st.left_rect, b.st.left_status, n,
                           b.st.right_rect, b.st.right_status, b.st.depth, st);
  HIPCHK(c, hipMemcpyAsync(right_rect_xy, b.st.right_rect, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(right_status, b.st.right_status, n, hipMemcpyDeviceToHost, st));
  if (score) HIPCHK(c, hipMemcpyAsync(score, b.st.depth, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_sparse_stereo_reconstruction(kvfe_ctx* c, const uint8_t* left_img,
                                              const uint8_t* right_img, size_t stride,
                                              const float* left_xy, int32_t n,
                                              kvfe_stereo_output* out) {
  DeviceGuard _dev(c);
  if (!c || !left_img || !right_img || !left_xy || !out || n < 0) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  const size_t N = (size_t)P.W * P.H;
  TRY(upload_image(c, b.raw_left[0], left_img, stride));
  TRY(upload_image(c, b.raw_right, right_img, stride));
  const FrameTab& K = b.ft[0];
  const int flags = FLAG_STEREO | FLAG_INIT;
  HIPCHK(c, hipMemcpyAsync(b.ss.flags, &flags, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(K.count, &n, sizeof(int), hipMemcpyHostToDevice, st));
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(K.kp, left_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
    launch_undistort_points(c->T.und_left_R, K.kp, n, nullptr, K.versor, st);
  }
  const unsigned char* srcs[2] = {b.raw_left[0], b.raw_right};
  launch_rectify(P, c->T, srcs, P.W, N, b.rect, nullptr, 0, st);
  if (n > 0) launch_stereo(P, c->T, b.rect[0], b.rect[1], K, b.st, b.ss, FLAG_STEREO, n, 0, st);
#define DL(dst, src, bytes) \
  if (dst) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st))
  if (n > 0) {
    DL(out->left_rect_xy, b.st.left_rect, sizeof(float2) * n);
    DL(out->left_status, b.st.left_status, (size_t)n);
    DL(out->right_rect_xy, b.st.right_rect, sizeof(float2) * n);
    DL(out->right_status, b.st.right_status, (size_t)n);
    DL(out->depth, b.st.depth, sizeof(double) * n);
    DL(out->right_xy, b.st.right_kp, sizeof(float2) * n);
    DL(out->keypoints_3d, b.st.kp3d, sizeof(double) * 3 * n);
  }
  DL(out->left_rect_img, b.rect[0], N);
  DL(out->right_rect_img, b.rect[1], N);
#undef DL
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

// ---- UndistorterRectifier / StereoCamera / StereoMatcher keypoint methods on their own ------------------
kvfe_status kvfe_check_undistorted_rectified_left_keypoints(kvfe_ctx* c, int32_t cam, const float* distorted_xy,
                                                            const float* undistorted_xy, int32_t n, float pixel_tol,
                                                            float* out_xy, uint8_t* out_status) {
  DeviceGuard _dev(c);
  if (!c || n < 0 || cam < 0 || cam > 1 || !out_xy || !out_status) return KVFE_ERR_INVALID_ARG;
  if (n > 0 && (!distorted_xy || !undistorted_xy)) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.lk.prev_pts, distorted_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.lk.next_pts, undistorted_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  launch_check_undistorted_rectified(c->T.map[cam], P.W, P.H, b.lk.prev_pts, b.lk.next_pts, n, pixel_tol,
                                     b.st.left_rect, b.st.left_status, st);
  HIPCHK(c, hipMemcpyAsync(out_xy, b.st.left_rect, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(out_status, b.st.left_status, (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_distort_unrectify_keypoints(kvfe_ctx* c, int32_t cam, const float* rect_xy, const uint8_t* status,
                                             int32_t n, float* out_xy) {
  DeviceGuard _dev(c);
  if (!c || n < 0 || cam < 0 || cam > 1 || !out_xy) return KVFE_ERR_INVALID_ARG;
  if (n > 0 && (!rect_xy || !status)) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  // round(px) indexes the map: a VALID keypoint outside the image is a contract violation in the reference
  // (cv::Mat::at asserts in debug builds); checked here on the host copy the caller handed over
  for (int i = 0; i < n; i++)
    if (status[i] == KVFE_KP_VALID) {
      const float x = rect_xy[2 * i], y = rect_xy[2 * i + 1];
      if (!(roundf(x) >= 0.f && roundf(x) <= (float)(P.W - 1) && roundf(y) >= 0.f && roundf(y) <= (float)(P.H - 1)))
        return KVFE_ERR_INVALID_ARG;
    }
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.st.right_rect, rect_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.st.right_status, status, (size_t)n, hipMemcpyHostToDevice, st));
  launch_distort_unrectify(c->T.map[cam], P.W, P.H, b.st.right_rect, b.st.right_status, n, b.st.right_kp, st);
  HIPCHK(c, hipMemcpyAsync(out_xy, b.st.right_kp, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_undistort_rectify_left_keypoints(kvfe_ctx* c, const float* xy, int32_t n, float* out_xy,
                                                  uint8_t* out_status) {
  DeviceGuard _dev(c);
  if (!c || n < 0 || !out_xy || !out_status || (n > 0 && !xy)) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.lk.prev_pts, xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  launch_undistort_points(c->T.und_left_RP, b.lk.prev_pts, n, b.lk.next_pts, nullptr, st);
  launch_check_undistorted_rectified(c->T.map[0], P.W, P.H, b.lk.prev_pts, b.lk.next_pts, n, 2.0f, b.st.left_rect,
                                     b.st.left_status, st);
  HIPCHK(c, hipMemcpyAsync(out_xy, b.st.left_rect, sizeof(float2) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(out_status, b.st.left_status, (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_distort_unrectify_right_keypoints(kvfe_ctx* c, const float* rect_xy, const uint8_t* status,
                                                   int32_t n, float* out_xy) {
  return kvfe_distort_unrectify_keypoints(c, 1, rect_xy, status, n, out_xy);
}

kvfe_status kvfe_undistort_rectify_stereo_frame(kvfe_ctx* c, const uint8_t* left, const uint8_t* right,
                                                size_t src_stride, uint8_t* left_rect, uint8_t* right_rect,
                                                size_t dst_stride) {
  DeviceGuard _dev(c);
  if (!c || !left || !right || !left_rect || !right_rect) return KVFE_ERR_INVALID_ARG;
  if (c->P.mono) return KVFE_ERR_UNSUPPORTED;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (src_stride < (size_t)P.W || dst_stride < (size_t)P.W) return KVFE_ERR_INVALID_ARG;
  hipStream_t st = c->stream;
  TRY(upload_image(c, b.raw_left[0], left, src_stride));
  TRY(upload_image(c, b.raw_right, right, src_stride));
  const unsigned char* srcs[2] = {b.raw_left[0], b.raw_right};
  launch_rectify(P, c->T, srcs, P.W, (size_t)P.W * P.H, b.rect, nullptr, 0, st);
  HIPCHK(c, hipMemcpy2DAsync(left_rect, dst_stride, b.rect[0], P.W, P.W, P.H, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpy2DAsync(right_rect, dst_stride, b.rect[1], P.W, P.W, P.H, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

kvfe_status kvfe_get_depth_from_rectified_matches(kvfe_ctx* c, const float* left_rect_xy, const uint8_t* left_status,
                                                  const float* right_rect_xy, uint8_t* right_status, int32_t n,
                                                  double* depth) {
  DeviceGuard _dev(c);
  if (!c || n < 0 || !depth || !right_status) return KVFE_ERR_INVALID_ARG;
  if (n > 0 && (!left_rect_xy || !left_status || !right_rect_xy)) return KVFE_ERR_INVALID_ARG;
  if (n == 0) return KVFE_OK;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.st.left_rect, left_rect_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.st.left_status, left_status, (size_t)n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.st.right_rect, right_rect_xy, sizeof(float2) * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.st.right_status, right_status, (size_t)n, hipMemcpyHostToDevice, st));
  launch_depth_from_matches(b.st.left_rect, b.st.left_status, b.st.right_rect, b.st.right_status, n,
                            P.fx_rect * P.baseline, P.min_point_dist, P.max_point_dist, b.st.depth, st);
  HIPCHK(c, hipMemcpyAsync(right_status, b.st.right_status, (size_t)n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(depth, b.st.depth, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return KVFE_OK;
}

// ---- Frame-level component calls ---------------------------------------------------------------------------
static kvfe_status frame_upload(kvfe_ctx* c, const FrameTab& F, const kvfe_frame* f, int n, hipStream_t st) {
  HIPCHK(c, hipMemcpyAsync(F.count, &n, sizeof(int), hipMemcpyHostToDevice, st));
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(F.kp, f->keypoints, sizeof(float2) * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(F.lmk, f->landmarks, sizeof(long long) * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(F.age, f->landmarks_age, sizeof(int) * n, hipMemcpyHostToDevice, st));
    if (f->versors) HIPCHK(c, hipMemcpyAsync(F.versor, f->versors, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  }
  return KVFE_OK;
}
static kvfe_status frame_download(kvfe_ctx* c, const FrameTab& F, kvfe_frame* f, hipStream_t st) {
  int n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, F.count, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  f->n_keypoints = n;
  const int m = std::min(n, f->capacity);
  if (m > 0) {
    HIPCHK(c, hipMemcpyAsync(f->keypoints, F.kp, sizeof(float2) * m, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(f->landmarks, F.lmk, sizeof(long long) * m, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(f->landmarks_age, F.age, sizeof(int) * m, hipMemcpyDeviceToHost, st));
    if (f->versors) HIPCHK(c, hipMemcpyAsync(f->versors, F.versor, sizeof(double) * 3 * m, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
  }
  return n > f->capacity ? KVFE_ERR_CAPACITY : KVFE_OK;
}
static bool frame_args_ok(const kvfe_frame* f, bool need_arrays) {
  if (!f || f->capacity < 0 || f->n_keypoints < 0 || f->n_keypoints > f->capacity) return false;
  if ((need_arrays || f->capacity > 0) && (!f->keypoints || !f->landmarks || !f->landmarks_age)) return false;
  return true;
}

kvfe_status kvfe_feature_detection_frame(kvfe_ctx* c, const uint8_t* img, size_t stride, kvfe_frame* frame,
                                         int64_t* landmark_counter) {
  DeviceGuard _dev(c);
  if (!c || !img || !landmark_counter || !frame_args_ok(frame, false) || !frame->versors) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  const int n = frame->n_keypoints;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  TRY(upload_image(c, b.raw_left[0], img, stride));
  const FrameTab& K = b.ft[0];
  TRY(frame_upload(c, K, frame, n, st));
  const int flags = FLAG_DETECT | FLAG_INIT;
  const long long counter = *landmark_counter;
  HIPCHK(c, hipMemcpyAsync(b.ss.flags, &flags, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.ss.n_tracked, &n, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.ss.lmk_counter, &counter, sizeof(long long), hipMemcpyHostToDevice, st));
  // mask discs around the keypoints with a landmark, GFTT, FeatureDetector.cpp:101-115 bookkeeping (ages, need),
  // ANMS, cornerSubPix + append with new landmark ids / age 1 / bearing vectors, counters
  launch_mineig(P, c->T, b.raw_left[0], P.W, (size_t)P.W * P.H, nullptr, K, b.ss, b.ds, 1, st);
  launch_select(P, c->T, K, b.ss, b.ds, -1, st);
  launch_subpix_append(P, c->T, b.raw_left[0], P.W, (size_t)P.W * P.H, K, b.ss, b.ds, 1, st);
  long long counter_out = 0;
  int fl = 0;
  HIPCHK(c, hipMemcpyAsync(&counter_out, b.ss.lmk_counter, sizeof(long long), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(&fl, b.ss.flags, sizeof(int), hipMemcpyDeviceToHost, st));
  const kvfe_status r = frame_download(c, K, frame, st);
  *landmark_counter = counter_out;
  if (fl & FLAG_OVERFLOW) {
    c->last_error = "device candidate / corner list capacity exceeded";
    return KVFE_ERR_CAPACITY;
  }
  return r;
}

kvfe_status kvfe_feature_tracking_frame(kvfe_ctx* c, const uint8_t* ref_img, const uint8_t* cur_img, size_t stride,
                                        kvfe_frame* ref_frame, kvfe_frame* cur_frame, const double ref_R_cur[9]) {
  DeviceGuard _dev(c);
  if (!c || !ref_img || !cur_img || !ref_R_cur || !frame_args_ok(ref_frame, false) || !frame_args_ok(cur_frame, false) ||
      !cur_frame->versors)
    return KVFE_ERR_INVALID_ARG;
  if (cur_frame->n_keypoints != 0) return KVFE_ERR_INVALID_ARG;   // CHECK(cur_frame->keypoints_.empty()), Tracker.cpp:150
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  const int n = ref_frame->n_keypoints;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  const size_t N = (size_t)P.W * P.H;
  TRY(upload_image(c, b.raw_left[0], ref_img, stride));
  TRY(upload_image(c, b.raw_left[1], cur_img, stride));
  launch_pyramid(P, b.raw_left[0], P.W, N, b.pyr[0], st);
  launch_pyramid(P, b.raw_left[1], P.W, N, b.pyr[1], st);
  const FrameTab &KM1 = b.ft[1], &K = b.ft[0], &LKF = b.ft[2];
  TRY(frame_upload(c, KM1, ref_frame, n, st));
  const int zero = 0, flags = FLAG_INIT;
  HIPCHK(c, hipMemcpyAsync(K.count, &zero, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(LKF.count, &zero, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.ss.flags, &flags, sizeof(int), hipMemcpyHostToDevice, st));
  // ref_frame_R_cur_frame = keyframe_R_ref_frame^-1 * keyframe_R_cur_frame with keyframe_R_ref_frame = I
  double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  HIPCHK(c, hipMemcpyAsync(b.ss.kf_R_ref, eye, sizeof(eye), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.kf_R_cur, ref_R_cur, sizeof(double) * 9, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemsetAsync(b.in_ts, 0, sizeof(long long), st));
  HIPCHK(c, hipMemsetAsync(b.in_force, 0, sizeof(int), st));
  b.ss.kf_R_cur = b.kf_R_cur;
  b.ss.in_timestamp = b.in_ts;
  b.ss.in_force_kf = b.in_force;
  launch_track_prepare(P, c->T, KM1, b.ss, b.lk, st);
  launch_lk(P, b.raw_left[0], P.W, N, b.pyr[0], b.raw_left[1], P.W, N, b.pyr[1], b.lk, std::max(n, 1), st);
  launch_track_finalize(P, c->T, KM1, LKF, K, b.ss, b.lk, st);   // survivors -> K (ids, ages, keypoints, versors)
  launch_mark_lost_tracks(P, KM1, b.lk, std::max(n, 1), st);     // ref_frame->landmarks_[i] = -1
  HIPCHK(c, hipStreamSynchronize(st));   // (eye[] is on this stack frame)
  if (n > 0) HIPCHK(c, hipMemcpy(ref_frame->landmarks, KM1.lmk, sizeof(long long) * n, hipMemcpyDeviceToHost));
  return frame_download(c, K, cur_frame, st);
}

static kvfe_status ransac_download(kvfe_ctx* c, Buffers& b, int32_t* inliers, kvfe_ransac_output* out,
                                   bool with_info) {
  DeviceGuard _dev(c);
  hipStream_t st = c->stream;
  int status = 0, cnt[6] = {0};
  HIPCHK(c, hipMemcpyAsync(&status, b.ss.trk_status, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(cnt, b.ss.trk_counts, sizeof(int) * 3, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(out->pose, b.ss.trk_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, st));
  if (with_info)
    HIPCHK(c, hipMemcpyAsync(out->info, b.ss.trk_info, sizeof(double) * 9, hipMemcpyDeviceToHost, st));
  else
    std::memset(out->info, 0, sizeof(out->info));
  HIPCHK(c, hipStreamSynchronize(st));
  out->status = status;
  out->n_inliers = status == KVFE_TRACKING_INVALID ? 0 : cnt[1];
  out->iterations = cnt[2];
  out->reserved0 = 0;
  if (inliers && out->n_inliers > 0)
    HIPCHK(c, hipMemcpy(inliers, b.rs.inliers, sizeof(int) * out->n_inliers, hipMemcpyDeviceToHost));
  return KVFE_OK;
}

kvfe_status kvfe_outlier_rejection_2d2d_given_rotation(kvfe_ctx* c, const double* f_ref,
                                                       const double* f_cur, int32_t n,
                                                       const double R_ref_cur[9], int32_t* inliers,
                                                       kvfe_ransac_output* out) {
  DeviceGuard _dev(c);
  // CHECK_GT(f_ref.size(), 0) (Tracker.cpp:243)
  if (!c || !f_ref || !f_cur || !R_ref_cur || !out || n <= 0) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.rs.f_ref, f_ref, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.rs.f_cur, f_cur, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.kf_R_cur, R_ref_cur, sizeof(double) * 9, hipMemcpyHostToDevice, st));
  launch_ransac_2d2d_points(P, c->T, b.rs.f_ref, b.rs.f_cur, n, b.kf_R_cur, b.rs, b.ss.trk_status,
                            b.ss.trk_pose, b.ss.trk_counts, st);
  return ransac_download(c, b, inliers, out, false);
}

kvfe_status kvfe_outlier_rejection_3d3d_given_rotation(
    kvfe_ctx* c, const float* ref_left_rect_xy, const float* ref_right_rect_x,
    const double* ref_points_3d, const float* cur_left_rect_xy, const float* cur_right_rect_x,
    const double* cur_points_3d, int32_t n, const double R_ref_cur[9], int32_t* inliers,
    kvfe_ransac_output* out) {
  DeviceGuard _dev(c);
  if (!c || !R_ref_cur || !out || n < 0) return KVFE_ERR_INVALID_ARG;
  if (n > 0 && (!ref_left_rect_xy || !ref_right_rect_x || !ref_points_3d || !cur_left_rect_xy ||
                !cur_right_rect_x || !cur_points_3d))
    return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  float* d_ref_left = reinterpret_cast<float*>(b.lk.prev_pts);
  float* d_cur_left = reinterpret_cast<float*>(b.lk.next_pts);
  float* d_ref_rx = b.lk.err;
  float* d_cur_rx = reinterpret_cast<float*>(b.st.depth);
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(d_ref_left, ref_left_rect_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_cur_left, cur_left_rect_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_ref_rx, ref_right_rect_x, sizeof(float) * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(d_cur_rx, cur_right_rect_x, sizeof(float) * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(b.lst.kp3d, ref_points_3d, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(b.st.kp3d, cur_points_3d, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, hipMemcpyAsync(b.kf_R_cur, R_ref_cur, sizeof(double) * 9, hipMemcpyHostToDevice, st));
  launch_ransac_3d3d_points(P, c->T, d_ref_left, d_ref_rx, b.lst.kp3d, d_cur_left, d_cur_rx, b.st.kp3d,
                            n, b.kf_R_cur, b.rs, b.ss.trk_status, b.ss.trk_pose, b.ss.trk_info,
                            b.ss.trk_counts, st);
  return ransac_download(c, b, inliers, out, true);
}

kvfe_status kvfe_outlier_rejection_2d2d(kvfe_ctx* c, const double* f_ref, const double* f_cur, int32_t n,
                                        int32_t* inliers, kvfe_ransac_output* out) {
  DeviceGuard _dev(c);
  // CHECK_GT(f_ref.size(), 0) (Tracker.cpp:243)
  if (!c || !f_ref || !f_cur || !out || n <= 0) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.rs.f_ref, f_ref, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.rs.f_cur, f_cur, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  launch_ransac_2d2d_nister_points(P, c->T, b.rs.f_ref, b.rs.f_cur, n, b.rs, b.ss.trk_status, b.ss.trk_pose,
                                   b.ss.trk_counts, st);
  return ransac_download(c, b, inliers, out, false);
}

kvfe_status kvfe_outlier_rejection_3d3d(kvfe_ctx* c, const double* ref_points_3d, const double* cur_points_3d,
                                        int32_t n, int32_t* inliers, kvfe_ransac_output* out) {
  DeviceGuard _dev(c);
  if (!c || !out || n < 0 || (n > 0 && (!ref_points_3d || !cur_points_3d))) return KVFE_ERR_INVALID_ARG;
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  hipStream_t st = c->stream;
  if (n > 0) {
    HIPCHK(c, hipMemcpyAsync(b.rs.f_ref, ref_points_3d, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(b.rs.f_cur, cur_points_3d, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  }
  launch_ransac_3d3d_arun_points(P, c->T, b.rs.f_ref, b.rs.f_cur, n, b.rs, b.ss.trk_status, b.ss.trk_pose,
                                 b.ss.trk_counts + 1, st);
  return ransac_download(c, b, inliers, out, false);
}

// Tracker::pnp (Tracker.cpp:1122-1288) + outlierRejectionPnP's status (VisionImuFrontend.cpp:146-173)
kvfe_status kvfe_pnp(kvfe_ctx* c, const kvfe_pnp_params* pp, const double* cam_bearing_vectors,
                     const double* F_points, int32_t n, int32_t* inliers, kvfe_ransac_output* out) {
  DeviceGuard _dev(c);
  if (!c || !pp || !out || n < 0 || (n > 0 && (!cam_bearing_vectors || !F_points))) return KVFE_ERR_INVALID_ARG;
  if (pp->pnp_algorithm != 3 && pp->pnp_algorithm != 1) {
    c->last_error = "pnp_algorithm: only 3 (EPNP) and 1 (KneipP3P) are implemented";
    return KVFE_ERR_UNSUPPORTED;
  }
  if (pp->optimize_2d3d_pose_from_inliers) {
    c->last_error = "optimize_2d3d_pose_from_inliers is not implemented (off in every shipped parameter set)";
    return KVFE_ERR_UNSUPPORTED;
  }
  TRY(ensure_comp(c));
  Buffers& b = c->comp;
  const KParams& P = c->Pc;
  if (n > P.kcap) return KVFE_ERR_CAPACITY;
  std::memset(out, 0, sizeof(*out));
  for (int i = 0; i < 12; i++) out->pose[i] = (i % 5 == 0) ? 1.0 : 0.0;
  if (n == 0) {  // "No 2D-3D correspondences found for 2D-3D RANSAC..." (Tracker.cpp:1130-1134)
    out->status = KVFE_TRACKING_FEW_MATCHES;
    return KVFE_OK;
  }
  const double avg_focal_length = 0.5 * (c->cfg.left.intrinsics[0] + c->cfg.left.intrinsics[1]);
  const double threshold = 1.0 - std::cos(std::atan(std::sqrt(2.0) * pp->ransac_threshold_pnp / avg_focal_length));
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(b.rs.f_ref, cam_bearing_vectors, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(b.rs.f_cur, F_points, sizeof(double) * 3 * n, hipMemcpyHostToDevice, st));
  launch_pnp(P, c->T, pp->pnp_algorithm, b.rs.f_ref, b.rs.f_cur, n, threshold, pp->min_pnp_inliers, b.rs.inliers, b.ss.trk_status,
             b.ss.trk_pose, b.ss.trk_counts, st);
  int status = 0, cnt[3] = {0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(&status, b.ss.trk_status, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(cnt, b.ss.trk_counts, sizeof(int) * 3, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(out->pose, b.ss.trk_pose, sizeof(double) * 12, hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  out->status = status;
  out->n_inliers = cnt[0];
  out->iterations = cnt[1];
  out->reserved0 = cnt[2];
  if (inliers && out->n_inliers > 0)
    HIPCHK(c, hipMemcpy(inliers, b.rs.inliers, sizeof(int) * out->n_inliers, hipMemcpyDeviceToHost));
  return KVFE_OK;
}

// ---- front-end level -------------------------------------------------------------------------
kvfe_status kvfe_frontend_step_device(kvfe_ctx* c, const void* left_dev, const void* right_dev,
                                      size_t row_stride, size_t image_stride,
                                      const kvfe_frame_input* inputs) {
  DeviceGuard _dev(c);
  if (!c || !left_dev || !inputs) return KVFE_ERR_INVALID_ARG;
  if (!right_dev) {
    if (!c->P.mono || c->P.rgbd) return KVFE_ERR_INVALID_ARG;
    right_dev = left_dev;  // the mono front-end never reads it
  }
  if (row_stride < (size_t)c->P.W) return KVFE_ERR_INVALID_ARG;
  if (!c->children.empty())
    return step_groups(c, reinterpret_cast<const unsigned char*>(left_dev),
                       reinterpret_cast<const unsigned char*>(right_dev), row_stride, image_stride,
                       inputs, false);
  if (c->cfg.params.stereo.equalize_image) {  // equalised copies live in the ctx-owned slots
    Buffers& b = c->fe;
    const KParams& P = c->P;
    const size_t N = (size_t)P.W * P.H;
    unsigned char* dl = b.raw_left[c->img_step % 3];
    unsigned char* dr = b.raw_right2[c->img_step % 2];
    launch_equalize_hist(P.W, P.H, P.B, reinterpret_cast<const unsigned char*>(left_dev), row_stride,
                         image_stride, dl, b.eq_hist, c->stream);
    launch_equalize_hist(P.W, P.H, P.B, reinterpret_cast<const unsigned char*>(right_dev), row_stride,
                         image_stride, dr, b.eq_hist + 256 * P.B, c->stream);
    c->img_step++;
    c->last_step_staged = false;
    return do_step(c, dl, dr, P.W, N, inputs);
  }
  c->last_step_staged = false;
  if (c->cfg.device_frames_persist) {   // the caller keeps frame k valid until step k+1 has completed: no copy
    c->frames_persist_call = true;
    const kvfe_status r = do_step(c, reinterpret_cast<const unsigned char*>(left_dev),
                                  reinterpret_cast<const unsigned char*>(right_dev), row_stride, image_stride, inputs);
    c->frames_persist_call = false;
    return r;
  }
  // the caller's buffers are only read by THIS step: the left image the next step's LK needs is copied into the
  // context (by the first pyramid launch, which reads it anyway)
  {
    Buffers& b = c->fe;
    const size_t bytes = (size_t)c->P.W * c->P.H * c->P.B;
    for (int i = 0; i < 2; i++)
      if (!b.lvl0[i]) TRY(dalloc(c, &b.lvl0[i], bytes, false));
  }
  // Round 5: with a library-owned stream the step is arranged exactly as with device_frames_persist (corner refinement off
  // the main stream, rectify / match / reject chain on the side stream, joined by the NEXT step's keyframe decision): the
  // contract above says the frames of a step are read until the step has COMPLETED -- kvfe_synchronize /
  // kvfe_frontend_get_output, which wait for both streams -- not until the next step has been enqueued, so nothing has
  // to be awaited in front of the next tracking launch.  (Round 4 awaited the chain there and therefore kept it on the main
  // stream.)  What is left of the option is the copy of the left frame: the next step's tracking reads the context's
  // copy, not the caller's buffer.  A caller-owned stream keeps the stream-ordered arrangement.
  c->own_level0 = true;
  c->frames_persist_call = c->own_stream;
  const kvfe_status st = do_step(c, reinterpret_cast<const unsigned char*>(left_dev),
                                 reinterpret_cast<const unsigned char*>(right_dev), row_stride, image_stride, inputs);
  c->own_level0 = false;
  c->frames_persist_call = false;
  return st;
}

kvfe_status kvfe_frontend_step_host(kvfe_ctx* c, const uint8_t* left, const uint8_t* right,
                                    size_t row_stride, size_t image_stride,
                                    const kvfe_frame_input* inputs) {
  DeviceGuard _dev(c);
  if (!c || !left || !inputs) return KVFE_ERR_INVALID_ARG;
  if (!right) {
    if (!c->P.mono || c->P.rgbd) return KVFE_ERR_INVALID_ARG;
    right = left;
  }
  if (row_stride < (size_t)c->P.W) return KVFE_ERR_INVALID_ARG;
  if (!c->children.empty()) return step_groups(c, left, right, row_stride, image_stride, inputs, true);
  Buffers& b = c->fe;
  const KParams& P = c->P;
  const size_t N = (size_t)P.W * P.H;
  if (P.rgbd) {   // left: colour / intensity image; right: depth image with the same geometry in elements
    const size_t bpp = P.depth_f32 ? 4 : 2;
    unsigned char* dl = b.raw_left[c->img_step % 3];
    unsigned char* dd = b.depth_slot[c->img_step % 2];
    for (int s = 0; s < P.B; s++) {
      HIPCHK(c, hipMemcpy2DAsync(dl + s * N, P.W, left + s * image_stride, row_stride, P.W, P.H,
                                 hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpy2DAsync(dd + s * N * bpp, P.W * bpp, right + s * image_stride * bpp, row_stride * bpp,
                                 P.W * bpp, P.H, hipMemcpyHostToDevice, c->stream));
    }
    c->img_step++;
    c->last_step_staged = false;
    return do_step(c, dl, dd, P.W, N, inputs);
  }
  const bool eq = c->cfg.params.stereo.equalize_image != 0;
  unsigned char* dl = b.raw_left[c->img_step % 3];
  unsigned char* dr = b.raw_right2[c->img_step % 2];
  // equalizeImage: the raw frames are uploaded into the rectified buffers as scratch -- which the previous step's tail
  // (stereo matching of its new corners, on the side stream) may still be reading
  if (eq) join_tail(c);
  unsigned char* ul = eq ? b.rect[0] : dl;  // (rectified buffers are free until rectification runs)
  unsigned char* ur = eq ? b.rect[1] : dr;
  // a few streams: the right image on its own stream (see up2_stream).  Its target slot was last read by the
  // rectification of the frame two steps back, and everything of a step is behind its ring event.
  static const bool up2_env = [] { const char* e = std::getenv("KVFE_RIGHT_UPLOAD_STREAM"); return !e || std::atoi(e) != 0; }();   // (A/B switch)
  hipStream_t rstream = c->stream;
  if (up2_env && P.B <= 4 && !eq && !P.mono && c->own_stream && c->side) {
    if (!c->up2_stream) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->up2_stream, hipStreamNonBlocking));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_up2, hipEventDisableTiming));
    }
    for (int back = 1; back <= 2; back++) {
      const int ps = (c->ring_pos + kvfe_ctx::RING - back) % kvfe_ctx::RING;
      if (c->ring_used[ps]) HIPCHK(c, hipStreamWaitEvent(c->up2_stream, c->ring_ev[ps], 0));
    }
    rstream = c->up2_stream;
  }
  if (row_stride == (size_t)P.W && image_stride == N) {  // tightly packed batch: one copy per side
    HostTimer _t6(6);
    HIPCHK(c, hipMemcpyAsync(ul, left, N * P.B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(ur, right, N * P.B, hipMemcpyHostToDevice, rstream));
  } else if (rstream != c->stream) {
    for (int s = 0; s < P.B; s++)
      HIPCHK(c, hipMemcpy2DAsync(ul + s * N, P.W, left + s * image_stride, row_stride, P.W, P.H,
                                 hipMemcpyHostToDevice, c->stream));
    for (int s = 0; s < P.B; s++)
      HIPCHK(c, hipMemcpy2DAsync(ur + s * N, P.W, right + s * image_stride, row_stride, P.W, P.H,
                                 hipMemcpyHostToDevice, rstream));
  } else {
    for (int s = 0; s < P.B; s++) {
      HIPCHK(c, hipMemcpy2DAsync(ul + s * N, P.W, left + s * image_stride, row_stride, P.W, P.H,
                                 hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpy2DAsync(ur + s * N, P.W, right + s * image_stride, row_stride, P.W, P.H,
                                 hipMemcpyHostToDevice, c->stream));
    }
  }
  if (eq) {
    launch_equalize_hist(P.W, P.H, P.B, ul, P.W, N, dl, b.eq_hist, c->stream);
    launch_equalize_hist(P.W, P.H, P.B, ur, P.W, N, dr, b.eq_hist + 256 * P.B, c->stream);
  }
  c->img_step++;
  c->last_step_staged = false;
  if (rstream != c->stream) {
    HIPCHK(c, hipEventRecord(c->ev_up2, rstream));
    c->right_pending = true;
  }
  c->quiet_check_call = true;
  const kvfe_status r = do_step(c, dl, dr, P.W, N, inputs);
  c->quiet_check_call = false;
  c->right_pending = false;
  return r;
}

// ---- staged input -----------------------------------------------------------------
// A HIP stream is not a hardware queue: the runtime multiplexes a process's streams of one priority onto
// GPU_MAX_HW_QUEUES (default 4) queues, and two streams that share a queue run in order.  A context has five streams
// (main, side, output, upload, top-up) and a process may hold several contexts: when the upload stream lands on the queue
// of a compute stream, the frames' transfer and the step simply add up (2.3 ms instead of 1.4, profiles/r5_analysis.md
// section 4).  Queues are pooled per priority, so the upload stream asks for the other pool.
static hipError_t create_stream_in_other_pool(hipStream_t* s) {
  int lo = 0, hi = 0;   // (numerically: greatest priority <= least)
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
  return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
}
static kvfe_status ensure_staging(kvfe_ctx* c, int slot) {
  DeviceGuard _dev(c);
  const size_t bytes = 2 * (size_t)c->P.W * c->P.H * c->P.B;
  if (!c->staging_ready) {
    // one-time resources; a call that fails half way (say, no memory for the input ring) leaves what it created in place
    // and the next call creates only what is missing -- nothing reads them before staging_ready is set
    if (!c->copy_stream) HIPCHK(c, create_stream_in_other_pool(&c->copy_stream));
    for (int i = 0; i < 4; i++)
      if (!c->step_done[i]) HIPCHK(c, hipEventCreateWithFlags(&c->step_done[i], hipEventDisableTiming));
    for (int i = 0; i < 4; i++)
      if (!c->ev_chain[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_chain[i], hipEventDisableTiming));
    if (!c->cfg.copy_inputs && !c->fork_swap) {   // many streams: device copies of the input ring slots (do_step)
      for (int i = 0; i < kvfe_ctx::RING; i++)
        if (!c->in_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->in_ev[i], hipEventDisableTiming));
      if (!c->in_dev_staging) {
        c->ring_bytes_dev = (c->ring_bytes + 255) & ~(size_t)255;
        TRY(dalloc(c, &c->in_dev_staging, c->ring_bytes_dev * kvfe_ctx::RING));
      }
    }
    if (c->cfg.params.stereo.equalize_image)
      for (int i = 0; i < 2; i++)
        if (!c->fe.eq_in[i]) TRY(dalloc(c, &c->fe.eq_in[i], (size_t)c->P.W * c->P.H * c->P.B, false));
    c->in_dev = c->in_dev_staging;   // (do_step takes the copy path once this is set: everything it needs exists)
    c->staging_ready = true;
  }
  if (!c->stage_host[slot]) {   // a slot's pinned memory on first use
    void* h = nullptr;
    HIPCHK(c, hipHostMalloc(&h, bytes, hipHostMallocDefault));
    c->host_allocs.push_back(h);
    c->stage_host[slot] = reinterpret_cast<unsigned char*>(h);
    HIPCHK(c, hipEventCreateWithFlags(&c->stage_copied[slot], hipEventDisableTiming));
  }
  return KVFE_OK;
}

kvfe_status kvfe_frontend_staging_buffer(kvfe_ctx* c, int32_t slot, uint8_t** left, uint8_t** right) {
  DeviceGuard _dev(c);
  if (!c || !left || !right || slot < 0 || slot >= KVFE_STAGING_SLOTS) return KVFE_ERR_INVALID_ARG;
  if (!c->children.empty()) return KVFE_ERR_UNSUPPORTED;  // staged input drives one stream group
  if (c->P.rgbd) return KVFE_ERR_UNSUPPORTED;             // (depth images: use the host / device step)
  TRY(ensure_staging(c, slot));
  *left = c->stage_host[slot];
  *right = c->stage_host[slot] + (size_t)c->P.W * c->P.H * c->P.B;
  return KVFE_OK;
}

kvfe_status kvfe_frontend_staging_wait(kvfe_ctx* c, int32_t slot) {
  DeviceGuard _dev(c);
  if (!c || slot < 0 || slot >= KVFE_STAGING_SLOTS) return KVFE_ERR_INVALID_ARG;
  if (c->stage_pending[slot]) {
    HIPCHK(c, hipEventSynchronize(c->stage_copied[slot]));
    c->stage_pending[slot] = false;
  }
  return KVFE_OK;
}

kvfe_status kvfe_frontend_step_staged(kvfe_ctx* c, int32_t slot, const kvfe_frame_input* inputs) {
  DeviceGuard _dev(c);
  if (!c || !inputs || slot < 0 || slot >= KVFE_STAGING_SLOTS) return KVFE_ERR_INVALID_ARG;
  if (!c->children.empty()) return KVFE_ERR_UNSUPPORTED;
  TRY(ensure_staging(c, slot));
  Buffers& b = c->fe;
  const KParams& P = c->P;
  const size_t N = (size_t)P.W * P.H, NB = N * P.B;
  const bool eq = c->cfg.params.stereo.equalize_image != 0;
  const long long n = c->img_step;
  unsigned char* dl = b.raw_left[n % 3];
  unsigned char* dr = b.raw_right2[n % 2];
  // the upload may start as soon as the last readers of its target are done: left slot n%3 was
  // frame n-3 (read by steps n-3 and n-2), right slot n%2 was frame n-2; the equalise inputs are
  // free once step n-1 has consumed them -> wait for step n-2 (n-1 with equalisation)
  const long long dep = eq ? n - 1 : n - 2;
  if (n > 0 && !c->last_step_staged) {
    // the previous step went through another entry point: order the upload after everything enqueued
    join_tail(c);
    HIPCHK(c, hipEventRecord(c->step_done[(n - 1) % 4], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->step_done[(n - 1) % 4], 0));
    c->step_done_valid[(n - 1) % 4] = true;
  } else if (dep >= 0 && c->step_done_valid[dep % 4]) {
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->step_done[dep % 4], 0));
    // the side stream's chain (the other reader of a frame's slots) is not covered by the main stream's step_done: the
    // upload of frame n rewrites the right slot of frame n-2 and the left slot of frame n-3 -- the chain of step n-2 is
    // behind both on the side stream (and behind the refinement of step n-3, which its predecessor's tail awaited)
    if (c->chain_pending) {
      if (c->staging_ready && c->chain_seq >= 2 && c->last_step_staged)
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_chain[(c->chain_seq - 2) % 4], 0));
      else if (!c->staging_ready || !c->last_step_staged)
        HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_main, 0));
    }
  }
  unsigned char* ul = eq ? b.eq_in[0] : dl;
  unsigned char* ur = eq ? b.eq_in[1] : dr;
  HIPCHK(c, hipMemcpyAsync(ul, c->stage_host[slot], NB, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(c, hipMemcpyAsync(ur, c->stage_host[slot] + NB, NB, hipMemcpyHostToDevice, c->copy_stream));
  HIPCHK(c, hipEventRecord(c->stage_copied[slot], c->copy_stream));
  c->stage_pending[slot] = true;
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->stage_copied[slot], 0));
  if (eq) {
    launch_equalize_hist(P.W, P.H, P.B, ul, P.W, N, dl, b.eq_hist, c->stream);
    launch_equalize_hist(P.W, P.H, P.B, ur, P.W, N, dr, b.eq_hist + 256 * P.B, c->stream);
  }
  c->img_step++;
  {
    // Many streams: every kernel of this step on the main stream (no fork) -- the next slot's upload is in flight while the
    // step runs, and with a DMA transfer in flight the cross-stream hand-overs of the forked step complete late (round 4,
    // tools/r4/staged_probe.py, 64 streams, 30 steps: 3.1 ms per step forked, 2.0 ms in order; the upload itself is
    // 0.86 ms).  A few streams (fork_swap: every kernel is a latency, the upload is a few hundred KB) keep the fork that
    // was tuned for them.
    c->serial_call = !c->fork_swap;
    c->inputs_by_copy_call = true;
    const kvfe_status r = do_step(c, dl, dr, P.W, N, inputs);
    c->serial_call = false;
    c->inputs_by_copy_call = false;
    if (r != KVFE_OK) return r;
  }
  HIPCHK(c, hipEventRecord(c->step_done[n % 4], c->stream));
  c->step_done_valid[n % 4] = true;
  c->last_step_staged = true;
  return KVFE_OK;
}

kvfe_status kvfe_frontend_update_map(kvfe_ctx* c, int32_t stream, const int64_t* landmark_ids, const double* xyz,
                                     int32_t n) {
  DeviceGuard _dev(c);
  join_tail(c);
  if (!c || stream < 0 || stream >= c->P.B || n < 0 || (n > 0 && (!landmark_ids || !xyz))) return KVFE_ERR_INVALID_ARG;
  for (kvfe_ctx* ch : c->children)
    if (stream >= ch->s0 && stream < ch->s0 + ch->P.B)
      return kvfe_frontend_update_map(ch, stream - ch->s0, landmark_ids, xyz, n);
  const KParams& P = c->P;
  if (n > P.map_cap) return KVFE_ERR_CAPACITY;
  // std::unordered_map semantics: one position per id (the last one given wins), looked up by id -> sorted for the
  // device's binary search
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return landmark_ids[a] < landmark_ids[b]; });
  std::vector<long long> ids;
  std::vector<double> pts;
  ids.reserve(n);
  pts.reserve(3 * (size_t)n);
  for (int k = 0; k < n; k++) {
    const int i = order[k];
    if (!ids.empty() && ids.back() == (long long)landmark_ids[i]) {   // duplicate id: keep the later entry
      for (int q = 0; q < 3; q++) pts[pts.size() - 3 + q] = xyz[3 * (size_t)i + q];
      continue;
    }
    ids.push_back((long long)landmark_ids[i]);
    for (int q = 0; q < 3; q++) pts.push_back(xyz[3 * (size_t)i + q]);
  }
  const int m = (int)ids.size();
  Buffers& b = c->fe;
  hipStream_t st = c->stream;
  if (m > 0) {
    HIPCHK(c, hipMemcpyAsync(b.ss.map_ids + (size_t)stream * P.map_cap, ids.data(), sizeof(long long) * m,
                             hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(b.ss.map_xyz + (size_t)stream * P.map_cap * 3, pts.data(), sizeof(double) * 3 * m,
                             hipMemcpyHostToDevice, st));
  }
  HIPCHK(c, hipMemcpyAsync(b.ss.map_n + stream, &m, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipStreamSynchronize(st));   // the host vectors go out of scope
  return KVFE_OK;
}

kvfe_status kvfe_frontend_reset(kvfe_ctx* c) {
  DeviceGuard _dev(c);
  join_tail(c);
  if (!c) return KVFE_ERR_INVALID_ARG;
  if (!c->children.empty()) {
    for (kvfe_ctx* ch : c->children) {
      TRY(kvfe_frontend_reset(ch));
      ch->ev_tracked_valid = false;
    }
    return KVFE_OK;
  }
  Buffers& b = c->fe;
  const size_t B = c->P.B;
  hipStream_t st = c->stream;
  HIPCHK(c, hipStreamSynchronize(st));
  HIPCHK(c, hipMemsetAsync(b.ss.flags, 0, sizeof(int) * B, st));
  HIPCHK(c, hipMemsetAsync(b.ss.lmk_counter, 0, sizeof(long long) * B, st));
  HIPCHK(c, hipMemsetAsync(b.ss.frame_count, 0, sizeof(long long) * B, st));
  for (int i = 0; i < 3; i++) HIPCHK(c, hipMemsetAsync(b.ft[i].count, 0, sizeof(int) * B, st));
  TRY(reset_tracker_status(c, b));
  std::vector<double> eye(B * 9, 0.0);
  for (size_t s = 0; s < B; s++) eye[s * 9] = eye[s * 9 + 4] = eye[s * 9 + 8] = 1.0;
  HIPCHK(c, hipMemcpy(b.ss.kf_R_ref, eye.data(), sizeof(double) * B * 9, hipMemcpyHostToDevice));
  HIPCHK(c, hipStreamSynchronize(st));
  c->prev_left = nullptr;
  c->img_step = 0;
  c->pyr_cur = 0;
  if (c->out_stream) HIPCHK(c, hipStreamSynchronize(c->out_stream));
  c->out_steps = 0;
  c->last_step_staged = false;
  for (int i = 0; i < 4; i++) c->step_done_valid[i] = false;
  c->role_k = 0;
  c->role_km1 = 1;
  c->role_lkf = 2;
  return KVFE_OK;
}

// The packed record of stream `s` written by the step `steps_back` steps before the latest one, in its pinned ring slot
// (kvfe_dev.hpp "output side"): one event wait for THAT step's transfer.  No device call touches the context's streams,
// so the step that is running meanwhile is not disturbed.
static kvfe_status locate_output(kvfe_ctx* c, int32_t s, int32_t steps_back, const unsigned char** rec_out) {
  if (c->out_steps <= steps_back) {
    c->last_error = "kvfe_frontend_get_output: no step has produced that output yet";
    return KVFE_ERR_INVALID_ARG;
  }
  const int slot = (int)((c->out_steps - 1 - steps_back) % OUT_RING);
  {
    HostTimer _t7(7);
    HIPCHK(c, hipEventSynchronize(c->ev_out[slot]));
  }
  if (c->prof_stride > 0 && steps_back == 0) {   // stage events of the profiled steps: collected once everything is complete
    join_tail(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
  }
  const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(c->out_host[slot]);
  const size_t end = (size_t)tab[c->P.B];
  if (end < c->out_tab_bytes || end > c->out_slot_size) {
    c->last_error = "kvfe_frontend_get_output: the output slot holds no valid record table";
    return KVFE_ERR_HIP;
  }
  if (end > c->out_copied[slot]) {   // the step needed more than its transfer was sized for: fetch the rest now
    // On a stream of its own: the output stream already holds the transfers of LATER steps, each waiting for its
    // step's records, so a top-up enqueued there would wait for the running step.  The slot's records
    // are complete (ev_out above follows the event of their gather) and its staging buffer is not rewritten before
    // OUT_RING more steps have been enqueued by this same thread.
    const size_t from = c->out_copied[slot];
    if (!c->topup_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->topup_stream, hipStreamNonBlocking));
    HIPCHK(c, hipMemcpyAsync(c->out_host[slot] + from, c->out_stage[slot] + from, end - from, hipMemcpyDeviceToHost,
                             c->topup_stream));
    HIPCHK(c, hipStreamSynchronize(c->topup_stream));
    c->out_copied[slot] = end;
    c->out_topups++;
  }
  *rec_out = c->out_host[slot] + (size_t)tab[s];
  return KVFE_OK;
}

// header fields of a record -> kvfe_frame_output; returns the record's layout
static OutLayout output_header(int rec_cap, const unsigned char* rec, kvfe_frame_output* out, int* n_rec, int* m_rec) {
  const OutHeader* h = reinterpret_cast<const OutHeader*>(rec);
  out->n_keypoints = h->n_keypoints;
  out->is_keyframe = (h->flags & FLAG_KEYFRAME) ? 1 : 0;
  out->n_tracked = h->n_tracked;
  out->n_detected = h->n_detected;
  out->n_measurements = h->n_meas;
  out->frame_id = h->frame_count - 1;
  out->tracking_status_mono = h->trk_status[0];
  out->tracking_status_stereo = h->trk_status[1];
  std::memcpy(out->lkf_T_k_mono, h->trk_pose, sizeof(double) * 12);
  std::memcpy(out->lkf_T_k_stereo, h->trk_pose + 12, sizeof(double) * 12);
  std::memcpy(out->info_mat_stereo_translation, h->trk_info, sizeof(double) * 9);
  out->nr_mono_putatives = h->trk_counts[0];
  out->nr_mono_inliers = h->trk_counts[1];
  out->mono_ransac_iters = h->trk_counts[2];
  out->nr_stereo_putatives = h->trk_counts[3];
  out->nr_stereo_inliers = h->trk_counts[4];
  out->reserved0 = 0;
  out->tracking_status_pnp = h->pnp_status;
  out->nr_pnp_inliers = h->pnp_counts[0];
  std::memcpy(out->W_T_k_pnp, h->pnp_pose, sizeof(double) * 12);
  *n_rec = std::min(h->n_keypoints, rec_cap);   // entries the record holds
  *m_rec = std::min(h->n_meas, rec_cap);
  return out_layout(*n_rec, *m_rec, (h->flags & FLAG_STEREO) != 0);
}

static kvfe_status copy_output_record(int rec_cap, const unsigned char* rec, kvfe_frame_output* out);

kvfe_status kvfe_frontend_get_output_at(kvfe_ctx* c, int32_t s, int32_t steps_back, kvfe_frame_output* out) {
  DeviceGuard _dev(c);
  if (!c || !out || s < 0 || s >= c->P.B || out->capacity < 0 || steps_back < 0 || steps_back >= OUT_RING)
    return KVFE_ERR_INVALID_ARG;
  for (kvfe_ctx* ch : c->children)
    if (s >= ch->s0 && s < ch->s0 + ch->P.B) {
      const kvfe_status r = kvfe_frontend_get_output_at(ch, s - ch->s0, steps_back, out);
      if (r != KVFE_OK) c->last_error = ch->last_error;
      return r;
    }
  const unsigned char* rec = nullptr;
  TRY(locate_output(c, s, steps_back, &rec));
  const kvfe_status r = copy_output_record(c->out_cap, rec, out);
  if (r == KVFE_ERR_CAPACITY)
    c->last_error = "a device-side list overflowed its capacity (candidates / corners / keypoints)";
  return r;
}

// record in a pinned ring slot -> the caller's kvfe_frame_output (no device call, no context state: safe from any thread)
static kvfe_status copy_output_record(int rec_cap, const unsigned char* rec, kvfe_frame_output* out) {
  if (out->capacity < 0) return KVFE_ERR_INVALID_ARG;
  int np = 0, mp = 0;
  const OutLayout L = output_header(rec_cap, rec, out, &np, &mp);
  const int flags = reinterpret_cast<const OutHeader*>(rec)->flags;
  const bool stereo = (flags & FLAG_STEREO) != 0;
  const int n = std::min(np, out->capacity);
  const int m = std::min(mp, out->capacity);
#define DL(dst, off, bytes) \
  if (dst && (bytes) > 0) std::memcpy(dst, rec + (off), bytes)
  DL(out->landmarks, L.lmk, sizeof(long long) * n);
  DL(out->landmarks_age, L.age, sizeof(int) * n);
  DL(out->keypoints, L.kp, sizeof(float2) * n);
  DL(out->versors, L.versor, sizeof(double) * 3 * n);
  if (stereo) {
    DL(out->left_rect_xy, L.left_rect, sizeof(float2) * n);
    DL(out->left_status, L.left_status, (size_t)n);
    DL(out->right_rect_xy, L.right_rect, sizeof(float2) * n);
    DL(out->right_status, L.right_status, (size_t)n);
    DL(out->depth, L.depth, sizeof(double) * n);
    DL(out->right_xy, L.right_kp, sizeof(float2) * n);
    DL(out->keypoints_3d, L.kp3d, sizeof(double) * 3 * n);
  }
  DL(out->meas_landmark, L.meas_lmk, sizeof(long long) * m);
  DL(out->meas_uL_uR_v, L.meas, sizeof(double) * 3 * m);
#undef DL
  return (flags & FLAG_OVERFLOW) ? KVFE_ERR_CAPACITY : KVFE_OK;
}

// Zero-copy variant: the array pointers of `out` are set to the arrays INSIDE the pinned record (stereo arrays NULL on a
// frame without stereo data); they stay valid until KVFE_OUTPUT_RING - 1 - steps_back further steps have been enqueued.
kvfe_status kvfe_frontend_view_output(kvfe_ctx* c, int32_t s, int32_t steps_back, kvfe_frame_output* out) {
  DeviceGuard _dev(c);
  if (!c || !out || s < 0 || s >= c->P.B || steps_back < 0 || steps_back >= OUT_RING) return KVFE_ERR_INVALID_ARG;
  for (kvfe_ctx* ch : c->children)
    if (s >= ch->s0 && s < ch->s0 + ch->P.B) {
      const kvfe_status r = kvfe_frontend_view_output(ch, s - ch->s0, steps_back, out);
      if (r != KVFE_OK) c->last_error = ch->last_error;
      return r;
    }
  const unsigned char* rec = nullptr;
  TRY(locate_output(c, s, steps_back, &rec));
  int np = 0, mp = 0;
  const OutLayout L = output_header(c->out_cap, rec, out, &np, &mp);
  const int flags = reinterpret_cast<const OutHeader*>(rec)->flags;
  const bool stereo = (flags & FLAG_STEREO) != 0;
  unsigned char* r = const_cast<unsigned char*>(rec);
  out->capacity = np;
  out->landmarks = reinterpret_cast<int64_t*>(r + L.lmk);
  out->landmarks_age = reinterpret_cast<int32_t*>(r + L.age);
  out->keypoints = reinterpret_cast<float*>(r + L.kp);
  out->versors = reinterpret_cast<double*>(r + L.versor);
  out->left_rect_xy = stereo ? reinterpret_cast<float*>(r + L.left_rect) : nullptr;
  out->left_status = stereo ? r + L.left_status : nullptr;
  out->right_rect_xy = stereo ? reinterpret_cast<float*>(r + L.right_rect) : nullptr;
  out->right_status = stereo ? r + L.right_status : nullptr;
  out->depth = stereo ? reinterpret_cast<double*>(r + L.depth) : nullptr;
  out->right_xy = stereo ? reinterpret_cast<float*>(r + L.right_kp) : nullptr;
  out->keypoints_3d = stereo ? reinterpret_cast<double*>(r + L.kp3d) : nullptr;
  out->meas_landmark = reinterpret_cast<int64_t*>(r + L.meas_lmk);
  out->meas_uL_uR_v = reinterpret_cast<double*>(r + L.meas);
  if (flags & FLAG_OVERFLOW) {
    c->last_error = "a device-side list overflowed its capacity (candidates / corners / keypoints)";
    return KVFE_ERR_CAPACITY;
  }
  return KVFE_OK;
}

// All streams of the context at once: outs[s] as for kvfe_frontend_get_output_at (one call, one event wait).
kvfe_status kvfe_frontend_get_outputs(kvfe_ctx* c, int32_t steps_back, kvfe_frame_output* outs) {
  if (!c || !outs) return KVFE_ERR_INVALID_ARG;
  const int B = c->P.B;
  // 64 records of ~77 KB are a millisecond of memcpy for one core -- as long as the whole step takes on the device --
  // so batches of 16 streams and more are copied by up to four threads (the first call waits for the transfer's event
  // once; the others find it complete)
  const int nthr = B >= 16 ? std::min(4, B / 8) : 1;
  if (nthr <= 1 || !c->children.empty()) {
    kvfe_status worst = KVFE_OK;
    for (int s = 0; s < B; s++) {
      const kvfe_status r = kvfe_frontend_get_output_at(c, s, steps_back, outs + s);
      if (r == KVFE_ERR_CAPACITY) worst = r;
      else if (r != KVFE_OK) return r;
    }
    return worst;
  }
  kvfe_status first = kvfe_frontend_get_output_at(c, 0, steps_back, outs);
  if (first != KVFE_OK && first != KVFE_ERR_CAPACITY) return first;
  std::vector<kvfe_status> res((size_t)nthr, KVFE_OK);
  std::vector<std::thread> pool;
  const int slot = (int)((c->out_steps - 1 - steps_back) % OUT_RING);
  const unsigned char* base = c->out_host[slot];
  const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(base);   // (complete: the first call saw to it)
  auto work = [&](int t) {
    for (int s = 1 + t; s < B; s += nthr) {
      const kvfe_status r = copy_output_record(c->out_cap, base + (size_t)tab[s], outs + s);
      if (r != KVFE_OK) res[(size_t)t] = r;
    }
  };
  for (int t = 1; t < nthr; t++) {
    try {
      pool.emplace_back(work, t);
    } catch (const std::system_error&) {   // no thread to be had: this one does the share itself
      work(t);
    }
  }
  work(0);
  for (auto& th : pool) th.join();
  kvfe_status worst = first;
  for (kvfe_status r : res)
    if (r == KVFE_ERR_CAPACITY) worst = r;
    else if (r != KVFE_OK) return r;
  if (worst == KVFE_ERR_CAPACITY)
    c->last_error = "a device-side list overflowed its capacity (candidates / corners / keypoints)";
  return worst;
}

kvfe_status kvfe_frontend_get_output(kvfe_ctx* c, int32_t s, kvfe_frame_output* out) {
  return kvfe_frontend_get_output_at(c, s, 0, out);
}

// ---- dense stereo --------------------------------------------------------
void kvfe_dense_stereo_params_default(kvfe_dense_stereo_params* p) {
  if (!p) return;
  // DenseStereoParams member initialisers, StereoMatchingParams.h:40-58
  *p = kvfe_dense_stereo_params{};
  p->use_sgbm = 1;
  p->pre_filter_cap = 31;
  p->sad_window_size = 11;
  p->min_disparity = 1;
  p->num_disparities = 64;
  p->uniqueness_ratio = 0;
  p->speckle_range = 3;
  p->speckle_window_size = 500;
  p->texture_threshold = 0;
  p->pre_filter_type = 1;   // cv::StereoBM::PREFILTER_XSOBEL
  p->pre_filter_size = 9;
  p->p1 = 120;
  p->p2 = 240;
  p->disp_12_max_diff = -1;
  p->use_mode_hh = 1;
}

// cv::StereoSGBM's own defaulting of its parameters (stereosgbm.cpp computeDisparitySGBM prologue)
static kvfe_status dense_params(kvfe_ctx* c, const kvfe_dense_stereo_params& dp, DenseParams* out) {
  auto unsupported = [&](const char* why) {
    c->last_error = std::string("dense stereo: ") + why;
    return KVFE_ERR_UNSUPPORTED;
  };
  if (dp.num_disparities <= 0 || dp.num_disparities % 16 != 0) return KVFE_ERR_INVALID_ARG;
  if (dp.num_disparities > 64) return unsupported("num_disparities > 64");
  DenseParams P{};
  P.W = c->P.W;
  P.H = c->P.H;
  if (P.W > 2048) return unsupported("image width > 2048");
  P.minD = dp.min_disparity;
  P.D = dp.num_disparities;
  if (!dp.use_sgbm) {   // cv::StereoBM::create(numDisparities, blockSize) + setters (StereoMatcher.cpp:66-90)
    // argument checks of StereoBMImpl::compute
    if (dp.pre_filter_type != 0 && dp.pre_filter_type != 1) return KVFE_ERR_INVALID_ARG;
    if (dp.pre_filter_type != 1) return unsupported("StereoBM PREFILTER_NORMALIZED_RESPONSE");
    if (dp.pre_filter_cap < 1 || dp.pre_filter_cap > 63) return KVFE_ERR_INVALID_ARG;
    const int wsz = dp.sad_window_size;
    if (wsz < 5 || wsz > 255 || wsz % 2 == 0 || wsz >= std::min(P.W, P.H)) return KVFE_ERR_INVALID_ARG;
    if (dp.texture_threshold < 0 || dp.uniqueness_ratio < 0) return KVFE_ERR_INVALID_ARG;
    if (dp.disp_12_max_diff >= 0) return unsupported("StereoBM disp12MaxDiff >= 0 (validateDisparity)");
    if ((long long)wsz * wsz * 2 * dp.pre_filter_cap > 32767) return unsupported("StereoBM window SAD above 16 bits");
    P.bm = 1;
    P.full_dp = 1;
    P.minX1 = std::max(P.D - 1 + P.minD, 0);            // lofs
    P.bm_rofs = -std::min(P.D - 1 + P.minD, 0);         // rofs
    P.width1 = P.W - P.bm_rofs - P.D + 1;
    if (P.minX1 >= P.W || P.bm_rofs >= P.W || P.width1 < 1) P.width1 = 0;   // everything FILTERED
    P.SW2 = wsz / 2;
    P.ftzero = dp.pre_filter_cap;
    P.uniq = dp.uniqueness_ratio;
    P.bm_texture = dp.texture_threshold;
    P.invalid_scaled = (P.minD - 1) * 16;
    P.speckle_win = dp.speckle_range >= 0 ? dp.speckle_window_size : 0;
    P.speckle_diff = dp.speckle_range;                  // (cv::StereoBM passes the range unscaled)
    P.median5 = dp.median_blur_disparity ? 1 : 0;
    // cv::getValidDisparityROI(roi1, roi2, minDisparity, numDisparities, blockSize) with the ROIs of the
    // rectification when both are non-empty (StereoMatcher.cpp:81-86), else the whole image
    const int32_t* r1 = c->rect.roi1;
    const int32_t* r2 = c->rect.roi2;
    const bool have = r1[2] > 0 && r1[3] > 0 && r2[2] > 0 && r2[3] > 0;
    const int a[4] = {have ? r1[0] : 0, have ? r1[1] : 0, have ? r1[2] : P.W, have ? r1[3] : P.H};
    const int b[4] = {have ? r2[0] : 0, have ? r2[1] : 0, have ? r2[2] : P.W, have ? r2[3] : P.H};
    const int maxD1 = P.minD + P.D - 1;
    const int xmin = std::max(a[0], b[0] + maxD1) + P.SW2, xmax = std::min(a[0] + a[2], b[0] + b[2] - P.minD) - P.SW2;
    const int ymin = std::max(a[1], b[1]) + P.SW2, ymax = std::min(a[1] + a[3], b[1] + b[3]) - P.SW2;
    int roi[4] = {xmin, ymin, xmax - xmin, ymax - ymin};
    if (roi[2] <= 0 || roi[3] <= 0) roi[0] = roi[1] = roi[2] = roi[3] = 0;
    // intersect with the image
    const int x0 = std::max(roi[0], 0), y0 = std::max(roi[1], 0);
    const int x1 = std::min(roi[0] + roi[2], P.W), y1 = std::min(roi[1] + roi[3], P.H);
    P.bm_roi[0] = x0;
    P.bm_roi[1] = y0;
    P.bm_roi[2] = std::max(x1 - x0, 0);
    P.bm_roi[3] = std::max(y1 - y0, 0);
    if (P.bm_roi[2] == 0 || P.bm_roi[3] == 0) P.bm_roi[2] = P.bm_roi[3] = 0;
    *out = P;
    return KVFE_OK;
  }
  const int maxD = P.minD + P.D;
  P.minX1 = std::max(maxD, 0);
  const int maxX1 = P.W + std::min(P.minD, 0);
  P.width1 = maxX1 - P.minX1;
  if (P.minD < 0) return unsupported("min_disparity < 0");
  const int bs = dp.sad_window_size > 0 ? dp.sad_window_size : 5;
  P.SW2 = bs / 2;
  P.ftzero = std::max(dp.pre_filter_cap, 15) | 1;
  P.uniq = dp.uniqueness_ratio >= 0 ? dp.uniqueness_ratio : 10;
  P.disp12 = dp.disp_12_max_diff > 0 ? dp.disp_12_max_diff : 1;
  P.P1 = dp.p1 > 0 ? dp.p1 : 2;
  P.P2 = std::max(dp.p2 > 0 ? dp.p2 : 5, P.P1 + 1);
  P.invalid_scaled = (P.minD - 1) * 16;
  P.speckle_win = dp.speckle_window_size;
  P.speckle_diff = 16 * dp.speckle_range;
  P.median5 = dp.median_blur_disparity ? 1 : 0;
  P.full_dp = dp.use_mode_hh ? 1 : 0;
  // 16-bit cost arithmetic: one path cost <= window * (2*ftzero + 63) + 2*P2, four of them per u16 sum
  const long long win = (long long)(2 * P.SW2 + 1) * (2 * P.SW2 + 1);
  if (win * (2 * P.ftzero + 63) + 2LL * P.P2 > 16383)
    return unsupported("sad_window_size / p2 leave the 16-bit cost range (OpenCV's CostType is short)");
  *out = P;
  return KVFE_OK;
}

static kvfe_status dense_ensure(kvfe_ctx* c, const DenseParams& P, int pairs) {
  DeviceGuard _dev(c);
  DenseBuffers& b = c->dense;
  const size_t ve = P.width1 > 0 ? dense_volume_elems(P) : 1;
  const size_t hand_need = dense_handoff_bytes(P, pairs);
  if (b.cap_pairs >= pairs && b.vol_elems == ve && b.hand_bytes >= dense_handoff_bytes(P, b.cap_pairs))
    return KVFE_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (void* p : c->dense_allocs) dev_free(p);
  c->dense_allocs.clear();
  b = DenseBuffers{};
  auto al = [&](void** p, size_t bytes) -> kvfe_status {
    HIPCHK(c, dev_malloc(p, std::max<size_t>(bytes, 16)));
    c->dense_allocs.push_back(*p);
    return KVFE_OK;
  };
  const size_t px = (size_t)P.W * P.H;
  TRY(al((void**)&b.left, px * pairs));
  TRY(al((void**)&b.right, px * pairs));
  TRY(al((void**)&b.rec, sizeof(uint2) * px * 2 * pairs));
  // (a guard band either side: the two-pass aggregation requests C a few columns past the ends of a row)
  constexpr size_t kVolGuard = 1024;   // elements
  for (int i = 0; i < 3; i++) {
    TRY(al((void**)&b.vol[i], sizeof(short) * (ve * pairs + 2 * kVolGuard)));
    b.vol[i] += kVolGuard;
  }
  for (int i = 0; i < 2; i++) TRY(al((void**)&b.disp[i], sizeof(short) * px * pairs));
  TRY(al((void**)&b.label, sizeof(int) * px * pairs));
  TRY(al((void**)&b.count, sizeof(int) * px * pairs));
  TRY(al((void**)&b.runlen, sizeof(int) * px * pairs));
  TRY(al((void**)&c->dense_dispf, sizeof(float) * px));
  TRY(al((void**)&c->dense_xyz, sizeof(float) * 3 * px));
  TRY(al((void**)&c->dense_minkey, 16));
  TRY(al((void**)&b.agsync, AGP_SYNC_WORDS * sizeof(unsigned)));
  // (the error word is read after every call.  On the context's stream: the streams of the library are non-blocking, a fill
  // on the null stream is not ordered in front of their work -- found with poisoned allocations, KVFE_GUARD_ALLOC=3)
  HIPCHK(c, hipMemsetAsync(b.agsync, 0, AGP_SYNC_WORDS * sizeof(unsigned), c->stream));
  if (hand_need) {
    TRY(al((void**)&b.hand, hand_need));
    b.hand_bytes = hand_need;
  }
  b.cap_pairs = pairs;
  b.vol_elems = ve;
  for (int i = 0; i < 2; i++)
    if (!c->dense_ev[i]) HIPCHK(c, hipEventCreate(&c->dense_ev[i]));
  return KVFE_OK;
}

kvfe_status kvfe_dense_stereo_reconstruction(kvfe_ctx* c, const kvfe_dense_stereo_params* params,
                                             int32_t n_pairs, const uint8_t* const* left_rect,
                                             const uint8_t* const* right_rect, size_t stride,
                                             int16_t* const* disparity, size_t dstride) {
  DeviceGuard _dev(c);
  if (!c || !params || n_pairs < 0 || (n_pairs > 0 && (!left_rect || !right_rect || !disparity)))
    return KVFE_ERR_INVALID_ARG;
  DenseParams P;
  TRY(dense_params(c, *params, &P));
  if (stride < (size_t)P.W || dstride < (size_t)P.W) return KVFE_ERR_INVALID_ARG;
  for (int i = 0; i < n_pairs; i++)
    if (!left_rect[i] || !right_rect[i] || !disparity[i]) return KVFE_ERR_INVALID_ARG;
  if (n_pairs == 0) return KVFE_OK;
  if (P.width1 <= 0) {   // computeDisparitySGBM: minX1 >= maxX1 -> all INVALID_DISP_SCALED (the 3x3 median and
                         // the speckle filter leave a constant image unchanged)
    for (int i = 0; i < n_pairs; i++)
      for (int y = 0; y < P.H; y++)
        for (int x = 0; x < P.W; x++) disparity[i][(size_t)y * dstride + x] = (int16_t)P.invalid_scaled;
    return KVFE_OK;
  }
  const int chunk = std::min<int>(8, n_pairs);   // pairs per launch group (three cost volumes of 42 MB per pair)
  TRY(dense_ensure(c, P, chunk));
  DenseBuffers& b = c->dense;
  const size_t px = (size_t)P.W * P.H;
  for (int i0 = 0; i0 < n_pairs; i0 += chunk) {
    const int n = std::min(chunk, n_pairs - i0);
    for (int i = 0; i < n; i++) {
      HIPCHK(c, hipMemcpy2DAsync(b.left + px * i, P.W, left_rect[i0 + i], stride, P.W, P.H, hipMemcpyHostToDevice,
                                 c->stream));
      HIPCHK(c, hipMemcpy2DAsync(b.right + px * i, P.W, right_rect[i0 + i], stride, P.W, P.H,
                                 hipMemcpyHostToDevice, c->stream));
    }
    // The two-pass launch's waves wait for each other with BOUNDED polls (a preempted or debugged process must not hang the
    // chip); a wait that ran out sets an error word and the launch's result is garbage.  The chunk is then repeated on the
    // eight independent sweeps, which wait for nothing -- same result, 0.69 instead of 0.27 ms per pair --
    // and the context counts it (kvfe_last_error names it, the call succeeds).  KVFE_DENSE_FORCE_FALLBACK=1 takes that path
    // on every chunk (tests/test_gpu_dense_twopass.py).
    static const bool force_fallback = [] { const char* e = std::getenv("KVFE_DENSE_FORCE_FALLBACK"); return e && std::atoi(e) != 0; }();
    for (int attempt = 0; attempt < 2; attempt++) {
      HIPCHK(c, hipEventRecord(c->dense_ev[0], c->stream));
      if (P.bm)
        launch_dense_bm(P, b, n, c->stream);
      else
        launch_dense_sgbm(P, b, n, c->stream, attempt == 0);
      HIPCHK(c, hipEventRecord(c->dense_ev[1], c->stream));
      HIPCHK(c, hipGetLastError());
      unsigned agg_err = 0;
      if (attempt == 0 && !P.bm) {
        HIPCHK(c, hipMemcpyAsync(&agg_err, b.agsync + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (force_fallback) agg_err = 1;
      }
      if (!agg_err) break;
      c->dense_fallbacks++;
      c->last_error = "dense stereo: a hand-over wait of the two-pass aggregation ran out; the chunk was repeated on the direction sweeps";
      HIPCHK(c, hipMemsetAsync(b.agsync + 1, 0, sizeof(unsigned), c->stream));
    }
    for (int i = 0; i < n; i++)
      HIPCHK(c, hipMemcpy2DAsync(disparity[i0 + i], dstride * sizeof(int16_t), b.disp[0] + px * i,
                                 P.W * sizeof(int16_t), P.W * sizeof(int16_t), P.H, hipMemcpyDeviceToHost,
                                 c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->dense_ev[0], c->dense_ev[1]) == hipSuccess) {
      c->dense_ms += ms;
      c->dense_pairs += n;
    }
  }
  return KVFE_OK;
}

// debug / test hook: the cost volumes of the FIRST pair of the last kvfe_dense_stereo_reconstruction
// call, [H][width1][D] int16: which = 0 / 1 partial sums of the path costs (u16; MODE_HH: four directions each), 2 = C(p,d)
kvfe_status kvfe_dense_debug_volume(kvfe_ctx* c, int32_t which, int16_t* out, size_t elems) {
  DeviceGuard _dev(c);
  if (!c || !out || which < 0 || which > 2 || !c->dense.vol[which] || elems > c->dense.vol_elems)
    return KVFE_ERR_INVALID_ARG;
  HIPCHK(c, hipMemcpy(out, c->dense.vol[which], elems * sizeof(int16_t), hipMemcpyDeviceToHost));
  return KVFE_OK;
}

kvfe_status kvfe_dense_profile_read(kvfe_ctx* c, double* kernel_ms, int64_t* pairs) {
  DeviceGuard _dev(c);
  if (!c) return KVFE_ERR_INVALID_ARG;
  if (kernel_ms) *kernel_ms = c->dense_ms;
  if (pairs) *pairs = c->dense_pairs;
  c->dense_ms = 0;
  c->dense_pairs = 0;
  return KVFE_OK;
}

kvfe_status kvfe_backproject_disparity_to_3d(kvfe_ctx* c, const float* disparity, size_t stride, float* xyz) {
  DeviceGuard _dev(c);
  if (!c || !disparity || !xyz || stride < (size_t)c->P.W) return KVFE_ERR_INVALID_ARG;
  // StereoCamera::backProjectDisparityTo3D CHECKs Q(3,2) != 0 and Q(3,3) == 0 (StereoCamera.cpp:188-190)
  if (c->rect.Q[14] == 0.0 || c->rect.Q[15] != 0.0) return KVFE_ERR_INVALID_ARG;
  DenseParams P{};
  P.W = c->P.W;
  P.H = c->P.H;
  P.D = 16;
  P.width1 = 1;
  if (!c->dense_dispf) TRY(dense_ensure(c, P, 1));
  ReprojectQ Q;
  for (int i = 0; i < 16; i++) Q.q[i] = c->rect.Q[i];
  HIPCHK(c, hipMemcpy2DAsync(c->dense_dispf, P.W * sizeof(float), disparity, stride * sizeof(float),
                             P.W * sizeof(float), P.H, hipMemcpyHostToDevice, c->stream));
  launch_reproject_to_3d(P.W, P.H, c->dense_dispf, Q, c->dense_minkey, c->dense_xyz, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(xyz, c->dense_xyz, sizeof(float) * 3 * (size_t)P.W * P.H, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return KVFE_OK;
}

kvfe_status kvfe_profile_enable(kvfe_ctx* c, int32_t on) {
  DeviceGuard _dev(c);
  join_tail(c);
  if (!c) return KVFE_ERR_INVALID_ARG;
  for (kvfe_ctx* ch : c->children) TRY(kvfe_profile_enable(ch, on));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  c->prof_stride = on > 0 ? on : 0;  // on = N: every N-th step records its stage events
  c->prof_step = 0;
  c->prof_on = false;
  if (on) {
    for (int i = 0; i < ST_COUNT; i++) {
      c->prof_ms[i] = c->prof_ms_active[i] = c->prof_active_streams[i] = 0;
      c->prof_active_launches[i] = 0;
    }
    c->prof_samples = 0;
    if (!c->prof_flags_host && c->children.empty()) {
      void* h = nullptr;
      if (hipHostMalloc(&h, sizeof(int) * (size_t)PROF_FLAG_SAMPLES * c->P.B, hipHostMallocDefault) == hipSuccess) {
        c->host_allocs.push_back(h);
        c->prof_flags_host = reinterpret_cast<int*>(h);
      }
    }
  }
  return KVFE_OK;
}

static kvfe_status profile_read_full(kvfe_ctx* c, kvfe_stage_times* out);

kvfe_status kvfe_profile_read(kvfe_ctx* c, kvfe_stage_times* user) {
  if (!c || !user) return KVFE_ERR_INVALID_ARG;
  // the caller's struct may be shorter than this build's (kvfe_stage_times::struct_size): fill a full one, copy what fits
  size_t n = user->struct_size > 0 ? (size_t)user->struct_size : offsetof(kvfe_stage_times, ms_active);
  n = std::min(n, sizeof(kvfe_stage_times));
  if (n < offsetof(kvfe_stage_times, name)) return KVFE_ERR_INVALID_ARG;
  kvfe_stage_times full;
  TRY(profile_read_full(c, &full));
  full.struct_size = (int32_t)n;
  std::memcpy(user, &full, n);
  return KVFE_OK;
}

static kvfe_status profile_read_full(kvfe_ctx* c, kvfe_stage_times* out) {
  DeviceGuard _dev(c);
  join_tail(c);
  if (!c || !out) return KVFE_ERR_INVALID_ARG;
  if (!c->children.empty()) {  // launches of all groups: summed durations, mean bytes per launch
    std::memset(out, 0, sizeof(*out));
    out->n_stages = ST_COUNT;
    out->n_groups = (int)c->children.size();
    for (kvfe_ctx* ch : c->children) {
      kvfe_stage_times t;
      TRY(profile_read_full(ch, &t));
      out->n_samples += t.n_samples;
      for (int s = 0; s < ST_COUNT; s++) {
        out->name[s] = t.name[s];
        out->ms_total[s] += t.ms_total[s];
        out->alg_bytes[s] += t.alg_bytes[s] * t.n_samples;
        out->ms_active[s] += t.ms_active[s];
        out->active_streams[s] += t.active_streams[s];
        out->active_launches[s] += t.active_launches[s];
        out->alg_bytes_per_stream[s] = t.alg_bytes_per_stream[s];
      }
    }
    for (int s = 0; s < ST_COUNT; s++) out->alg_bytes[s] /= std::max(out->n_samples, 1);
    return KVFE_OK;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  std::memset(out, 0, sizeof(*out));
  out->n_stages = ST_COUNT;
  out->n_groups = 1;
  out->n_samples = c->prof_samples;
  const double N = (double)c->P.W * c->P.H * c->P.B;
  for (int s = 0; s < ST_COUNT; s++) {
    out->name[s] = kStageNames[s];
    out->ms_total[s] = c->prof_ms[s];
    out->ms_active[s] = c->prof_ms_active[s];
    out->active_streams[s] = c->prof_active_streams[s];
    out->active_launches[s] = c->prof_active_launches[s];
  }
  // algorithmic bytes per launch of the dense kernels (DESIGN.md "roofline accounting")
  double pyr_out = 0;
  for (int l = 1; l < c->P.nlevels; l++) pyr_out += (double)c->P.lw[l] * c->P.lh[l];
  out->alg_bytes[ST_PYRAMID] = N + pyr_out * c->P.B;              // read level 0, write levels 1..L
  out->alg_bytes[ST_MINEIG] = N;                                  // read the left image once
  out->alg_bytes[ST_RECTIFY] = 4.0 * N;                           // read L,R raw + write L,R rectified
  for (int s = 0; s < ST_COUNT; s++) out->alg_bytes_per_stream[s] = out->alg_bytes[s] / c->P.B;
  return KVFE_OK;
}

}  // extern "C"
