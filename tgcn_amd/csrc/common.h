// common.h -- error reporting, launch timing, tuning switches and small host helpers shared by every section
// Part of the single translation unit tgcn_hip.hip (included once, inside its anonymous namespace).
#pragma once


thread_local char g_err[512] = "";

#define TGCN_FAIL(code, ...)                    \
  do {                                          \
    snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
    return (code);                              \
  } while (0)

#define TGCN_CHECK_LAUNCH(what)                                                         \
  do {                                                                                  \
    hipError_t e_ = hipGetLastError();                                                  \
    if (e_ != hipSuccess) TGCN_FAIL(TGCN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

constexpr int kBlock = 256;

// ---- optional launch timing (bench / tests): hipEvent pairs recorded around launches on their own stream.
// State of the CALLING THREAD (round 6: the ABI holds no process-global mutable state): a thread that asked for a
// record gets the launches IT issues; the replicas' threads of an nn.DataParallel process neither see nor disturb it.
struct ProfRec { hipEvent_t a, b; int kind; };
thread_local std::vector<ProfRec> g_prof;
thread_local int g_prof_cap = 0;

struct ProfScope {
  hipEvent_t b = nullptr;
  hipStream_t st;
  ProfScope(int kind, hipStream_t s) : st(s) {
    if (g_prof_cap <= 0) return;
    if ((int)g_prof.size() >= g_prof_cap) return;
    ProfRec r;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    r.kind = kind;
    (void)hipEventRecord(r.a, st);
    b = r.b;
    g_prof.push_back(r);
  }
  ~ProfScope() { if (b) (void)hipEventRecord(b, st); }
};

// Kernels that take more than 64 KB of dynamic LDS.  The attribute belongs to the calling thread's current device
// (nn.DataParallel drives several devices from one process), so it is set once per (device, kernel).
inline void allow_large_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert(std::make_pair(dev, fn)).second)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);   // a refusal surfaces as a launch error
}

// Compute units of the calling thread's current device (256 on MI355X), read once per device: round sizes of one-workgroup-per-CU
// kernels follow from it instead of a literal.
// Dynamic LDS a workgroup of the calling thread's current device can opt in to (163,840 B on gfx950; 65,536 on gfx942), read once per device:
// kernels that keep a whole weight resident (project_x3_stream_kernel: up to 150 KB) are only chosen where it fits, the tiled forms otherwise.
inline int lds_optin_limit() {
  static std::atomic<int> cached[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 64 * 1024;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess || v <= 0) {
    (void)hipGetLastError();
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
  }
  cached[dev].store(v, std::memory_order_relaxed);
  return v;
}

inline int cu_count() {
  static std::atomic<int> cached[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cached[dev].store(v, std::memory_order_relaxed);
  return v;
}

// ---- developer switches (tgcn_set_tuning): state of the CALLING THREAD, like the launch timing above.  A switch set by a test or a tool
// selects another kernel for the same arithmetic in the launches that thread issues afterwards; other threads (DataParallel replicas,
// the autograd engine's workers) keep the defaults, which are what ships.
struct TuneInt {
  int v;
  int load() const { return v; }
  void store(int x) { v = x; }
};
thread_local TuneInt g_hop_variant{0};
thread_local TuneInt g_hop_remap{1};        // hop_kernel: row blocks in XCD-contiguous ranges (1) or round robin over the XCDs (0)
thread_local TuneInt g_hop_seg_remap{0};    // hop_kernel: segment blocks in XCD-contiguous ranges (1) or round robin over the XCDs (0)
thread_local TuneInt g_hop_mix{0};          // hop_kernel: row blocks dealt evenly among the segment blocks (1) or all in front (0)
thread_local TuneInt g_hop_stream{1};       // hop_kernel: non-temporal entries / stores / partial rows when the output exceeds the Infinity Cache (0: never)
thread_local TuneInt g_hop_lds_pad{0};      // hop_kernel: bytes of unused dynamic LDS per workgroup (occupancy limiter, developer A/B)
thread_local TuneInt g_proj_variant{0};     // 1: force the streaming-W kernel
thread_local TuneInt g_overlap{0};          // layer driver: projection of pass i on a side stream under the hops of pass i+1
thread_local TuneInt g_x3_form{2};          // bf16x3 projection, aligned operands, >= 96 output columns: 2 = A fragments from registers (+20 %), 1 = both operands through LDS
thread_local TuneInt g_compact_proj{0};     // compacted forward: 0 two row-mapped projections (compact rows, empty rows), 1 one projection over all vertices in order
thread_local TuneInt g_fuse_last{0};        // compacted forward: 1 = last hop's short rows gathered inside the projection (project_x3_gather_kernel), bitwise the same
                                            // result; measured SLOWER (cfg5 290 -> 329 ms: the gathers want the hop kernel's occupancy), so 0 = hop + projection ships
thread_local TuneInt g_x3_tail{1};          // project_x3v2_kernel: rows of a thinly filled last round as 128-row tiles (0: 256-row tiles throughout)
thread_local TuneInt g_small_narrow{1};     // C <= 4 inputs of the one-launch path: input-side recursion (0: output-side kernel)
thread_local TuneInt g_small_dense{2};      // small dense operands: 2 bf16x3 matrix pipe, 1 fp32 matrix pipe, 0 vector-ALU kernels only

struct SideStream { hipStream_t st = nullptr; hipEvent_t hops_done[2] = {nullptr, nullptr}; hipEvent_t proj_done[2] = {nullptr, nullptr}; };
std::mutex g_side_mu;
SideStream g_side[16];

// One helper stream + 4 events per device, created on first use and kept for the life of the process.
SideStream* side_stream() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream& s = g_side[dev];
  if (!s.st) {
    if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess) { s.st = nullptr; return nullptr; }
    for (int i = 0; i < 2; ++i) {
      if (hipEventCreateWithFlags(&s.hops_done[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&s.proj_done[i], hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(s.st);
        s.st = nullptr;
        return nullptr;
      }
    }
  }
  return &s;
}


inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Device of a caller's data pointer against the calling thread's current device.  The stream a caller hands in belongs to its current
// device, and every launch goes there: a module moved to cuda:1 and called while cuda:0 is current would run on the wrong GPU (the
// reference's torch ops follow the TENSOR's device, tgcn/nn/gcn.py:141,147; SURVEY.md 8b "honour ... the device of the pointers").
// One hipPointerGetAttributes per layer / hop call; skipped while the stream is being captured into a hipGraph (the capture was started
// on that device by construction).
inline int check_pointer_device(const void* p, hipStream_t st, const char* who) {
  if (!p) return TGCN_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return TGCN_OK; }
  if (cs != hipStreamCaptureStatusNone) return TGCN_OK;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return TGCN_OK;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return TGCN_OK; }    // not a pointer the runtime knows: nothing to say
  if (at.type == hipMemoryTypeDevice && at.device != cur)
    TGCN_FAIL(TGCN_ERR_INVALID, "%s: the data is on device %d but the calling thread's current device is %d -- make the tensor's device current "
                                "(hipSetDevice / torch.cuda.device) and pass that device's stream", who, at.device, cur);
  return TGCN_OK;
}
