// orbx_multi.cpp — the sharded form of the hot path for a C / C++ host: one process, several MI355X, one
// orbx_ctx per device.  Frames (and consecutive frame pairs) are independent units, so a batch is cut into contiguous even
// blocks, one per device (pairs never straddle devices), every device runs the fused extract + match call on its block, and
// the only exchange is an ncclAllGather of the per-frame keypoint counts over xGMI.  No data-path collective.
//
// RCCL is loaded at run time (dlopen("librccl.so")), and only when a multi-device context is created: a single-GPU user of
// liborbx.so neither links nor loads it.  Every RCCL failure maps to ORBX_E_RCCL.  (bench.py and the Python tests use
// torch.distributed over the same RCCL with one process per GPU; this file gives a C++ host -- which is what the reference
// is, demo/demo_initialization.cpp:65-113 -- the same thing without Python.)
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbx.h"
#include "orbx_knobs.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL entry points)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;   // ncclSuccess == 0
const int kNcclInt32 = 2;   // ncclInt32
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string* err) {
    if (lib) return true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { *err = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found"); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd) { *err = "librccl.so lacks an expected symbol"; return false; }
    return true;
  }
};
Rccl g_rccl;

}  // namespace

// One issuing host thread per device: issuing a batch costs the host ~75 us per device (a dozen launches, pair list, events), which
// one thread would pay n times in a row for every batch.  A worker owns its device's context while a command runs.
struct Cmd {
  int kind = 0;  // 0 = issue a block, 1 = wait for the oldest block in flight, 2 = stop
  int nb = 0;
  const uint8_t* img = nullptr;
  int width = 0, height = 0, stride = 0;
  size_t frameStride = 0;
  orbx_keypoint* kps = nullptr;
  uint8_t* desc = nullptr;
  int capacity = 0;
  int32_t* nOut = nullptr;
  orbx_bounds b{};
  int window = 0, checkOri = 0;
  float nnratio = 0;
  int32_t* m12 = nullptr;
  int32_t* nm = nullptr;
};
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Cmd> q;
  long issued = 0, done = 0;  // commands pushed / completed
  int rc = ORBX_OK;           // result of the last completed command
  std::string err;
};
struct Batch {  // a batch in flight
  int nFrames = 0;
  std::vector<int> lo, hi;
  std::vector<int32_t*> dN;  // per device: the caller's count array of its block
  int32_t* countsAll = nullptr;
};

struct orbx_multi {
  int n = 0;
  std::vector<int> dev;
  std::vector<orbx_ctx*> ctx;
  std::vector<ncclComm_t> comm;       // useRccl only
  bool useRccl = false;               // n > 1, or ORBX_MULTI_FORCE_RCCL=1: a one-rank communicator on one device (the RCCL code
                                      // paths -- dlopen, ncclCommInitAll, grouped ncclAllGather, destroy -- on a one-GPU box)
  int32_t* hAll = nullptr;            // pinned [padCap * n]: device 0's copy of the gathered counts lands here
  std::vector<hipStream_t> st;        // one stream per device for the collective
  std::vector<int32_t*> dSend, dRecv; // per device: counts of its block (padded), counts of all blocks
  int padCap = 0;                     // entries per block in the gather buffers
  int depth = 0;                      // orbx_multi_set_pipeline_depth
  std::vector<Worker*> workers;
  std::deque<Batch> inflight;
  std::string err;
};

namespace {

void workerMain(orbx_multi* m, int r) {
  Worker& w = *m->workers[r];
  (void)hipSetDevice(m->dev[r]);
  for (;;) {
    Cmd c;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      w.cv.wait(lk, [&] { return !w.q.empty(); });
      c = w.q.front();
      w.q.pop_front();
    }
    int rc = ORBX_OK;
    std::string err;
    if (c.kind == 2) {
      std::lock_guard<std::mutex> lk(w.mu);
      w.done++;
      w.cv.notify_all();
      return;
    }
    if (c.kind == 0) {
      std::vector<int32_t> first, second;  // consecutive pairs (2k, 2k + 1) of the block (the context keeps its own copy)
      for (int p = 0; p + 1 < c.nb; p += 2) { first.push_back(p); second.push_back(p + 1); }
      rc = orbx_extract_match_batch_device_async(m->ctx[r], c.nb, c.img, c.width, c.height, c.stride, c.frameStride, c.kps, c.desc,
                                                 c.capacity, c.nOut, (int)first.size(), first.data(), second.data(), &c.b, c.window,
                                                 c.nnratio, c.checkOri, c.m12, c.nm, nullptr);
    } else {
      rc = orbx_wait_one(m->ctx[r]);
    }
    if (rc != ORBX_OK) err = std::string("device block: ") + orbx_last_error(m->ctx[r]);
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.rc = rc;
      w.err = err;
      w.done++;
    }
    w.cv.notify_all();
  }
}

void push(Worker& w, const Cmd& c) {
  {
    std::lock_guard<std::mutex> lk(w.mu);
    w.q.push_back(c);
    w.issued++;
  }
  w.cv.notify_all();
}
// waits until the worker has completed everything pushed so far; returns the last command's result
int drain(Worker& w, std::string* err) {
  std::unique_lock<std::mutex> lk(w.mu);
  w.cv.wait(lk, [&] { return w.done == w.issued; });
  if (w.rc != ORBX_OK && err) *err = w.err;
  return w.rc;
}

// counts of the oldest batch -> padded send buffers (-1 beyond the block), all-gather on the collective streams (the contexts'
// own streams keep running the next batches), back to the host from device 0's copy
int gatherCounts(orbx_multi* m, const Batch& B) {
  const int n = m->n, per = m->padCap;
  for (int r = 0; r < n; r++) {
    if (hipSetDevice(m->dev[r]) != hipSuccess) return ORBX_E_HIP;
    const int nb = B.hi[r] - B.lo[r];
    if (hipMemsetAsync(m->dSend[r], 0xff, sizeof(int32_t) * (size_t)per, m->st[r]) != hipSuccess) return ORBX_E_HIP;
    if (nb > 0 && hipMemcpyAsync(m->dSend[r], B.dN[r], sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToDevice, m->st[r]) != hipSuccess)
      return ORBX_E_HIP;
  }
  if (m->useRccl) {
    if (g_rccl.GroupStart() != 0) return ORBX_E_RCCL;
    for (int r = 0; r < n; r++) {
      const ncclResult_t q = g_rccl.AllGather(m->dSend[r], m->dRecv[r], (size_t)per, kNcclInt32, m->comm[r], m->st[r]);
      if (q != 0) {
        (void)g_rccl.GroupEnd();
        m->err = std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(q) : "error");
        return ORBX_E_RCCL;
      }
    }
    if (g_rccl.GroupEnd() != 0) return ORBX_E_RCCL;
  } else {
    if (hipMemcpyAsync(m->dRecv[0], m->dSend[0], sizeof(int32_t) * (size_t)per, hipMemcpyDeviceToDevice, m->st[0]) != hipSuccess)
      return ORBX_E_HIP;
  }
  // device 0's copy comes back into page-locked memory (an asynchronous copy command: nothing above has been waited for yet);
  // only then are the devices' collective streams synchronised, one after the other -- all of them have been running meanwhile
  if (hipSetDevice(m->dev[0]) != hipSuccess) return ORBX_E_HIP;
  if (hipMemcpyAsync(m->hAll, m->dRecv[0], sizeof(int32_t) * (size_t)per * n, hipMemcpyDeviceToHost, m->st[0]) != hipSuccess) return ORBX_E_HIP;
  for (int r = 0; r < n; r++) {
    if (hipSetDevice(m->dev[r]) != hipSuccess) return ORBX_E_HIP;
    if (hipStreamSynchronize(m->st[r]) != hipSuccess) return ORBX_E_HIP;  // every device's copy of the counts is complete
  }
  for (int r = 0; r < n; r++)
    for (int f = B.lo[r]; f < B.hi[r]; f++) B.countsAll[f] = m->hAll[(size_t)r * per + (f - B.lo[r])];
  return ORBX_OK;
}

}  // namespace

extern "C" {

int orbx_multi_wait_one(orbx_multi* m);
int orbx_multi_wait(orbx_multi* m);

int orbx_multi_shard_range(int n_frames, int n_devices, int r, int* lo, int* hi) {
  if (n_frames < 0 || n_devices < 1 || r < 0 || r >= n_devices || !lo || !hi) return ORBX_E_BADARG;
  const int pairs = (n_frames + 1) / 2, per = (pairs + n_devices - 1) / n_devices;
  const long long a = (long long)r * per * 2, b = (long long)(r + 1) * per * 2;
  *lo = (int)(a < n_frames ? a : n_frames);
  *hi = (int)(b < n_frames ? b : n_frames);
  return ORBX_OK;
}

void orbx_multi_destroy(orbx_multi* m) {
  if (!m) return;
  for (Worker* w : m->workers) {
    if (!w) continue;
    if (w->th.joinable()) {
      Cmd c;
      c.kind = 2;
      push(*w, c);
      w->th.join();
    }
    delete w;
  }
  m->workers.clear();
  for (int r = 0; r < m->n; r++) {
    if (r < (int)m->dev.size()) (void)hipSetDevice(m->dev[r]);
    if (r < (int)m->comm.size() && m->comm[r] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[r]);
    if (r < (int)m->dSend.size() && m->dSend[r]) (void)hipFree(m->dSend[r]);
    if (r < (int)m->dRecv.size() && m->dRecv[r]) (void)hipFree(m->dRecv[r]);
    if (r < (int)m->st.size() && m->st[r]) (void)hipStreamDestroy(m->st[r]);
    if (r < (int)m->ctx.size() && m->ctx[r]) orbx_destroy(m->ctx[r]);
  }
  if (m->hAll) (void)hipHostFree(m->hAll);
  delete m;
}

int orbx_multi_create(const orbx_params* params, int n_devices, const int* device_ids, int max_width, int max_height,
                      int max_batch_per_device, orbx_multi** out) {
  if (!params || !out || n_devices < 1 || n_devices > 64 || !device_ids || max_batch_per_device < 1) return ORBX_E_BADARG;
  *out = nullptr;
  for (int a = 0; a < n_devices; a++)
    for (int b = a + 1; b < n_devices; b++)
      if (device_ids[a] == device_ids[b]) return ORBX_E_BADARG;  // one context per device; a communicator cannot hold a device twice
  orbx_multi* m = new orbx_multi();
  m->n = n_devices;
  m->dev.assign(device_ids, device_ids + n_devices);
  m->ctx.assign(n_devices, nullptr);
  m->st.assign(n_devices, nullptr);
  m->dSend.assign(n_devices, nullptr);
  m->dRecv.assign(n_devices, nullptr);
  m->padCap = (max_batch_per_device + 1) & ~1;
  for (int r = 0; r < n_devices; r++) {
    const int rc = orbx_create(params, device_ids[r], max_width, max_height, max_batch_per_device, nullptr, &m->ctx[r]);
    if (rc != ORBX_OK) { orbx_multi_destroy(m); return rc; }
    if (hipSetDevice(device_ids[r]) != hipSuccess || hipStreamCreateWithFlags(&m->st[r], hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&m->dSend[r], sizeof(int32_t) * (size_t)m->padCap) != hipSuccess ||
        hipMalloc((void**)&m->dRecv[r], sizeof(int32_t) * (size_t)m->padCap * n_devices) != hipSuccess) {
      orbx_multi_destroy(m);
      return ORBX_E_HIP;
    }
  }
  if (hipHostMalloc((void**)&m->hAll, sizeof(int32_t) * (size_t)m->padCap * n_devices, hipHostMallocDefault) != hipSuccess) {
    orbx_multi_destroy(m);
    return ORBX_E_HIP;
  }
  {
    m->useRccl = n_devices > 1 || orbx::knobOn(orbx::KNOB_MULTI_FORCE_RCCL);  // (orbx_debug_set "multi_force_rccl": tests)
  }
  if (m->useRccl) {
    std::string e;
    if (!g_rccl.load(&e)) { orbx_multi_destroy(m); return ORBX_E_RCCL; }
    m->comm.assign(n_devices, nullptr);
    const ncclResult_t q = g_rccl.CommInitAll(m->comm.data(), n_devices, device_ids);
    if (q != 0) { orbx_multi_destroy(m); return ORBX_E_RCCL; }
  }
  m->workers.assign(n_devices, nullptr);
  for (int r = 0; r < n_devices; r++) {
    m->workers[r] = new Worker();
    m->workers[r]->th = std::thread(workerMain, m, r);
  }
  *out = m;
  return ORBX_OK;
}

int orbx_multi_size(const orbx_multi* m) { return m ? m->n : ORBX_E_BADARG; }
orbx_ctx* orbx_multi_ctx(orbx_multi* m, int r) { return (m && r >= 0 && r < m->n) ? m->ctx[r] : nullptr; }
const char* orbx_multi_last_error(const orbx_multi* m) { return m ? m->err.c_str() : "null context"; }

// Lanes per device context (orbx_set_pipeline_depth on every device): how many batches may be in flight.
int orbx_multi_set_pipeline_depth(orbx_multi* m, int depth) {
  if (!m || depth < 0 || depth > 8) return ORBX_E_BADARG;
  const int w = orbx_multi_wait(m);
  if (w != ORBX_OK) return w;
  for (int r = 0; r < m->n; r++) {
    const int rc = orbx_set_pipeline_depth(m->ctx[r], depth);
    if (rc != ORBX_OK) { m->err = std::string("device context: ") + orbx_last_error(m->ctx[r]); return rc; }
  }
  m->depth = depth;
  return ORBX_OK;
}

// Stream-ordered form: every device's worker thread issues its block (frames already resident in its HBM; consecutive pairs
// (2k, 2k + 1) matched) and the call returns once all blocks are queued.  d_*[r] are device r's arrays for ITS block, laid out
// as for orbx_extract_match_batch_device; blocks = orbx_multi_shard_range(n_frames, n, r).
int orbx_multi_extract_match_batch_device_async(orbx_multi* m, int n_frames, const uint8_t* const* d_imgs, int width, int height,
                                                int stride, size_t frame_stride_bytes, orbx_keypoint* const* d_kps,
                                                uint8_t* const* d_desc32, int capacity, int32_t* const* d_n_out, const orbx_bounds* bounds,
                                                int window_size, float nnratio, int check_orientation, int32_t* const* d_matches12,
                                                int32_t* const* d_nmatches, int32_t* counts_all /* host, n_frames */) {
  if (!m || n_frames < 1 || !d_imgs || !d_kps || !d_desc32 || !d_n_out || !bounds || !d_matches12 || !d_nmatches || !counts_all)
    return ORBX_E_BADARG;
  const int n = m->n;
  Batch B;
  B.nFrames = n_frames;
  B.lo.resize(n); B.hi.resize(n); B.dN.resize(n);
  B.countsAll = counts_all;
  for (int r = 0; r < n; r++) {
    orbx_multi_shard_range(n_frames, n, r, &B.lo[r], &B.hi[r]);
    if (B.hi[r] - B.lo[r] > m->padCap) { m->err = "a block is larger than max_batch_per_device"; return ORBX_E_BADARG; }
    B.dN[r] = d_n_out[r];
  }
  // as many batches in flight as the device contexts take (their lanes; two in the two-half-batches mode): one more first
  // waits for the oldest, whose error -- if any -- this call returns
  const int maxInFlight = m->depth > 0 ? m->depth : 2;
  while ((int)m->inflight.size() >= maxInFlight) {
    const int w = orbx_multi_wait_one(m);
    if (w != ORBX_OK) return w;
  }
  for (int r = 0; r < n; r++) {
    const int nb = B.hi[r] - B.lo[r];
    if (nb == 0) continue;
    Cmd c;
    c.kind = 0; c.nb = nb; c.img = d_imgs[r]; c.width = width; c.height = height; c.stride = stride; c.frameStride = frame_stride_bytes;
    c.kps = d_kps[r]; c.desc = d_desc32[r]; c.capacity = capacity; c.nOut = d_n_out[r]; c.b = *bounds; c.window = window_size;
    c.nnratio = nnratio; c.checkOri = check_orientation; c.m12 = d_matches12[r]; c.nm = d_nmatches[r];
    push(*m->workers[r], c);
  }
  int rcAll = ORBX_OK;
  std::string errIssue;
  for (int r = 0; r < n; r++) {
    std::string e;
    const int rc = B.hi[r] - B.lo[r] == 0 ? ORBX_OK : drain(*m->workers[r], &e);
    if (rc != ORBX_OK && rcAll == ORBX_OK) { rcAll = rc; errIssue = e; }
  }
  if (rcAll != ORBX_OK) {
    // Some block of THIS batch was not issued.  The batches issued before it are complete batches: they are finished as
    // usual -- blocks waited for, counts gathered into their counts_all, their own errors reported (the first one wins over
    // the issue error: it is older) -- and only what the other devices queued for this batch is drained and forgotten.
    int older = ORBX_OK;
    std::string errOlder;
    while (!m->inflight.empty()) {
      const int w = orbx_multi_wait_one(m);
      if (w != ORBX_OK && older == ORBX_OK) { older = w; errOlder = m->err; }
    }
    for (int r = 0; r < n; r++) (void)orbx_wait(m->ctx[r]);
    m->err = older != ORBX_OK ? errOlder : errIssue;
    return older != ORBX_OK ? older : rcAll;
  }
  m->inflight.push_back(B);
  return ORBX_OK;
}

// Completes the oldest batch in flight: every worker waits for its block, then the per-frame keypoint counts of all blocks are
// all-gathered over RCCL into every device's copy -- on the collective streams, while the contexts' own streams already run the
// batches issued after it -- and written to that batch's counts_all in global frame order.
int orbx_multi_wait_one(orbx_multi* m) {
  if (!m) return ORBX_E_BADARG;
  if (m->inflight.empty()) return ORBX_OK;
  const Batch B = m->inflight.front();
  m->inflight.pop_front();
  for (int r = 0; r < m->n; r++) {
    if (B.hi[r] - B.lo[r] == 0) continue;
    Cmd c;
    c.kind = 1;
    push(*m->workers[r], c);
  }
  int rcAll = ORBX_OK;
  for (int r = 0; r < m->n; r++) {
    std::string e;
    const int rc = B.hi[r] - B.lo[r] == 0 ? ORBX_OK : drain(*m->workers[r], &e);
    if (rc != ORBX_OK && rcAll == ORBX_OK) { rcAll = rc; m->err = e; }
  }
  if (rcAll != ORBX_OK) return rcAll;
  return gatherCounts(m, B);
}

int orbx_multi_wait(orbx_multi* m) {
  if (!m) return ORBX_E_BADARG;
  int rcAll = ORBX_OK;
  while (!m->inflight.empty()) {
    const int rc = orbx_multi_wait_one(m);
    if (rc != ORBX_OK && rcAll == ORBX_OK) rcAll = rc;
  }
  return rcAll;
}

// The synchronous form: issue + wait.
int orbx_multi_extract_match_batch_device(orbx_multi* m, int n_frames, const uint8_t* const* d_imgs, int width, int height, int stride,
                                          size_t frame_stride_bytes, orbx_keypoint* const* d_kps, uint8_t* const* d_desc32, int capacity,
                                          int32_t* const* d_n_out, const orbx_bounds* bounds, int window_size, float nnratio,
                                          int check_orientation, int32_t* const* d_matches12, int32_t* const* d_nmatches,
                                          int32_t* counts_all /* host, n_frames */) {
  if (!m) return ORBX_E_BADARG;
  int rc = orbx_multi_wait(m);
  if (rc != ORBX_OK) return rc;
  rc = orbx_multi_extract_match_batch_device_async(m, n_frames, d_imgs, width, height, stride, frame_stride_bytes, d_kps, d_desc32,
                                                   capacity, d_n_out, bounds, window_size, nnratio, check_orientation, d_matches12,
                                                   d_nmatches, counts_all);
  if (rc != ORBX_OK) return rc;
  return orbx_multi_wait(m);
}

}  // extern "C"
