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

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbx.h"

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

struct orbx_multi {
  int n = 0;
  std::vector<int> dev;
  std::vector<orbx_ctx*> ctx;
  std::vector<ncclComm_t> comm;       // n > 1 only
  std::vector<hipStream_t> st;        // one stream per device for the collective
  std::vector<int32_t*> dSend, dRecv; // per device: counts of its block (padded), counts of all blocks
  int padCap = 0;                     // entries per block in the gather buffers
  std::string err;
};

extern "C" {

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
  for (int r = 0; r < m->n; r++) {
    if (r < (int)m->dev.size()) (void)hipSetDevice(m->dev[r]);
    if (r < (int)m->comm.size() && m->comm[r] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[r]);
    if (r < (int)m->dSend.size() && m->dSend[r]) (void)hipFree(m->dSend[r]);
    if (r < (int)m->dRecv.size() && m->dRecv[r]) (void)hipFree(m->dRecv[r]);
    if (r < (int)m->st.size() && m->st[r]) (void)hipStreamDestroy(m->st[r]);
    if (r < (int)m->ctx.size() && m->ctx[r]) orbx_destroy(m->ctx[r]);
  }
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
  if (n_devices > 1) {
    std::string e;
    if (!g_rccl.load(&e)) { orbx_multi_destroy(m); return ORBX_E_RCCL; }
    m->comm.assign(n_devices, nullptr);
    const ncclResult_t q = g_rccl.CommInitAll(m->comm.data(), n_devices, device_ids);
    if (q != 0) { orbx_multi_destroy(m); return ORBX_E_RCCL; }
  }
  *out = m;
  return ORBX_OK;
}

int orbx_multi_size(const orbx_multi* m) { return m ? m->n : ORBX_E_BADARG; }
orbx_ctx* orbx_multi_ctx(orbx_multi* m, int r) { return (m && r >= 0 && r < m->n) ? m->ctx[r] : nullptr; }
const char* orbx_multi_last_error(const orbx_multi* m) { return m ? m->err.c_str() : "null context"; }

// Every device extracts the frames of its block (already resident in its HBM) and matches the block's consecutive pairs
// (2k, 2k + 1); then the per-frame keypoint counts of all blocks are all-gathered over RCCL into every device's copy and
// returned to the host in global frame order.  d_* [r] are device r's arrays for ITS block, laid out as for
// orbx_extract_match_batch_device; blocks = orbx_multi_shard_range(n_frames, n, r).
int orbx_multi_extract_match_batch_device(orbx_multi* m, int n_frames, const uint8_t* const* d_imgs, int width, int height, int stride,
                                          size_t frame_stride_bytes, orbx_keypoint* const* d_kps, uint8_t* const* d_desc32, int capacity,
                                          int32_t* const* d_n_out, const orbx_bounds* bounds, int window_size, float nnratio,
                                          int check_orientation, int32_t* const* d_matches12, int32_t* const* d_nmatches,
                                          int32_t* counts_all /* host, n_frames */) {
  if (!m || n_frames < 1 || !d_imgs || !d_kps || !d_desc32 || !d_n_out || !bounds || !d_matches12 || !d_nmatches || !counts_all)
    return ORBX_E_BADARG;
  const int n = m->n;
  std::vector<int> lo(n), hi(n);
  for (int r = 0; r < n; r++) {
    orbx_multi_shard_range(n_frames, n, r, &lo[r], &hi[r]);
    if (hi[r] - lo[r] > m->padCap) { m->err = "a block is larger than max_batch_per_device"; return ORBX_E_BADARG; }
  }
  // 1. issue every device's block (stream-ordered: the call returns once the launches are queued)
  std::vector<std::vector<int32_t>> first(n), second(n);
  for (int r = 0; r < n; r++) {
    const int nb = hi[r] - lo[r];
    if (nb == 0) continue;
    for (int p = 0; p + 1 < nb; p += 2) { first[r].push_back(p); second[r].push_back(p + 1); }
    const int rc = orbx_extract_match_batch_device_async(m->ctx[r], nb, d_imgs[r], width, height, stride, frame_stride_bytes, d_kps[r],
                                                         d_desc32[r], capacity, d_n_out[r], (int)first[r].size(), first[r].data(),
                                                         second[r].data(), bounds, window_size, nnratio, check_orientation,
                                                         d_matches12[r], d_nmatches[r], nullptr);
    if (rc != ORBX_OK) {
      m->err = std::string("device block: ") + orbx_last_error(m->ctx[r]);
      for (int q = 0; q < n; q++) (void)orbx_wait(m->ctx[q]);
      return rc;
    }
  }
  int rcAll = ORBX_OK;
  for (int r = 0; r < n; r++) {
    const int rc = orbx_wait(m->ctx[r]);
    if (rc != ORBX_OK && rcAll == ORBX_OK) { rcAll = rc; m->err = std::string("device block: ") + orbx_last_error(m->ctx[r]); }
  }
  if (rcAll != ORBX_OK) return rcAll;
  // 2. counts -> padded send buffers (-1 beyond the block), all-gather, back to the host from device 0's copy
  const int per = m->padCap;
  for (int r = 0; r < n; r++) {
    if (hipSetDevice(m->dev[r]) != hipSuccess) return ORBX_E_HIP;
    const int nb = hi[r] - lo[r];
    if (hipMemsetAsync(m->dSend[r], 0xff, sizeof(int32_t) * (size_t)per, m->st[r]) != hipSuccess) return ORBX_E_HIP;
    if (nb > 0 && hipMemcpyAsync(m->dSend[r], d_n_out[r], sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToDevice, m->st[r]) != hipSuccess)
      return ORBX_E_HIP;
  }
  if (n > 1) {
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
  std::vector<int32_t> all((size_t)per * n);
  for (int r = 0; r < n; r++) {
    if (hipSetDevice(m->dev[r]) != hipSuccess) return ORBX_E_HIP;
    if (r == 0 && hipMemcpyAsync(all.data(), m->dRecv[0], sizeof(int32_t) * all.size(), hipMemcpyDeviceToHost, m->st[0]) != hipSuccess)
      return ORBX_E_HIP;
    if (hipStreamSynchronize(m->st[r]) != hipSuccess) return ORBX_E_HIP;  // every device's copy of the counts is complete
  }
  for (int r = 0; r < n; r++)
    for (int f = lo[r]; f < hi[r]; f++) counts_all[f] = all[(size_t)r * per + (f - lo[r])];
  return ORBX_OK;
}

}  // extern "C"
