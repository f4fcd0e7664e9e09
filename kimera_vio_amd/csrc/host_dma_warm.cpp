// DMA engine warm-up (round 6, profiles/r6_analysis.md section "6.5 ms inside hipMemcpyAsync").
//
// The HIP runtime moves a device <-> pinned-host transfer with one of the device's SDMA engines; it asks the HSA runtime
// which engines are idle (hsa_amd_memory_copy_engine_status) and takes one of them, and the HSA runtime creates an
// engine's queue the first time a transfer lands on it: ~6.5 ms inside that hipMemcpyAsync call, where the call otherwise
// takes 2 us (API trace: hsa_queue_create 5 - 9 ms).  A host that runs ahead of the device finds the last engine still
// busy and is handed the next one, so the first context of a process met six to nine such stalls at scattered steps of
// its first ~110 -- 6 - 8 % of a freshly started front-end's step loop, and a whole-length idle gap when the stall falls
// right behind a synchronisation.  One small transfer on every engine and in both directions, once per process and device,
// at context creation, moves that cost out of the step loop.  Plain HSA calls (the runtime is already initialised by HIP;
// hsa_init / hsa_shut_down only move its reference count); a failure here is not an error of the context: the step loop
// then simply pays the stalls as before.
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace kvfe {

// dev_buf: device memory of the context's device, host_buf: pinned host memory (hipHostMalloc), both >= bytes
int warm_dma_engines(int device_ordinal, void* dev_buf, void* host_buf, size_t bytes) {
  static const bool off = [] { const char* e = std::getenv("KVFE_DMA_WARMUP"); return e && std::atoi(e) == 0; }();   // (A/B switch)
  if (off) return 0;
  static std::mutex mu;
  static std::vector<int> done;
  std::lock_guard<std::mutex> lk(mu);
  if (std::find(done.begin(), done.end(), device_ordinal) != done.end()) return 0;
  done.push_back(device_ordinal);   // (one attempt per process and device)
  if (hsa_init() != HSA_STATUS_SUCCESS) return -1;
  int engines = 0, rc = 0;
  hsa_amd_pointer_info_t di, hi;
  di.size = sizeof(di);
  hi.size = sizeof(hi);
  hsa_signal_t sig;
  sig.handle = 0;
  do {
    if (hsa_amd_pointer_info(dev_buf, &di, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
        hsa_amd_pointer_info(host_buf, &hi, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) {
      rc = -2;
      break;
    }
    const hsa_agent_t gpu = di.agentOwner, cpu = hi.agentOwner;
    if (hsa_signal_create(1, 0, nullptr, &sig) != HSA_STATUS_SUCCESS) {
      rc = -3;
      break;
    }
    for (int dir = 0; dir < 2; dir++) {   // 0: device -> host (the per-step records), 1: host -> device (staged frames)
      const hsa_agent_t da = dir == 0 ? cpu : gpu, sa = dir == 0 ? gpu : cpu;
      void* dst = dir == 0 ? host_buf : dev_buf;
      const void* src = dir == 0 ? dev_buf : host_buf;
      uint32_t mask = 0;
      if (hsa_amd_memory_copy_engine_status(da, sa, &mask) != HSA_STATUS_SUCCESS) continue;
      for (int e = 0; e < 16; e++) {
        if (!(mask & (1u << e))) continue;
        hsa_signal_store_relaxed(sig, 1);
        if (hsa_amd_memory_async_copy_on_engine(dst, da, src, sa, bytes, 0, nullptr, sig, (hsa_amd_sdma_engine_id_t)(1u << e),
                                                false) != HSA_STATUS_SUCCESS)
          continue;
        // (bounded: 2 s; a transfer that never signals must not hang context creation)
        if (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) >= 1) {
          rc = -4;
          break;
        }
        engines++;
      }
      if (rc) break;
    }
  } while (false);
  if (sig.handle && rc != -4) hsa_signal_destroy(sig);
  hsa_shut_down();
  if (std::getenv("KVFE_HOST_PROF")) std::fprintf(stderr, "KVFE_HOST_PROF warm_dma_engines: device %d, %d engine-direction pairs, rc %d\n", device_ordinal, engines, rc);
  return rc;
}

}  // namespace kvfe
