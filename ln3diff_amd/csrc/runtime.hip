// Stream facilities of the C ABI (include/ln3d.h): HIP streams restricted to a subset of the compute units, so that independent
// sub-batches of the denoise loop ("lanes", ln3diff_amd/sgm/sampling.py) own disjoint halves of the chip and run fully
// asynchronously - one lane's HBM-bound phases (norms, residual read-modify-write epilogues) fall under the other's MFMA main
// loops instead of every CU of the device doing the same phase at the same time.
//
// CU mask bit i of hipExtStreamCreateWithCUMask addresses XCD (i % 8), CU (i / 8) of that XCD on this part (the driver deals
// the bits round-robin over the 8 XCDs), so "every other CU of every XCD" keeps the GEMMs' XCD-aware tile walk (block b on XCD
// b % 8) intact, which a split by XCD would not.
#include "common.h"
#include "../../include/ln3d.h"

namespace {
struct Masked { hipStream_t s; int cus; };
constexpr int kMaxMasked = 32;
Masked g_masked[kMaxMasked];
std::atomic<int> g_nmasked{0};
int device_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    n = v;
  }
  return n;
}
}  // namespace

// compute units a launch on `s` can occupy: the mask's population for streams made by ln3d_stream_create_cu_mask, else the device's
int ln3d_stream_cus(hipStream_t s) {
  const int n = g_nmasked.load(std::memory_order_acquire);
  for (int i = 0; i < n; ++i)
    if (g_masked[i].s == s) return g_masked[i].cus;
  return device_cus();
}

extern "C" int ln3d_device_cus(void) { return device_cus(); }

extern "C" int ln3d_stream_create_cu_mask(const uint32_t* mask, int words, void** stream_out) {
  if (!mask || words <= 0 || words > 32 || !stream_out) return LN3D_ERR_BAD_ARG;
  int bits = 0;
  for (int w = 0; w < words; ++w) bits += __builtin_popcount(mask[w]);
  const int cus = device_cus();
  if (bits <= 0) return LN3D_ERR_BAD_ARG;
  if (bits > cus) bits = cus;
  const int slot = g_nmasked.load(std::memory_order_acquire);
  if (slot >= kMaxMasked) return LN3D_ERR_UNSUPPORTED;
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) { (void)hipGetLastError(); return LN3D_ERR_LAUNCH; }
  g_masked[slot].s = s; g_masked[slot].cus = bits;
  g_nmasked.store(slot + 1, std::memory_order_release);
  *stream_out = (void*)s;
  return LN3D_OK;
}

extern "C" int ln3d_stream_cu_count(void* stream) { return ln3d_stream_cus((hipStream_t)stream); }
