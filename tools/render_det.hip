// Determinism probe for the decoder path of csrc/render.hip (GPU box): ln3d_query_points twice on the same inputs, outputs must be
// bit-identical.   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLN3D_RENDER_ABL=n] tools/render_det.hip -o build/render_det
#include "../ln3diff_amd/csrc/render.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
  const int H = 128, W = 128; const int64_t P = 1 << 20;
  std::vector<float> planes((size_t)3 * H * W * 32), pts(P * 3), w0(64 * 32), b0(64), w1(4 * 64), b1(4);
  uint64_t s = 12345; auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) / 8388608.0f - 1.0f; };
  for (auto& v : planes) v = 4.f * rnd();
  for (auto& v : pts) v = 0.45f * rnd();
  for (auto& v : w0) v = rnd(); for (auto& v : b0) v = 0.1f * rnd(); for (auto& v : w1) v = rnd(); for (auto& v : b1) v = rnd();
  float *dp, *dpts, *dw0, *db0, *dw1, *db1, *sig[2], *rgb[2], *scal;
  hipMalloc(&dp, planes.size() * 4); hipMalloc(&dpts, pts.size() * 4); hipMalloc(&dw0, 8192); hipMalloc(&db0, 256); hipMalloc(&dw1, 1024); hipMalloc(&db1, 16);
  hipMalloc(&scal, LN3D_RENDER_SCRATCH_FLOATS * 4);
  for (int i = 0; i < 2; ++i) { hipMalloc(&sig[i], P * 4); hipMalloc(&rgb[i], P * 12); }
  hipMemcpy(dp, planes.data(), planes.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dpts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw0, w0.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(db0, b0.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w1.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db1, b1.data(), 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 2; ++i) {
      int rc = ln3d_query_points(dp, H, W, dpts, P, dw0, db0, dw1, db1, 0.9f, sig[i], rgb[i], scal, nullptr);
      if (rc) { printf("rc %d\n", rc); return 1; }
    }
    hipDeviceSynchronize();
    std::vector<float> a(P), b(P);
    hipMemcpy(a.data(), sig[0], P * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), sig[1], P * 4, hipMemcpyDeviceToHost);
    int64_t bad = 0, first = -1;
    for (int64_t i = 0; i < P; ++i) if (memcmp(&a[i], &b[i], 4)) { if (first < 0) first = i; ++bad; }
    int hist[64] = {0};
    for (int64_t i = 0; i < P; ++i) if (memcmp(&a[i], &b[i], 4)) hist[i & 63]++;
    printf("  lanes:"); for (int l = 0; l < 64; ++l) if (hist[l]) printf(" %d:%d", l, hist[l]); printf("\n");
    { int64_t g0 = -1; int ng = 0; for (int64_t i = 0; i < P; ++i) if (memcmp(&a[i], &b[i], 4) && (i >> 6) != g0) { g0 = i >> 6; if (ng++ < 12) printf(" g%lld(b%lld,w%lld)", (long long)g0, (long long)((g0 % 8192) / 4), (long long)(g0 % 4)); } printf("  groups %d\n", ng); }
    printf("rep %d: %lld of %lld sigma differ (first %lld: %g vs %g), lane of first %lld\n", rep, (long long)bad, (long long)P, (long long)first,
           first >= 0 ? a[first] : 0.f, first >= 0 ? b[first] : 0.f, (long long)(first & 63));
  }
  return 0;
}
