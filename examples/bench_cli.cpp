// bench_cli.cpp -- the benchmark's timed step from a C++ host: no Python, no torch, only include/sepfinder.h.
//
// What it runs is the loop of the reference's caller (ros_ws/src/multi_robot_separators/scripts/find_separators.py:59-133)
// in batch operation at BASELINE.json configs[1]: per step one NetVLAD query of robot B's N keyframes against robot A's N
// descriptors (find_matches) and one estimate_transformation per returned candidate, as the library's begin / retire
// pair -- step k is issued before step k - 1 is retired, so the device never waits for the host:
//
//     for (k = 0; k < K; ++k) { if (in_flight == 6) { sf_step_retire(h, &r); consume(r); }  sf_step_issue(h, slot_a, slot_b); }
//     while (in_flight) { sf_step_retire(h, &r); consume(r); }      // sf_step_issue never waits for the device
//
// The synthetic workload follows SURVEY.md section 8(d) / multi_robot_slam_separators_amd/synth.py (same recipe, its
// own random stream): K features per keyframe inside a 640 x 480 pin-hole image, 20 % of B's keyframes are revisits of
// the same-index A keyframe (40 % of the points seen from a pose <= 30 deg / 2 m away, 2 cm noise, 5 % descriptor
// bit flips), every B NetVLAD row is a perceptual alias of A's (distance ~0.05 < netvlad_distance).
//
// Build:  make -C multi_robot_slam_separators_amd/csrc bench_cli
// Usage:  examples/bench_cli [--keyframes 10000] [--features 500] [--dim 4096] [--iterations 500] [--steps 200]
//                            [--warmup 20] [--true-frac 0.2] [--seed 12345] [--estimator 3d3d|pnp]
// Prints ONE JSON line (value = candidate pairs verified per second over the timed steps).
#include <hip/hip_runtime_api.h>
#include <sepfinder.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {   // xoshiro256**, seeded through splitmix64
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    for (auto& v : s) { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; v = z ^ (z >> 31); }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uniform(double a, double b) { return a + (b - a) * uniform(); }
  double spare = 0.0; bool has_spare = false;
  double normal() {
    if (has_spare) { has_spare = false; return spare; }
    double u, v, q;
    do { u = uniform(-1.0, 1.0); v = uniform(-1.0, 1.0); q = u * u + v * v; } while (q >= 1.0 || q == 0.0);
    const double f = std::sqrt(-2.0 * std::log(q) / q);
    spare = v * f; has_spare = true;
    return u * f;
  }
};

constexpr double FX = 600.0, FY = 600.0, CX = 320.0, CY = 240.0;
constexpr int WIDTH = 640, HEIGHT = 480;

void make_point(Rng& g, float* p) {   // base frame (x forward, y left, z up); projection uniform inside the image
  const double u = g.uniform(0.0, WIDTH - 1.0), v = g.uniform(0.0, HEIGHT - 1.0), z = g.uniform(1.0, 20.0);
  p[0] = (float)z; p[1] = (float)(-(u - CX) / FX * z); p[2] = (float)(-(v - CY) / FY * z);
}
void project(const float* p, sf_keypoint& k) {
  const double xc = -(double)p[1], yc = -(double)p[2], zc = (double)p[0];
  k.x = (float)(FX * xc / zc + CX); k.y = (float)(FY * yc / zc + CY);
  k.size = 31.f; k.angle = -1.f; k.response = 0.01f; k.octave = 0; k.class_id = -1;
}

template <class F>
void parallel_rows(int n, F body) {
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  for (unsigned t = 0; t < hw; ++t) th.emplace_back([=]() { for (int i = (int)t; i < n; i += (int)hw) body(i); });
  for (auto& x : th) x.join();
}

#define CHECK_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s -> %s\n", #e, hipGetErrorString(_e)); return 2; } } while (0)
#define CHECK_SF(e) do { int _r = (e); if (_r != SF_OK) { fprintf(stderr, "%s -> %d: %s\n", #e, _r, sf_last_error(h)); return 3; } } while (0)

}  // namespace

int main(int argc, char** argv) {
  int n = 10000, k = 500, dim = 4096, iterations = 500, steps = 200, warmup = 20, est = 0, overlap_steps = 1;
  double true_frac = 0.2;
  uint64_t seed = 12345;
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string a = argv[i];
    if (a == "--keyframes") n = atoi(argv[i + 1]);
    else if (a == "--features") k = atoi(argv[i + 1]);
    else if (a == "--dim") dim = atoi(argv[i + 1]);
    else if (a == "--iterations") iterations = atoi(argv[i + 1]);
    else if (a == "--steps") steps = atoi(argv[i + 1]);
    else if (a == "--warmup") warmup = atoi(argv[i + 1]);
    else if (a == "--true-frac") true_frac = atof(argv[i + 1]);
    else if (a == "--seed") seed = strtoull(argv[i + 1], nullptr, 10);
    else if (a == "--estimator") est = std::string(argv[i + 1]) == "pnp" ? 1 : 0;
    else if (a == "--overlap-steps") { overlap_steps = atoi(argv[i + 1]); }
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
  }
  constexpr int COLS = 32;
  const auto t_gen0 = std::chrono::steady_clock::now();
  // ---- synthetic inputs on the host ------------------------------------------------------------------------
  std::vector<uint8_t> desc[2];
  std::vector<float> xyz[2];
  std::vector<sf_keypoint> kp[2];
  for (int r = 0; r < 2; ++r) { desc[r].resize((size_t)n * k * COLS); xyz[r].resize((size_t)n * k * 3); kp[r].resize((size_t)n * k); }
  std::vector<uint8_t> is_true((size_t)n, 0);
  const int n_ov = (int)std::lround(0.4 * k);
  parallel_rows(n, [&](int i) {
    Rng g(seed * 1000003ull + (uint64_t)i);
    for (int r = 0; r < 2; ++r) {
      float* x = xyz[r].data() + (size_t)i * k * 3;
      uint8_t* d = desc[r].data() + (size_t)i * k * COLS;
      for (int j = 0; j < k; ++j) make_point(g, x + 3 * j);
      for (int j = 0; j < k * COLS; j += 8) { const uint64_t w = g.next(); memcpy(d + j, &w, 8); }
    }
    if (g.uniform() < true_frac) {
      is_true[i] = 1;
      // T_gt: rotation <= 30 deg about a random axis, translation <= 2 m;  p_A = R p_B + t  =>  p_B = R^T (p_A - t)
      double ax[3] = {g.normal(), g.normal(), g.normal()};
      const double an = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
      for (double& v : ax) v /= an;
      const double ang = g.uniform(0.0, 30.0) * M_PI / 180.0, c = std::cos(ang), s = std::sin(ang), C = 1.0 - c;
      const double R[9] = {c + ax[0] * ax[0] * C, ax[0] * ax[1] * C - ax[2] * s, ax[0] * ax[2] * C + ax[1] * s,
                           ax[1] * ax[0] * C + ax[2] * s, c + ax[1] * ax[1] * C, ax[1] * ax[2] * C - ax[0] * s,
                           ax[2] * ax[0] * C - ax[1] * s, ax[2] * ax[1] * C + ax[0] * s, c + ax[2] * ax[2] * C};
      double t[3] = {g.normal(), g.normal(), g.normal()};
      const double tn = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]), tl = g.uniform(0.0, 2.0);
      for (double& v : t) v *= tl / tn;
      std::vector<int> sel(k), dst(k);
      for (int j = 0; j < k; ++j) sel[j] = dst[j] = j;
      for (int j = 0; j < n_ov; ++j) {     // partial Fisher-Yates: n_ov distinct sources and destinations
        std::swap(sel[j], sel[j + (int)(g.next() % (uint64_t)(k - j))]);
        std::swap(dst[j], dst[j + (int)(g.next() % (uint64_t)(k - j))]);
      }
      const float* xa = xyz[0].data() + (size_t)i * k * 3;
      float* xb = xyz[1].data() + (size_t)i * k * 3;
      const uint8_t* da = desc[0].data() + (size_t)i * k * COLS;
      uint8_t* db = desc[1].data() + (size_t)i * k * COLS;
      for (int j = 0; j < n_ov; ++j) {
        const float* pa = xa + 3 * sel[j];
        const double q[3] = {pa[0] - t[0], pa[1] - t[1], pa[2] - t[2]};
        for (int a = 0; a < 3; ++a)
          xb[3 * dst[j] + a] = (float)(R[a] * q[0] + R[3 + a] * q[1] + R[6 + a] * q[2] + 0.02 * g.normal());
        for (int b = 0; b < COLS; ++b) {
          uint8_t flips = 0;
          for (int bit = 0; bit < 8; ++bit) flips |= (uint8_t)((g.uniform() < 0.05) << bit);
          db[(size_t)dst[j] * COLS + b] = da[(size_t)sel[j] * COLS + b] ^ flips;
        }
      }
    }
    for (int r = 0; r < 2; ++r)
      for (int j = 0; j < k; ++j) project(xyz[r].data() + ((size_t)i * k + j) * 3, kp[r][(size_t)i * k + j]);
  });
  std::vector<float> nv_a((size_t)n * dim), nv_b((size_t)n * dim);
  parallel_rows(n, [&](int i) {
    Rng g(seed * 7000003ull + 17ull + (uint64_t)i);
    float* a = nv_a.data() + (size_t)i * dim;
    float* b = nv_b.data() + (size_t)i * dim;
    double na = 0.0;
    for (int d = 0; d < dim; ++d) { a[d] = (float)g.normal(); na += (double)a[d] * a[d]; }
    const float ia = (float)(1.0 / std::sqrt(na));
    double nb = 0.0;
    const double sc = 0.05 / std::sqrt((double)dim);
    for (int d = 0; d < dim; ++d) { a[d] *= ia; b[d] = (float)(a[d] + sc * g.normal()); nb += (double)b[d] * b[d]; }
    const float ib = (float)(1.0 / std::sqrt(nb));
    for (int d = 0; d < dim; ++d) b[d] *= ib;
  });
  const double t_gen = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gen0).count();

  // ---- the handle ------------------------------------------------------------------------------------------
  sf_params p;
  sf_default_params(&p);
  p.fx = FX; p.fy = FY; p.cx = CX; p.cy = CY; p.image_width = WIDTH; p.image_height = HEIGHT;
  const float L[12] = {0, 0, 1, 0, -1, 0, 0, 0, 0, -1, 0, 0};     // base <- optical (z_cam forward = x_base)
  memcpy(p.local_transform, L, sizeof(L));
  p.iterations = iterations;
  p.estimation_type = est;
  p.netvlad_dimensions = dim;
  p.netvlad_max_matches_nb = n;             // batch operation: walk every row
  p.nn_precision = 1;
  p.max_features = k;
  p.desc_bytes = COLS;
  p.store_capacity = 2 * n;
  sf_handle h = nullptr;
  {
    const int rc = sf_create(&p, 0, &h);
    if (rc != SF_OK) { fprintf(stderr, "sf_create -> %d: %s\n", rc, sf_last_error(nullptr)); return 3; }
    // --overlap-steps 0: both steps in flight on the handle's stream (SF_OPT_STEP_OVERLAP, default 1: two streams)
    CHECK_SF(sf_set_option(h, SF_OPT_STEP_OVERLAP, overlap_steps));
  }
  // ---- make everything resident in HBM (untimed) -----------------------------------------------------------
  int32_t slot[2] = {-1, -1};
  {
    const int CH = 1024;
    uint8_t* d_desc = nullptr; float* d_xyz = nullptr; sf_keypoint* d_kp = nullptr;
    CHECK_HIP(hipMalloc((void**)&d_desc, (size_t)CH * k * COLS));
    CHECK_HIP(hipMalloc((void**)&d_xyz, (size_t)CH * k * 12));
    CHECK_HIP(hipMalloc((void**)&d_kp, (size_t)CH * k * sizeof(sf_keypoint)));
    for (int r = 0; r < 2; ++r)
      for (int s0 = 0; s0 < n; s0 += CH) {
        const int m = std::min(CH, n - s0);
        CHECK_HIP(hipMemcpy(d_desc, desc[r].data() + (size_t)s0 * k * COLS, (size_t)m * k * COLS, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_xyz, xyz[r].data() + (size_t)s0 * k * 3, (size_t)m * k * 12, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_kp, kp[r].data() + (size_t)s0 * k, (size_t)m * k * sizeof(sf_keypoint), hipMemcpyHostToDevice));
        int32_t first = -1;
        CHECK_SF(sf_store_add_keyframes_device(h, m, k, COLS, d_desc, d_xyz, d_kp, &first));
        CHECK_SF(sf_synchronize(h));
        if (slot[r] < 0) slot[r] = first;
      }
    (void)hipFree(d_desc); (void)hipFree(d_xyz); (void)hipFree(d_kp);
    float* d_nv = nullptr;
    CHECK_HIP(hipMalloc((void**)&d_nv, (size_t)n * dim * 4));
    CHECK_HIP(hipMemcpy(d_nv, nv_a.data(), (size_t)n * dim * 4, hipMemcpyHostToDevice));
    CHECK_SF(sf_nn_append_received_f32_device(h, d_nv, n, dim));      // robot A's descriptors, as received by B
    CHECK_SF(sf_synchronize(h));
    CHECK_HIP(hipMemcpy(d_nv, nv_b.data(), (size_t)n * dim * 4, hipMemcpyHostToDevice));
    CHECK_SF(sf_nn_append_local_f32_device(h, d_nv, n, dim));         // robot B's own
    CHECK_SF(sf_synchronize(h));
    (void)hipFree(d_nv);
  }
  const int32_t slot_a = slot[0], slot_b = slot[1];

  // ---- warm-up: the driver's count, then until three consecutive steps agree within 3 % (at most 50) --------
  sf_step_result r;
  long long pairs = 0;
  int inflight = 0;
  const int depth = 6;        // steps kept in flight (the library's default SF_OPT_STEP_DEPTH)
  auto issue = [&]() -> int {
    int rc;
    if (inflight >= depth) { rc = sf_step_retire(h, &r); if (rc != SF_OK) return rc; pairs += r.n_matches; --inflight; }
    rc = sf_step_issue(h, slot_a, slot_b);      // queues the whole step on the device; does not wait
    if (rc != SF_OK) return rc;
    ++inflight;
    return SF_OK;
  };
  for (int i = 0; i < warmup; ++i) CHECK_SF(issue());
  {
    std::vector<double> ts;
    for (int i = 0; i < 50; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      CHECK_SF(issue());
      ts.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      const size_t m = ts.size();
      if (m >= 5 && std::max({ts[m - 1], ts[m - 2], ts[m - 3]}) < 1.03 * std::min({ts[m - 1], ts[m - 2], ts[m - 3]})) break;
    }
  }
  while (inflight) { CHECK_SF(sf_step_retire(h, &r)); --inflight; }
  CHECK_SF(sf_synchronize(h));

  // ---- the timed steps ---------------------------------------------------------------------------------------
  pairs = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int s = 0; s < steps; ++s) CHECK_SF(issue());
  while (inflight) { CHECK_SF(sf_step_retire(h, &r)); pairs += r.n_matches; --inflight; }
  CHECK_SF(sf_synchronize(h));
  const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  // ---- the last step's outcome against the planted truth -------------------------------------------------------
  int correct = 0, accepted = 0;
  for (int i = 0; i < r.n_matches; ++i) {
    const sf_match& m = r.matches[i];
    const bool truth = m.idx_local == m.idx_other && is_true[(size_t)m.idx_local];
    const bool ok = r.record_of_match[i] >= 0 && r.records[r.record_of_match[i]].success;
    correct += ok == truth;
    accepted += ok;
  }
  printf("{\"metric\": \"candidate keyframe-pair verifications/sec (NetVLAD NN + ORB match + RANSAC) @1/2/4/8 GPU\", "
         "\"value\": %.1f, \"unit\": \"pairs/s\", \"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, "
         "\"host\": \"C++ (examples/bench_cli.cpp): sf_step_issue + sf_step_retire over include/sepfinder.h, no torch\", \"overlap_steps\": %d, "
         "\"config\": {\"workload\": \"BASELINE configs[1] shape: 2 robots x %d keyframes, %d-D fp32 NetVLAD, %d x %d-bit "
         "ORB per keyframe, <= %d RANSAC hypotheses per pass (%s), %.0f %% true revisits\"}, "
         "\"check\": {\"matches_last_step\": %d, \"accepted_last_step\": %d, \"decisions_matching_ground_truth\": %d, "
         "\"streamed\": %d}, \"input_generation_s\": %.2f}\n",
         (double)pairs / elapsed, steps, warmup, elapsed / steps * 1e3, overlap_steps, n, dim, k, COLS * 8, iterations,
         est ? "PnP" : "3D-3D", 100.0 * true_frac, r.n_matches, accepted, correct, r.streamed, t_gen);
  sf_destroy(h);
  return correct == r.n_matches ? 0 : 4;
}
