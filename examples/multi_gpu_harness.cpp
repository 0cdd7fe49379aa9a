// multi_gpu_harness.cpp — a C++ host driving the batched local BA over several GPUs of one node through the C ABI alone
// (lld_ba_multi_*, include/lld_amd.h): one process, one host thread + one context per shard inside the library, the block partition of
// the windows, and the gather of the fixed-stride result records on the first device - SURVEY.md 7 step 7 / 8e, north_star.
//
//   multi_gpu_harness <in> <out> [devices] [repeats]
//     <in>       int32 n_windows, then n_windows problems in the layout of `harness ba` (tests/test_cpp_multi_gpu.py writes it)
//     devices    "all" (default: every visible device) or a comma list; a device may repeat ("0,0": two shards on GPU 0 - how a 1-GPU box
//                exercises partition, threads and gather)
//     <out>      per window the outputs in the layout of `harness ba`, then int32 n_shards and (first, count) per shard, int32 records verified
// Prints one line per repeat: windows/s of solve + gather, the slowest shard's solve and gather times.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lld_amd.hpp"

namespace {
struct Reader {
  FILE* f;
  explicit Reader(const char* path) : f(std::fopen(path, "rb")) { if (!f) throw std::runtime_error(std::string("cannot open ") + path); }
  ~Reader() { std::fclose(f); }
  template <class T> void get(T* p, size_t n) { if (n && std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("short read"); }
  template <class T> void get(std::vector<T>& v, size_t n) { v.resize(n); get(v.data(), n); }
};
struct Writer {
  FILE* f;
  explicit Writer(const char* path) : f(std::fopen(path, "wb")) { if (!f) throw std::runtime_error(std::string("cannot open ") + path); }
  ~Writer() { std::fclose(f); }
  template <class T> void put(const T* p, size_t n) { if (n && std::fwrite(p, sizeof(T), n, f) != n) throw std::runtime_error("short write"); }
  template <class T> void put(const std::vector<T>& v) { put(v.data(), v.size()); }
};
void check(int st, const char* what) { if (st != LLD_OK) throw std::runtime_error(std::string(what) + ": " + lld_status_string(st)); }
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: multi_gpu_harness <in> <out> [all | d0,d1,...] [repeats]\n"); return 2; }
  try {
    std::vector<int32_t> devices;
    const std::string dev_arg = argc > 3 ? argv[3] : "all";
    if (dev_arg == "all") { const int n = lld_device_count(); for (int d = 0; d < n; d++) devices.push_back(d); }
    else for (size_t at = 0; at < dev_arg.size();) { size_t e = dev_arg.find(',', at); if (e == std::string::npos) e = dev_arg.size(); devices.push_back(std::atoi(dev_arg.substr(at, e - at).c_str())); at = e + 1; }
    if (devices.empty()) { std::fprintf(stderr, "multi_gpu_harness: %s\n", lld_status_string(LLD_ERR_NO_DEVICE)); return 1; }
    const int repeats = argc > 4 ? std::max(1, std::atoi(argv[4])) : 1;
    Reader r(argv[1]);
    int32_t nw = 0; r.get(&nw, 1);
    std::vector<lld_amd::BAWindow> wins((size_t)nw);
    double gamma = 1.0;
    for (auto& w : wins) {
      int32_t h[8]; r.get(h, 8);
      double camg[6]; r.get(camg, 6);
      w.cam = lld_camera{camg[0], camg[1], camg[2], camg[3], camg[4]}; gamma = camg[5];
      w.n_free_cams = h[1];
      r.get(w.cam_qt, 7 * (size_t)h[0]);
      r.get(w.pt_xyz, 3 * (size_t)h[2]); r.get(w.pt_obs_start, (size_t)h[2] + 1); r.get(w.pt_obs_cam, h[3]);
      r.get(w.pt_obs_uvr, 3 * (size_t)h[3]); r.get(w.pt_obs_inv_sigma2, h[3]);
      r.get(w.line_x0, 3 * (size_t)h[4]); r.get(w.line_dir, 3 * (size_t)h[4]); r.get(w.ln_obs_start, (size_t)h[4] + 1);
      r.get(w.ln_obs_cam, h[5]); r.get(w.ln_obs_left, 4 * (size_t)h[5]); r.get(w.ln_obs_right, 4 * (size_t)h[5]);
      r.get(w.ln_obs_octave, 2 * (size_t)h[5]);
    }
    std::vector<lld_ba_window> views;
    for (const auto& w : wins) views.push_back(w.view());
    lld_ba_params p; lld_ba_params_default(&p); p.gamma = gamma;
    lld_ba_multi* m = nullptr;
    check(lld_ba_multi_create((int32_t)devices.size(), devices.data(), nw, views.data(), &p, &m), "lld_ba_multi_create");
    int32_t verified = 0;
    for (int rep = 0; rep < repeats; rep++) {
      const auto t0 = std::chrono::steady_clock::now();
      check(lld_ba_multi_solve(m, nullptr), "lld_ba_multi_solve");
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      check(lld_ba_multi_verify_gathered(m, &verified), "lld_ba_multi_verify_gathered");
      double solve_ms = 0, gather_ms = 0; lld_ba_multi_times_ms(m, &solve_ms, &gather_ms);
      std::printf("multi: %d windows on %zu shard(s): %.1f windows/s (%.3f ms solve + gather; slowest shard: solve %.3f ms, gather %.3f ms), %d gathered records verified\n",
                  nw, devices.size(), 1e3 * nw / ms, ms, solve_ms, gather_ms, verified);
    }
    Writer wr(argv[2]);
    for (int i = 0; i < nw; i++) {
      const lld_amd::BAWindow& w = wins[(size_t)i];
      lld_amd::BAOutput o;
      o.cam_qt.resize(w.cam_qt.size()); o.pt_xyz.resize(w.pt_xyz.size()); o.line_x0.resize(w.line_x0.size()); o.line_dir.resize(w.line_dir.size());
      o.pt_obs_outlier.resize(w.pt_obs_cam.size()); o.ln_edge_outlier.resize(2 * w.ln_obs_cam.size()); o.line_removed.resize(w.line_x0.size() / 3);
      lld_ba_result res{};
      res.cam_qt = o.cam_qt.data(); res.pt_xyz = o.pt_xyz.data(); res.line_x0 = o.line_x0.data(); res.line_dir = o.line_dir.data();
      res.pt_obs_outlier = o.pt_obs_outlier.data(); res.ln_edge_outlier = o.ln_edge_outlier.data(); res.line_removed = o.line_removed.data();
      check(lld_ba_multi_download(m, i, &res), "lld_ba_multi_download");
      wr.put(o.cam_qt); wr.put(o.pt_xyz); wr.put(o.line_x0); wr.put(o.line_dir);
      wr.put(o.pt_obs_outlier); wr.put(o.ln_edge_outlier); wr.put(o.line_removed);
      const double chi2[2] = {res.stats.chi2_round1, res.stats.chi2_final};
      wr.put(chi2, 2);
      const int32_t st[4] = {res.stats.lm_iterations[0], res.stats.lm_iterations[1], res.stats.aborted, res.stats.n_lines_removed};
      wr.put(st, 4);
    }
    const int32_t ns = (int32_t)devices.size();
    wr.put(&ns, 1);
    for (int d = 0; d < ns; d++) { int32_t fc[2]; lld_ba_multi_shard(nw, ns, d, &fc[0], &fc[1]); wr.put(fc, 2); }
    wr.put(&verified, 1);
    lld_ba_multi_destroy(m);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "multi_gpu_harness: %s\n", e.what());
    return 1;
  }
}
