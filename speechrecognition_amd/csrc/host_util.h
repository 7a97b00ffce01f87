// host_util.h -- small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/srgpu.h"

#define SRHOST_POOL_GLOBAL 0   // MixtureModel::GLOBAL_POOLING  (sietill/Mixtures.hpp:20-24)
#define SRHOST_POOL_MIXTURE 1  // MixtureModel::MIXTURE_POOLING
#define SRHOST_POOL_NONE 2     // MixtureModel::NO_POOLING

namespace srhost {

// finalised tables, one row per density in mixture order (what sr_model_create consumes)
struct MixsetTables {
  uint32_t dim = 0;
  std::vector<uint32_t> dens_off;
  std::vector<double> means, inv_vars, norm, logw;
  std::vector<uint32_t> dens_mean, dens_var;  // accumulator rows (MixtureDensity{mean_idx, var_idx}) per density
  uint32_t n_mean = 0, n_var = 0;
};

// returns nullptr on success or the reference's error text (sietill/Mixtures.cpp:753-825)
const char* load_mixset(const char* path, uint32_t dim, int pooling, MixsetTables* out);

// stores the message for sr_last_error() and returns `code`
int set_error(int code, const char* msg);

// Exception barrier of the C ABI (include/srgpu.h: "No exceptions cross this boundary"): every extern "C" entry runs its
// body through this, so that a failed host allocation (std::bad_alloc from new / std::vector), an over-long container
// (std::length_error), a refused std::thread (std::system_error) or anything else becomes an SR_E* code plus
// sr_last_error() text instead of std::terminate -> SIGABRT in the caller's process.
template <typename F>
int guarded(const char* entry, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return set_error(SR_ENOMEM, (std::string(entry) + ": out of host memory").c_str());
  } catch (const std::length_error& e) {
    return set_error(SR_ELIMIT, (std::string(entry) + ": size too large for a host container (" + e.what() + ")").c_str());
  } catch (const std::exception& e) {
    return set_error(SR_EINTERNAL, (std::string(entry) + ": unexpected exception: " + e.what()).c_str());
  } catch (...) {
    return set_error(SR_EINTERNAL, (std::string(entry) + ": unexpected exception").c_str());
  }
}

// fn(i0, i1) over [0, n) on up to 16 host threads (one call on the caller's thread when n is small or threads are refused)
template <typename F>
inline void parallel_ranges(size_t n, size_t min_per_thread, F&& fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  size_t nt = std::min<size_t>(16, hw ? hw : 1);
  if (min_per_thread) nt = std::min(nt, n / min_per_thread);
  if (nt <= 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> pool;
  pool.reserve(nt);
  size_t started = 0;
  try {
    for (; started + 1 < nt; started++) pool.emplace_back(fn, n * started / nt, n * (started + 1) / nt);
  } catch (const std::system_error&) {  // a refused thread: its range and the rest run here
  }
  fn(n * started / nt, n);
  for (auto& th : pool) th.join();
}

}  // namespace srhost
