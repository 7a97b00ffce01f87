// host_util.h -- small host-side helpers shared by the C-ABI translation units.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <exception>
#include <mutex>
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

// Worker threads that cannot outlive their scope and cannot take the process down: the destructor joins (a joinable
// std::thread that is destroyed calls std::terminate -- what an exception thrown between emplace_back and join used to do),
// a worker's exception is kept and rethrown by wait() on the caller's thread (where `guarded` turns it into an SR_E* code),
// and a thread the system refuses (std::system_error) runs its work on the calling thread instead.
class ThreadGroup {
 public:
  explicit ThreadGroup(size_t expected) { pool_.reserve(expected); }  // (no reallocation while threads exist)
  ThreadGroup(const ThreadGroup&) = delete;
  ThreadGroup& operator=(const ThreadGroup&) = delete;
  ~ThreadGroup() { join_all(); }
  template <typename F>
  void run(F fn) {
    auto body = [this, fn]() mutable {
      try {
        fn();
      } catch (...) {
        std::lock_guard<std::mutex> g(mu_);
        if (!first_) first_ = std::current_exception();
      }
    };
    if (pool_.size() < pool_.capacity()) {
      try {
        pool_.emplace_back(body);
        return;
      } catch (const std::system_error&) {  // refused: below, on this thread
      }
    }
    body();
  }
  // the calling thread's own share: exceptions are kept like a worker's, so that wait() reports the first of all
  template <typename F>
  void run_here(F fn) {
    try {
      fn();
    } catch (...) {
      std::lock_guard<std::mutex> g(mu_);
      if (!first_) first_ = std::current_exception();
    }
  }
  void wait() {
    join_all();
    if (first_) { std::exception_ptr e = first_; first_ = nullptr; std::rethrow_exception(e); }
  }

 private:
  void join_all() noexcept {
    for (std::thread& t : pool_)
      if (t.joinable()) t.join();
  }
  std::vector<std::thread> pool_;
  std::mutex mu_;
  std::exception_ptr first_;
};

// fn(i0, i1) over [0, n) on up to `max_threads` (default 16) host threads; one call on the caller's thread when n is small
template <typename F>
inline void parallel_ranges(size_t n, size_t min_per_thread, F&& fn, size_t max_threads = 16) {
  const unsigned hw = std::thread::hardware_concurrency();
  size_t nt = std::min<size_t>(max_threads, hw ? hw : 1);
  if (min_per_thread) nt = std::min(nt, n / min_per_thread);
  if (nt <= 1) { fn((size_t)0, n); return; }
  ThreadGroup g(nt - 1);
  for (size_t t = 0; t + 1 < nt; t++) {
    const size_t a = n * t / nt, b = n * (t + 1) / nt;
    g.run([&fn, a, b]() { fn(a, b); });
  }
  g.run_here([&]() { fn(n * (nt - 1) / nt, n); });
  g.wait();
}

}  // namespace srhost
