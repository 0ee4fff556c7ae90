// A small persistent pool for the host stages of the post-processing (one image per task): threads are created once
// per detector instead of once per call, and the first exception of a batch is re-thrown on the caller's thread.
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ocr {

class ThreadPool {
 public:
  explicit ThreadPool(int threads) {
    for (int t = 0; t < threads; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~ThreadPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int size() const { return (int)workers_.size(); }

  // runs fn(0) .. fn(n - 1), the caller's thread included; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      n_ = n;
      next_ = 0;
      pending_ = n;
      error_ = nullptr;
      ++generation_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
    if (error_) std::rethrow_exception(error_);
  }

 private:
  void work() {
    for (;;) {
      int i;
      const std::function<void(int)>* fn;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (!fn_ || next_ >= n_) return;
        i = next_++;
        fn = fn_;
      }
      try {
        (*fn)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(m_);
        if (!error_) error_ = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, next_ = 0, pending_ = 0;
  unsigned long generation_ = 0;
  bool stop_ = false;
  std::exception_ptr error_;
};

}  // namespace ocr
