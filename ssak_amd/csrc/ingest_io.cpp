// f2 (host side): the file reads of a batch.  Replaces the reference's dataloader worker processes reading audio files
// (ssak/utils/dataset.py:630-645 -> ssak/utils/audio.py:24-105 load_audio; wav2vec_train.py:360 runs 6 workers) for the byte
// ranges the caller has cut out of PCM WAV files: n ranges (path, file offset, length) are read by `threads` native threads with
// pread straight into the destinations (slices of one pinned staging buffer).  The Python form of the same loop -- a task per
// reader on a ThreadPoolExecutor -- spent 3 of its ~3 ms per batch of 32 files handing tasks over under the interpreter lock.
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "kernels.h"

namespace {

// One batch of ranges; the readers (pool threads + the caller) claim indices from `next`.
struct ReadJob {
  const char* const* paths;
  const int64_t* file_offsets;
  const int64_t* nbytes;
  void* const* dst;
  int n;
  std::atomic<int> next{0}, failed{-1};
  std::vector<int> err;
};

void read_some(ReadJob& j) {
  for (;;) {
    const int i = j.next.fetch_add(1, std::memory_order_relaxed);
    if (i >= j.n) return;
    if (j.nbytes[i] <= 0) continue;
    const int fd = open(j.paths[i], O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
      j.err[i] = errno ? errno : EIO;
      int expect = -1;
      j.failed.compare_exchange_strong(expect, i);
      continue;
    }
    int64_t got = 0;
    while (got < j.nbytes[i]) {
      const ssize_t r = pread(fd, (char*)j.dst[i] + got, (size_t)(j.nbytes[i] - got), (off_t)(j.file_offsets[i] + got));
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) {
        j.err[i] = r < 0 ? errno : ENODATA;  // ENODATA: the file ends inside the range
        int expect = -1;
        j.failed.compare_exchange_strong(expect, i);
        break;
      }
      got += r;
    }
    close(fd);
  }
}

// The reader threads live as long as the process (round 5 spawned and joined them per call: ~25 us per thread per batch, and a
// cold file system wants the readers waiting in pread, not being created).  One job at a time (`call`): the callers are the one
// prefetch thread of an ingest, and two ingests simply take turns.  The pool grows to the largest `threads - 1` ever asked for;
// a job wakes only as many workers as it wants (`want`), the others stay parked.
struct ReaderPool {
  std::mutex call;  // one ssak_read_ranges at a time
  std::mutex m;
  std::condition_variable wake, done;
  std::vector<std::thread> workers;
  ReadJob* job = nullptr;
  uint64_t generation = 0;
  int want = 0, running = 0;
  bool stop = false;

  void worker_main() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      wake.wait(lk, [&] { return stop || (generation != seen && want > 0); });
      if (stop) return;
      seen = generation;
      --want;
      ReadJob* j = job;
      lk.unlock();
      read_some(*j);
      lk.lock();
      if (--running == 0) done.notify_all();
    }
  }

  // runs `j` on `helpers` pool threads + the calling thread; returns when every range is read or has failed
  void run(ReadJob& j, int helpers) {
    std::lock_guard<std::mutex> one(call);
    {
      std::unique_lock<std::mutex> lk(m);
      try {
        while ((int)workers.size() < helpers) workers.emplace_back([this] { worker_main(); });
      } catch (...) {  // no more threads to be had: the ones that exist (and the caller) take all the ranges
      }
      helpers = std::min(helpers, (int)workers.size());
      job = &j;
      want = running = helpers;
      ++generation;
    }
    if (helpers > 0) wake.notify_all();
    read_some(j);
    std::unique_lock<std::mutex> lk(m);
    running -= want;  // helpers that have not woken up yet are not needed any more (every range is claimed): they stay parked
    want = 0;
    done.wait(lk, [&] { return running == 0; });
    job = nullptr;
  }

  ~ReaderPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    wake.notify_all();
    for (auto& t : workers) t.join();
  }
};

ReaderPool* g_pool = nullptr;
std::once_flag g_pool_once;
// a forked child has none of the parent's threads: it starts over with an empty pool (the parent's object is leaked, not destroyed:
// its mutexes may be held by threads that do not exist in the child)
void pool_after_fork_child() { g_pool = new ReaderPool(); }
ReaderPool& pool() {
  std::call_once(g_pool_once, [] {
    g_pool = new ReaderPool();
    pthread_atfork(nullptr, nullptr, pool_after_fork_child);
  });
  return *g_pool;
}

}  // namespace

extern "C" int ssak_read_ranges(const char* const* paths, const int64_t* file_offsets, const int64_t* nbytes, void* const* dst, int n,
                                int threads) {
  SSAK_REQUIRE(n >= 0 && (n == 0 || (paths && file_offsets && nbytes && dst)), "read_ranges: null pointer");
  if (n == 0) return SSAK_OK;
  const int T = std::max(1, std::min(std::min(threads, n), 256));
  ReadJob j;
  j.paths = paths, j.file_offsets = file_offsets, j.nbytes = nbytes, j.dst = dst, j.n = n;
  j.err.assign(n, 0);
  if (T == 1)
    read_some(j);
  else
    pool().run(j, T - 1);
  const int f = j.failed.load();
  SSAK_REQUIRE(f < 0, "read_ranges: %s: %s", paths[f], j.err[f] == ENODATA ? "short read (the file ends inside the range)" : strerror(j.err[f]));
  return SSAK_OK;
}

// Drops the page-cache pages of the files (posix_fadvise DONTNEED after an fdatasync of dirty pages): the next read of them
// comes from the storage device.  Measurement aid for the cold-cache ingest figure (bench.py: ingest.cold); returns the number
// of files it could not open.
extern "C" int ssak_drop_file_cache(const char* const* paths, int n) {
  SSAK_REQUIRE(n >= 0 && (n == 0 || paths), "drop_file_cache: null pointer");
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
      ++bad;
      continue;
    }
    fdatasync(fd);  // dirty pages cannot be dropped
    if (posix_fadvise(fd, 0, 0, POSIX_FADV_DONTNEED) != 0) ++bad;
    close(fd);
  }
  return bad;
}
