// f2 (host side): the file reads of a batch.  Replaces the reference's dataloader worker processes reading audio files
// (ssak/utils/dataset.py:630-645 -> ssak/utils/audio.py:24-105 load_audio; wav2vec_train.py:360 runs 6 workers) for the byte
// ranges the caller has cut out of PCM WAV files: n ranges (path, file offset, length) are read by `threads` native threads with
// pread straight into the destinations (slices of one pinned staging buffer).  The Python form of the same loop -- a task per
// reader on a ThreadPoolExecutor -- spent 3 of its ~3 ms per batch of 32 files handing tasks over under the interpreter lock.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "kernels.h"

extern "C" int ssak_read_ranges(const char* const* paths, const int64_t* file_offsets, const int64_t* nbytes, void* const* dst, int n,
                                int threads) {
  SSAK_REQUIRE(n >= 0 && (n == 0 || (paths && file_offsets && nbytes && dst)), "read_ranges: null pointer");
  if (n == 0) return SSAK_OK;
  const int T = std::max(1, std::min(threads, n));
  std::atomic<int> next{0}, failed{-1};
  std::vector<int> err(n, 0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) return;
      if (nbytes[i] <= 0) continue;
      const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
      if (fd < 0) {
        err[i] = errno ? errno : EIO;
        int expect = -1;
        failed.compare_exchange_strong(expect, i);
        continue;
      }
      int64_t got = 0;
      while (got < nbytes[i]) {
        const ssize_t r = pread(fd, (char*)dst[i] + got, (size_t)(nbytes[i] - got), (off_t)(file_offsets[i] + got));
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) {
          err[i] = r < 0 ? errno : ENODATA;  // ENODATA: the file ends inside the range
          int expect = -1;
          failed.compare_exchange_strong(expect, i);
          break;
        }
        got += r;
      }
      close(fd);
    }
  };
  if (T == 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    pool.reserve(T - 1);
    try {
      for (int t = 1; t < T; ++t) pool.emplace_back(work);
    } catch (...) {  // no more threads to be had: the ones that started (and this one) take all the ranges
    }
    work();
    for (auto& th : pool) th.join();
  }
  const int f = failed.load();
  SSAK_REQUIRE(f < 0, "read_ranges: %s: %s", paths[f], err[f] == ENODATA ? "short read (the file ends inside the range)" : strerror(err[f]));
  return SSAK_OK;
}
