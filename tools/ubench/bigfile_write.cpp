// What bounds a BIG output file (config 4 at its stated size: two good files of 16 GiB each)?  Two writer threads, one file each,
// sequential write() of 4 MiB pieces, rate printed per GiB, in four flavours:
//   plain      nothing else
//   sfr        sync_file_range(SYNC_FILE_RANGE_WRITE) on every 64 MiB written (writeback starts at once instead of when the
//              dirty limit is reached)
//   sfr_drop   ... and, two steps behind, wait for that range's writeback and drop it from the page cache (posix_fadvise DONTNEED):
//              the file never holds more than ~192 MiB of cache
//   (any)      on another directory, e.g. /dev/shm, as a control
//   bigfile_write DIR GiB plain|sfr|sfr_drop [files]
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const size_t gib = argc > 2 ? (size_t)atoll(argv[2]) : 16;
    const std::string mode = argc > 3 ? argv[3] : "plain";
    const int files = argc > 4 ? atoi(argv[4]) : 2;
    const size_t piece = 4u << 20, step = 64u << 20;
    std::vector<std::vector<double>> per(files);
    std::vector<std::thread> th;
    const double t00 = now();
    for (int f = 0; f < files; ++f)
        th.emplace_back([&, f] {
            std::vector<char> src(piece, 'A' + f);
            char path[256];
            snprintf(path, sizeof(path), "%s/big%d.bin", dir, f);
            int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) { perror("open"); return; }
            double t0 = now();
            size_t done = 0;
            for (size_t g = 0; g < gib; ++g) {
                for (size_t o = 0; o < (1u << 30); o += piece) {
                    if (write(fd, src.data(), piece) != (ssize_t)piece) { perror("write"); return; }
                    done += piece;
                    if (mode != "plain" && done % step == 0) {
                        sync_file_range(fd, (off_t)(done - step), (off_t)step, SYNC_FILE_RANGE_WRITE);
                        if (mode == "sfr_drop" && done >= 3 * step) {
                            const off_t a = (off_t)(done - 3 * step);
                            sync_file_range(fd, a, (off_t)step, SYNC_FILE_RANGE_WAIT_BEFORE | SYNC_FILE_RANGE_WRITE | SYNC_FILE_RANGE_WAIT_AFTER);
                            posix_fadvise(fd, a, (off_t)step, POSIX_FADV_DONTNEED);
                        }
                    }
                }
                const double t1 = now();
                per[f].push_back(1.073741824 / (t1 - t0));
                t0 = t1;
            }
            close(fd);
            unlink(path);
        });
    for (auto& t : th) t.join();
    const double total = now() - t00;
    printf("%s %s: %d file(s) x %zu GiB in %.2f s = %.2f GB/s in all; GB/s per GiB, file 0:", dir, mode.c_str(), files, gib, total, files * gib * 1.073741824 / total);
    for (double v : per[0]) printf(" %.1f", v);
    printf("\n");
    return 0;
}
